#!/bin/bash
# sanity of the rebuilt library after the attention-inside experiment was removed
timeout 600 python -m pytest tests/test_hip_chain.py tests/test_hip_fused.py -q -m gpu 2>&1 | tail -2
timeout 200 python profiles/draft_probe.py 6 300 2>/dev/null
