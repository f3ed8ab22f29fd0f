#!/bin/bash
# last sanity of the round on the committed tree: smoke, the chain tests, the ABI test
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python -m pytest tests/test_hip_chain.py tests/test_abi.py -q 2>&1 | tail -1
