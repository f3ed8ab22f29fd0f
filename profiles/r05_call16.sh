#!/bin/bash
# Round 5, call 16: the M-row segment at engine level beside a co-located verify (lock-step vs the oracle engine)
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_reference_replays_gpu.py -q -m gpu -s -k "resident_segments" > gpurun_out/r05/c16_lockstep_segment.txt 2>&1; grep -E "full size|passed|failed|Error|assert" gpurun_out/r05/c16_lockstep_segment.txt | tail -5
