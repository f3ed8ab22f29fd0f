#!/bin/bash
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_hip_chain.py -x -q -m gpu 2>&1 | tail -2
for v in 0 1; do
  echo "== SSD_CHAIN_SEG=$v"
  SSD_CHAIN_SEG=$v timeout 200 python profiles/draft_probe.py 6 300 2>/dev/null
done
SSD_CHAIN_SEG=1 timeout 300 python profiles/ktrace_probe.py > gpurun_out/r04/ktrace_1b_chain.txt 2>&1
grep -A8 "slot 12\]" gpurun_out/r04/ktrace_1b_chain.txt | head -9
