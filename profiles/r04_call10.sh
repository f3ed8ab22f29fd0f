#!/bin/bash
mkdir -p gpurun_out/r04
timeout 300 python -m pytest tests/test_hip_fused.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python profiles/ktrace_probe.py > gpurun_out/r04/ktrace_1b_c10.txt 2>&1
sed -n 4,12p gpurun_out/r04/ktrace_1b_c10.txt
timeout 200 python profiles/draft_probe.py 6 300 2>/dev/null | tee gpurun_out/r04/draft_probe10.json
timeout 300 python profiles/async_probe.py 2>/dev/null | tee gpurun_out/r04/async_probe10.json
SSD_ARGMAX_FUSED=0 timeout 300 python profiles/async_probe.py 2>/dev/null | tee gpurun_out/r04/async_probe10_noargmaxfuse.json
