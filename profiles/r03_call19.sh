#!/bin/bash
# the driver's round-end commands on the final code (after the prefill GEMM change)
mkdir -p gpurun_out/r03
timeout 560 python -m pytest tests/ -x -q -m gpu > gpurun_out/r03/gpu_suite_final.txt 2>&1
echo "rc=$?" >> gpurun_out/r03/gpu_suite_final.txt
tail -4 gpurun_out/r03/gpu_suite_final.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" > gpurun_out/r03/smoke_final.txt 2>&1
tail -2 gpurun_out/r03/smoke_final.txt
