#!/bin/bash
# prefill GEMM sweep 2 (U, k-steps per barrier, 7-wave workgroups)
mkdir -p gpurun_out/r03
timeout 300 python profiles/micro/prefill_gemm_probe2.py > gpurun_out/r03/pf_probe4.txt 2>&1
echo "rc=$?" >> gpurun_out/r03/pf_probe4.txt
tail -50 gpurun_out/r03/pf_probe4.txt
