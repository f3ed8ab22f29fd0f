#!/bin/bash
mkdir -p gpurun_out/r04
timeout 300 python profiles/ktrace_pf_probe.py > gpurun_out/r04/ktrace_pf.txt 2>&1; echo "rc=$?"; cat gpurun_out/r04/ktrace_pf.txt | tail -40
