#!/bin/bash
# Round 4, call 5: barrier-free prologue (producer waves + LDS counters) in the fused norm+GEMM kernels; rmsnorm slab classes
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_hip_fused.py -x -q -m gpu > gpurun_out/r04/c7_unit_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r04/c7_unit_tests.txt
tail -3 gpurun_out/r04/c7_unit_tests.txt
timeout 300 python profiles/ktrace_probe.py > gpurun_out/r04/ktrace_1b_after5.txt 2>&1; echo "rc=$?" >> gpurun_out/r04/ktrace_1b_after5.txt
grep -E "^\[slot|===" gpurun_out/r04/ktrace_1b_after5.txt | head -8
timeout 200 python profiles/draft_probe.py 6 300 > gpurun_out/r04/draft_probe5.json 2> gpurun_out/r04/draft_probe5.err; cat gpurun_out/r04/draft_probe5.json
timeout 300 python profiles/async_probe.py > gpurun_out/r04/async_probe5.json 2> gpurun_out/r04/async_probe5.err; cat gpurun_out/r04/async_probe5.json
