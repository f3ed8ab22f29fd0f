#!/bin/bash
# Round 4, call 4: second version of the latency-ordered draft kernels (x loads only from the lanes that own a chunk)
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_hip_fused.py tests/test_hip_attn_oproj.py tests/test_hip_ops.py -x -q -m gpu > gpurun_out/r04/c4_unit_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r04/c4_unit_tests.txt
tail -3 gpurun_out/r04/c4_unit_tests.txt
timeout 300 python profiles/ktrace_probe.py > gpurun_out/r04/ktrace_1b_after2.txt 2>&1; echo "rc=$?" >> gpurun_out/r04/ktrace_1b_after2.txt
grep -E "^\[slot|===" gpurun_out/r04/ktrace_1b_after2.txt | head -12
timeout 200 python profiles/draft_probe.py 6 300 > gpurun_out/r04/draft_probe2.json 2> gpurun_out/r04/draft_probe2.err; cat gpurun_out/r04/draft_probe2.json
timeout 300 python profiles/async_probe.py > gpurun_out/r04/async_probe2.json 2> gpurun_out/r04/async_probe2.err; cat gpurun_out/r04/async_probe2.json
