#!/bin/bash
# prefill GEMM: new default launch shapes -- tests, then the c4 / c2 bench lines
mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_real_shapes_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "prefill or pf or real or chunk" > gpurun_out/r03/pf_tests.txt 2>&1
echo "rc=$?" >> gpurun_out/r03/pf_tests.txt
tail -5 gpurun_out/r03/pf_tests.txt
timeout 600 python bench.py > gpurun_out/r03/bench_c4_pf2.json 2> gpurun_out/r03/bench_c4_pf2.err
echo "bench rc=$?"
cat gpurun_out/r03/bench_c4_pf2.json | cut -c1-900
timeout 300 python bench.py --workload c2 > gpurun_out/r03/bench_c2_pf2.json 2> gpurun_out/r03/bench_c2_pf2.err
cat gpurun_out/r03/bench_c2_pf2.json | cut -c1-600
