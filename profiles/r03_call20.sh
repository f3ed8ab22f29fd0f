#!/bin/bash
# rope_store_kv_parts (the prefill QKV GEMM's slabs consumed by the RoPE / KV-store kernel): bit-identity test, then the c4 line
mkdir -p gpurun_out/r03
timeout 45 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "rope" > gpurun_out/r03/rope_parts_tests.txt 2>&1
echo "rc=$?" >> gpurun_out/r03/rope_parts_tests.txt
tail -3 gpurun_out/r03/rope_parts_tests.txt
timeout 55 python bench.py > gpurun_out/r03/bench_c4_ropeparts.json 2> gpurun_out/r03/bench_c4_ropeparts.err
echo "bench rc=$?"
python - <<'P'
import json
try:
    d = json.load(open('gpurun_out/r03/bench_c4_ropeparts.json'))
    print(d['value'], d['ms_per_step'], d['ttft_p50_ms'], d['mean_accepted_len'], d['ttft_samples_ms'])
except Exception as e:
    print('no bench line', e)
P
