#!/bin/bash
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_hip_attn_qkv.py -x -q -m gpu > gpurun_out/r04/attn_qkv_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r04/attn_qkv_tests.txt; tail -12 gpurun_out/r04/attn_qkv_tests.txt
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_reference_replays_gpu.py tests/test_real_shapes_gpu.py -x -q -m gpu -k "qwen or Qwen" > gpurun_out/r04/qwen_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r04/qwen_tests.txt; tail -5 gpurun_out/r04/qwen_tests.txt
timeout 400 python bench.py --workload c5t --no-cpu-baseline > gpurun_out/r04/bench_c5t_qkvattn.json 2> gpurun_out/r04/bench_c5t_qkvattn.err; echo "c5t rc=$?"
SSD_FUSE_QKV_ATTN=0 timeout 400 python bench.py --workload c5t --no-cpu-baseline > gpurun_out/r04/bench_c5t_noqkvattn.json 2> gpurun_out/r04/bench_c5t_noqkvattn.err
python - <<'P'
import json
for w in ("bench_c5t_qkvattn", "bench_c5t_noqkvattn"):
    try:
        d = json.load(open(f"gpurun_out/r04/{w}.json"))
        print(w, d["value"], d["ms_per_step"], d.get("step_roofline_frac"), "ttft", d.get("ttft_p50_ms"), d.get("mean_accepted_len"), d.get("cache_hit_rate"), d["reference_protocol"]["stream_head"][:8])
    except Exception as e:
        print(w, "no line:", e)
P
