#!/bin/bash
# Round 4, call 8: granule all-reduce (2 processes on one GPU, both protocols, fenced form), TP=2-on-one-GPU engine tests, the per-kind
# table of a TP = 4 / 8 rank with the new collective, the reverted fused kernels + kept attention / norm changes, c4 line
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_custom_ar_gpu.py tests/test_tp_one_gpu.py tests/test_hip_fused.py tests/test_hip_attn_oproj.py tests/test_hip_ops.py -x -q -m gpu > gpurun_out/r04/c8_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r04/c8_tests.txt
tail -4 gpurun_out/r04/c8_tests.txt
timeout 400 python profiles/tp_shard_probe.py 8 > gpurun_out/r04/tp_shard_per_kind.txt 2> gpurun_out/r04/tp_shard.err; echo "tp rc=$?"
tail -30 gpurun_out/r04/tp_shard_per_kind.txt
timeout 200 python profiles/draft_probe.py 6 300 > gpurun_out/r04/draft_probe8.json 2>/dev/null; cat gpurun_out/r04/draft_probe8.json
timeout 300 python profiles/async_probe.py > gpurun_out/r04/async_probe8.json 2>/dev/null; cat gpurun_out/r04/async_probe8.json
timeout 300 python bench.py --steps 20 --no-cpu-baseline > gpurun_out/r04/bench_c4_call8.json 2> gpurun_out/r04/bench_c4_call8.err; echo "c4 rc=$?"
python - <<'P'
import json
try:
    d = json.load(open("gpurun_out/r04/bench_c4_call8.json"))
    print("c4", d["value"], d["ms_per_step"], d.get("step_roofline_frac"), d.get("step_roofline_frac_target_only"), "ttft", d.get("ttft_p50_ms"), d.get("ttft_first_sighting_ms"), d.get("ttft_cold_process_ms"), d.get("mean_accepted_len"), d["roofline"]["frac"])
except Exception as e:
    print("c4 no line:", e)
P
