#!/bin/bash
# Round 5, call 13: rope_store with one item per thread: parity tests (bit-exact goldens), c5t and c4 lines
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_attn_qkv.py tests/test_hip_fuzz.py -q -m gpu 2>&1 | tail -3
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_real_shapes_gpu.py -q -m gpu -k "qwen or rope or prefill or decoder_layer" 2>&1 | tail -3
for wl in c5t c4; do
  timeout 600 python bench.py --workload $wl > gpurun_out/r05/c13_bench_${wl}.json 2> gpurun_out/r05/c13_bench_${wl}.err
  python - $wl <<'PY'
import json, sys
wl = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r05/c13_bench_{wl}.json").read().strip().splitlines()[-1])
    print(wl, {k: d.get(k) for k in ("value", "ms_per_step", "mean_accepted_len", "cache_hit_rate", "ttft_p50_ms", "value_reference_protocol", "value_random_pair", "step_roofline_frac", "step_roofline_frac_target_only")})
except Exception as e:
    print(wl, "bench parse failed", e)
PY
done
