# Round-3 GPU call 16: final c4 lines (the driver's --steps 20 and the default 40)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03; mkdir -p $O
timeout 500 python bench.py > $O/bench_c4_n1.json 2> $O/bench_c4.err
timeout 500 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c4_n1_s20.json 2> /dev/null
timeout 300 python -m pytest tests/test_model_gpu.py tests/test_real_shapes_gpu.py -m gpu -q -x 2>&1 | tail -n 2
python - <<'PY'
import json
for f in ("bench_c4_n1", "bench_c4_n1_s20"):
    d = json.load(open(f"gpurun_out/r03/{f}.json"))
    print(f, {k: d.get(k) for k in ("value", "steps", "ms_per_step", "mean_accepted_len", "cache_hit_rate", "ttft_p50_ms", "value_reference_protocol")})
PY
