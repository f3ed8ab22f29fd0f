# Round-3 GPU call 12: deferred co-located draft prefill -- c4 line (TTFT), async engine tests, bench tests
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03
mkdir -p $O
timeout 500 python bench.py > $O/bench_c4_n1.json 2> $O/bench_c4.err
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_reference_replays_gpu.py tests/test_engine_temperature_gpu.py tests/test_eagle_gpu.py tests/test_bench_gpu.py -m gpu -q --no-header -p no:cacheprovider --timeout=300 2>&1 \
  | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|destroy_process" | tail -15 > $O/t12_engine.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03/bench_c4_n1.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "mean_accepted_len", "cache_hit_rate", "ttft_p50_ms", "ttft_after_first_round_p50_ms", "step_roofline_frac", "value_reference_protocol")})
PY
tail -n 4 $O/t12_engine.txt
