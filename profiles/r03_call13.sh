# Round-3 GPU call 13: prefill split-K partials consumed by the norms (kernel test, real-shape layer test, engine tests, c4 TTFT)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03
mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_real_shapes_gpu.py tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider --timeout=300 2>&1 \
  | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|destroy_process" | tail -15 > $O/t13.txt
timeout 500 python bench.py > $O/bench_c4_n1_pfparts.json 2> $O/bench_c4_pfparts.err
SSD_PF_PARTS=0 timeout 500 python bench.py --steps 5 --warmup 2 --ref-seqs 0 --no-roofline --no-cpu-baseline > $O/bench_c4_n1_nopfparts.json 2> /dev/null
python - <<'PY'
import json
for f in ("bench_c4_n1_pfparts", "bench_c4_n1_nopfparts"):
    d = json.load(open(f"gpurun_out/r03/{f}.json"))
    print(f, {k: d.get(k) for k in ("value", "ms_per_step", "ttft_p50_ms", "ttft_after_first_round_p50_ms", "ttft_samples_ms")})
PY
tail -n 4 $O/t13.txt
