# Round-3 GPU call 15: the driver's own round-end commands on the final code -- pytest -m gpu (whole suite), smoke(), bench.py
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|destroy_process" | tail -30 > $O/t15_full_suite.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" > $O/smoke.txt 2>&1
timeout 500 python bench.py > $O/bench_c4_n1.json 2> $O/bench_c4.err
tail -n 5 $O/t15_full_suite.txt; tail -n 2 $O/smoke.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03/bench_c4_n1.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "mean_accepted_len", "cache_hit_rate", "ttft_p50_ms", "ttft_after_first_round_p50_ms", "step_roofline_frac", "value_reference_protocol")})
print(d["roofline"]["frac"], d["roofline"].get("rocprof"), d["roofline"]["traffic_source"][:40], d["cpu_baseline"]["value"])
PY
