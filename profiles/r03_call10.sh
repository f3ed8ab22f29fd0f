# Round-3 GPU call 10: the other workloads' bench lines (c2 = configs[1], c5t = configs[4]'s models, c4e = EAGLE-3) + the async draft round
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03
mkdir -p $O
timeout 200 python profiles/async_probe.py llama-3.1-8b > $O/async_probe5.json 2> $O/async_probe5.err
timeout 300 python bench.py --workload c2 > $O/bench_c2_n1.json 2> $O/bench_c2.err
timeout 400 python bench.py --workload c5t --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c5t_n1.json 2> $O/bench_c5t.err
timeout 500 python bench.py --workload c4e --steps 20 --warmup 3 --no-cpu-baseline --ref-seqs 1 --ref-output-len 256 > $O/bench_c4e_n1.json 2> $O/bench_c4e.err
cat $O/async_probe5.json; for w in c2 c5t c4e; do python - <<PY
import json
try:
    d = json.load(open("$O/bench_${w}_n1.json"))
    print("$w", {k: d.get(k) for k in ("value", "ms_per_step", "mean_accepted_len", "cache_hit_rate", "ttft_p50_ms", "step_roofline_frac", "value_reference_protocol")})
except Exception as e:
    print("$w", "ERR", e)
PY
done
