#!/bin/bash
mkdir -p gpurun_out/r04
timeout 400 python bench.py --workload c2 --no-cpu-baseline > gpurun_out/r04/bench_c2_n1.json 2> gpurun_out/r04/bench_c2_n1.err; echo "c2 rc=$?"
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_c4_nocpu.json 2> gpurun_out/r04/bench_c4_nocpu.err; echo "c4 rc=$?"
python - <<'P'
import json
for f in ("bench_c2_n1", "bench_c4_nocpu"):
    try:
        d = json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], d["step_roofline_frac"], "|", r["kernel"][:60], r["frac"], r["avg_launch_us"], "|", {k: (v["us"], v["GBps"]) for k, v in r["per_kind"].items() if "draft" in k})
    except Exception as e:
        print(f, "no line", e)
P
tail -3 gpurun_out/r04/bench_c2_n1.err
