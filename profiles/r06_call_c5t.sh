#!/bin/bash
# c5t (Qwen3-32B + Qwen3-0.6B, async k = 7 f = 3, one GPU) line + the Qwen tests
mkdir -p gpurun_out/r06u
timeout 500 python bench.py --workload c5t --no-cpu-baseline > gpurun_out/r06u/bench_c5t.json 2> gpurun_out/r06u/bench_c5t.err
python - <<P
import json
d = json.loads(open("gpurun_out/r06u/bench_c5t.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("step_roofline_frac_target_only"), d.get("ttft_p50_ms"), d.get("mean_accepted_len"))
print({k: (v["us"], v["GBps"]) for k, v in d["roofline"]["per_kind"].items()})
P
timeout 600 python -m pytest tests -q -m gpu -k "qwen or Qwen" -x 2>&1 | tail -4
