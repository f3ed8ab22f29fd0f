#!/bin/bash
# Round 5, call 15: the driver's round-end commands once more on another box (flakiness check of the committed tree)
mkdir -p gpurun_out/r05
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1800 python -m pytest tests/ -x -q -m gpu > gpurun_out/r05/c15_gpu_suite_full.txt 2>&1; grep -E "passed|failed" gpurun_out/r05/c15_gpu_suite_full.txt | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/r05/c15_gpu_suite_full.txt | head
timeout 600 python bench.py > gpurun_out/r05/c15_bench_c4.json 2> gpurun_out/r05/c15_bench_c4.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/c15_bench_c4.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "ttft_p50_ms", "value_reference_protocol", "value_random_pair")}, d["roofline"]["frac"], d["roofline"].get("rocprof"), d["roofline"].get("traffic_source"))
PY
