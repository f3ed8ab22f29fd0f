#!/bin/bash
# Round 5, call 10: the two re-thresholded tests, the full-depth 70B run with the exact-arithmetic criterion, bench lines of every workload
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_reference_replays_gpu.py -q -m gpu -s -k "undamped" > gpurun_out/r05/c10_lockstep_undamped.txt 2>&1; grep -E "full size|passed|failed|Error|assert" gpurun_out/r05/c10_lockstep_undamped.txt | tail -6
SSD_FULL_70B=1 timeout 1500 python -m pytest tests/test_real_shapes_gpu.py -q -m gpu -s -k "full_depth_70b" > gpurun_out/r05/c10_full_depth_70b.txt 2>&1; grep -E "70B x|truth|last layer|argmax vs|passed|failed|skipped|Error" gpurun_out/r05/c10_full_depth_70b.txt | tail -8
for wl in c4 c2 c5t c4e; do
  timeout 600 python bench.py --workload $wl > gpurun_out/r05/c10_bench_${wl}.json 2> gpurun_out/r05/c10_bench_${wl}.err
  python - $wl <<'PY'
import json, sys
wl = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r05/c10_bench_{wl}.json").read().strip().splitlines()[-1])
    print(wl, {k: d.get(k) for k in ("value", "ms_per_step", "mean_accepted_len", "cache_hit_rate", "ttft_p50_ms", "value_reference_protocol", "value_random_pair", "step_roofline_frac", "step_roofline_frac_target_only")}, d["roofline"]["frac"])
except Exception as e:
    print(wl, "bench parse failed", e)
PY
done
