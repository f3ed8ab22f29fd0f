#!/bin/bash
# Round 5, call 11: Qwen3-0.6B through the M-row segment (parity + A/B at the branch-slice widths of draft data-parallelism), c4 / c2 lines, full-depth 70B
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_hip_tree_segment.py -x -q -m gpu -s > gpurun_out/r05/c11_tree_seg_tests.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r05/c11_tree_seg_tests.txt | tail -3
for w in 24 12 6; do timeout 300 python profiles/tree_seg_probe.py 300 qwen3-0.6b $w 2> gpurun_out/r05/c11_probe_qwen_$w.err | tee -a gpurun_out/r05/c11_tree_seg_probe_qwen.txt; done
for w in 12 6; do timeout 300 python profiles/tree_seg_probe.py 300 llama-3.2-1b $w 2> gpurun_out/r05/c11_probe_1b_$w.err | tee -a gpurun_out/r05/c11_tree_seg_probe_1b_slices.txt; done
for wl in c4 c2; do
  timeout 600 python bench.py --workload $wl > gpurun_out/r05/c11_bench_${wl}.json 2> gpurun_out/r05/c11_bench_${wl}.err
  python - $wl <<'PY'
import json, sys
wl = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r05/c11_bench_{wl}.json").read().strip().splitlines()[-1])
    print(wl, {k: d.get(k) for k in ("value", "ms_per_step", "mean_accepted_len", "cache_hit_rate", "ttft_p50_ms", "value_reference_protocol", "value_random_pair", "step_roofline_frac", "step_roofline_frac_target_only")}, d["roofline"]["frac"], d["roofline"]["kernel"][:40])
except Exception as e:
    print(wl, "bench parse failed", e)
PY
done
SSD_FULL_70B=1 timeout 1500 python -m pytest tests/test_real_shapes_gpu.py -q -m gpu -s -k "full_depth_70b" > gpurun_out/r05/c11_full_depth_70b.txt 2>&1; grep -E "70B x|truth\||last layer|argmax vs|passed|failed|skipped" gpurun_out/r05/c11_full_depth_70b.txt | grep -v "print" | tail -8
