#!/bin/bash
# Round 5, call 8: split fork + undamped parity tests, the c4 line, kernel trace + PMC passes on c4
mkdir -p gpurun_out/r05
timeout 300 python -m pytest tests/test_hip_ops.py -q -m gpu -k "fork" > gpurun_out/r05/c8_fork_tests.txt 2>&1; tail -2 gpurun_out/r05/c8_fork_tests.txt
timeout 900 python -m pytest tests/test_real_shapes_gpu.py -q -m gpu -s -k "eight_layer" > gpurun_out/r05/c8_70b_cut.txt 2>&1; grep -E "70B x|passed|failed|Error" gpurun_out/r05/c8_70b_cut.txt | tail -5
timeout 300 python profiles/tree_seg_probe.py 300 > gpurun_out/r05/c8_tree_seg_probe.txt 2> gpurun_out/r05/c8_tree_seg_probe.err; cat gpurun_out/r05/c8_tree_seg_probe.txt
timeout 600 python bench.py > gpurun_out/r05/c8_bench_c4.json 2> gpurun_out/r05/c8_bench_c4.err; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05/c8_bench_c4.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "mean_accepted_len", "cache_hit_rate", "ttft_p50_ms", "value_reference_protocol", "value_random_pair", "step_roofline_frac")})
    print(d.get("random_pair")); print(d["roofline"]["frac"], d["roofline"].get("rocprof"))
except Exception as e:
    print("bench parse failed", e)
PY
tail -3 gpurun_out/r05/c8_bench_c4.err
timeout 1500 python -m pytest tests/test_reference_replays_gpu.py -q -m gpu -s -k "full_size" > gpurun_out/r05/c8_lockstep.txt 2>&1; grep -E "full size|passed|failed|Error|assert" gpurun_out/r05/c8_lockstep.txt | tail -12
bash profiles/collect_r05.sh trace > gpurun_out/r05/c8_collect_trace.log 2>&1; head -30 gpurun_out/r05/c4_kernel_stats.txt
bash profiles/collect_r05.sh pmc > gpurun_out/r05/c8_collect_pmc.log 2>&1; tail -5 gpurun_out/r05/c8_collect_pmc.log
