#!/bin/bash
# Round 4, call 3: the latency-ordered draft kernels (x / slab loads first, o_proj weights behind the K/V tiles, unrolled LDS combines,
# rmsnorm slab loads in one round trip): unit + engine tests, in-kernel ramp profile after, draft probes, the async lock-step test, c2.
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_hip_fused.py tests/test_hip_attn_oproj.py tests/test_hip_ops.py tests/test_hip_fuzz.py tests/test_model_gpu.py -x -q -m gpu > gpurun_out/r04/c3_unit_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r04/c3_unit_tests.txt
tail -5 gpurun_out/r04/c3_unit_tests.txt
timeout 300 python profiles/ktrace_probe.py > gpurun_out/r04/ktrace_1b_after.txt 2>&1; echo "rc=$?" >> gpurun_out/r04/ktrace_1b_after.txt
grep -E "^\[slot|===" gpurun_out/r04/ktrace_1b_after.txt | head -40
timeout 200 python profiles/draft_probe.py 6 300 > gpurun_out/r04/draft_probe.json 2> gpurun_out/r04/draft_probe.err; cat gpurun_out/r04/draft_probe.json
timeout 300 python profiles/async_probe.py > gpurun_out/r04/async_probe.json 2> gpurun_out/r04/async_probe.err; cat gpurun_out/r04/async_probe.json
timeout 900 python -m pytest tests/test_reference_replays_gpu.py -x -q -m gpu -s -k "full_size_async" > gpurun_out/r04/lockstep_async.txt 2>&1; echo "rc=$?" >> gpurun_out/r04/lockstep_async.txt
grep -E "full size|passed|failed|Error|rc=" gpurun_out/r04/lockstep_async.txt | head
timeout 300 python bench.py --workload c2 --steps 40 --no-cpu-baseline > gpurun_out/r04/bench_c2.json 2> gpurun_out/r04/bench_c2.err; echo "c2 rc=$?"
python - <<'P'
import json
try:
    d = json.load(open("gpurun_out/r04/bench_c2.json"))
    print("c2", d["value"], d["ms_per_step"], d.get("step_roofline_frac"), d.get("ttft_p50_ms"), d.get("mean_accepted_len"))
    for k, v in d["roofline"]["per_kind"].items():
        if k.startswith("draft"): print(k, v)
except Exception as e:
    print("c2 no line:", e)
P
