#!/bin/bash
# Round 4, final validation on the final code: the driver's own round-end commands
mkdir -p gpurun_out/r04
( time timeout 2400 python -m pytest tests/ -q -m gpu --durations=12 ) > gpurun_out/r04/gpu_suite_full.txt 2>&1; echo "suite rc=$?" | tee -a gpurun_out/r04/gpu_suite_full.txt
tail -25 gpurun_out/r04/gpu_suite_full.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/smoke_final.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r04/smoke_final.txt
timeout 600 python bench.py > gpurun_out/r04/bench_c4_n1.json 2> gpurun_out/r04/bench_c4_n1.err; echo "bench rc=$?"
python - <<'P'
import json
try:
    d = json.load(open("gpurun_out/r04/bench_c4_n1.json"))
    print("c4", d["value"], d["ms_per_step"], d.get("step_roofline_frac"), d.get("step_roofline_frac_target_only"), "ttft", d.get("ttft_p50_ms"), d.get("ttft_first_sighting_ms"), d.get("mean_accepted_len"), d.get("value_reference_protocol"), d["roofline"]["frac"], d["cpu_baseline"]["value"])
except Exception as e:
    print("c4 no line:", e)
P
