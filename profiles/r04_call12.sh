#!/bin/bash
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_reference_replays_gpu.py -x -q -m gpu -s -k "eagle3_llama8b" > gpurun_out/r04/lockstep_eagle.txt 2>&1; echo "rc=$?" >> gpurun_out/r04/lockstep_eagle.txt
grep -E "full size|passed|failed|Error|rc=" gpurun_out/r04/lockstep_eagle.txt | head
timeout 600 python -m pytest tests/test_custom_ar_gpu.py -x -q -m gpu > gpurun_out/r04/ar_stress.txt 2>&1; echo "rc=$?" >> gpurun_out/r04/ar_stress.txt; tail -3 gpurun_out/r04/ar_stress.txt
for w in c2 c5t c4e; do
  timeout 400 python bench.py --workload $w > gpurun_out/r04/bench_${w}_n1.json 2> gpurun_out/r04/bench_${w}_n1.err; echo "$w rc=$?"
done
python - <<'P'
import json
for w in ("c2", "c5t", "c4e"):
    try:
        d = json.load(open(f"gpurun_out/r04/bench_{w}_n1.json"))
        print(w, d["value"], d["ms_per_step"], d.get("step_roofline_frac"), "ttft", d.get("ttft_p50_ms"), d.get("mean_accepted_len"), d.get("cache_hit_rate"), d.get("value_reference_protocol"))
    except Exception as e:
        print(w, "no line:", e)
P
bash profiles/collect_r04.sh pmc > gpurun_out/r04/collect_pmc.log 2>&1; tail -5 gpurun_out/r04/collect_pmc.log
python profiles/make_traffic.py gpurun_out/r04 r04 > gpurun_out/r04/make_traffic.log 2>&1; cp profiles/traffic_r04.json gpurun_out/r04/ 2>/dev/null; tail -4 gpurun_out/r04/make_traffic.log
