#!/bin/bash
mkdir -p gpurun_out/r04
for v in 0 1; do
  SSD_CHAIN_SEG=$v timeout 400 python bench.py --workload c2 --no-cpu-baseline > gpurun_out/r04/bench_c2_chain_$v.json 2> gpurun_out/r04/bench_c2_chain_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04/bench_c2_chain_$v.json").read().strip().splitlines()[-1])
print("SSD_CHAIN_SEG=$v", d["value"], d["ms_per_step"], d["step_roofline_frac"], d["mean_accepted_len"], d.get("value_reference_protocol"), d["reference_protocol"]["stream_head"][:8])
PY
done
