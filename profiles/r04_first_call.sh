#!/bin/bash
# First GPU call of the next round (prepared at the end of round 3, when the GPU budget was spent):
#  1. the BPRE form of the prefill GEMM: bit-identity + timing probe, then the prefill tests and the c4 line with SSD_PF_BPRE=2 (or 1, whichever the probe favours)
#  2. bench.py --workload c4e on the constructed EAGLE pair (CPU-validated only so far)
mkdir -p gpurun_out/r04
timeout 120 python profiles/micro/prefill_gemm_probe_bpre.py > gpurun_out/r04/pf_probe_bpre.txt 2>&1; echo "rc=$?" >> gpurun_out/r04/pf_probe_bpre.txt
tail -14 gpurun_out/r04/pf_probe_bpre.txt
SSD_PF_BPRE=2 timeout 300 python -m pytest tests/test_hip_ops.py tests/test_real_shapes_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "prefill or pf or decoder_layer or chunk" > gpurun_out/r04/bpre_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r04/bpre_tests.txt
tail -3 gpurun_out/r04/bpre_tests.txt
SSD_PF_BPRE=2 timeout 200 python bench.py > gpurun_out/r04/bench_c4_bpre.json 2> gpurun_out/r04/bench_c4_bpre.err; echo "bench rc=$?"
timeout 200 python bench.py > gpurun_out/r04/bench_c4_plain.json 2> gpurun_out/r04/bench_c4_plain.err
timeout 400 python bench.py --workload c4e > gpurun_out/r04/bench_c4e_constructed.json 2> gpurun_out/r04/bench_c4e_constructed.err; echo "c4e rc=$?"
python - <<'P'
import json
for f in ("bench_c4_bpre", "bench_c4_plain", "bench_c4e_constructed"):
    try:
        d = json.load(open(f"gpurun_out/r04/{f}.json"))
        print(f, d["value"], d["ms_per_step"], d.get("ttft_p50_ms"), d.get("mean_accepted_len"), d.get("cache_hit_rate"))
    except Exception as e:
        print(f, "no line:", e)
P
