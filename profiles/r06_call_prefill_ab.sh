#!/bin/bash
# same-box A/B of the prefill's small kernels: TTFT of the c4 line with the library before / after the rope_store_kv_parts change
mkdir -p gpurun_out/r06w
L=ssd_amd/_lib
cp $L/libssdhip.so $L/new.so.keep
for g in "128 32 8 128" "128 32 8 64" "128 16 8 128" "512 64 8 128" "64 64 8 128" "100 64 8 128"; do
  timeout 200 python profiles/micro/prefill_small_probe.py $g 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r06w/attn_geoms.txt
cat gpurun_out/r06w/attn_geoms.txt
for rep in 1 2; do
  for v in old new; do
    cp $L/lib${v/old/ssdhip_old}.so.keep $L/libssdhip.so 2>/dev/null || cp $L/new.so.keep $L/libssdhip.so
    timeout 400 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r06w/bench_${v}_$rep.json 2> gpurun_out/r06w/bench_${v}_$rep.err
    python -c "
import json; d=json.loads(open('gpurun_out/r06w/bench_${v}_$rep.json').read().strip().splitlines()[-1]); print('$v', $rep, d['ms_per_step'], d['ttft_p50_ms'], d['ttft_samples_ms'])"
  done
done
cp $L/new.so.keep $L/libssdhip.so
