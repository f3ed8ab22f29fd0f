mkdir -p gpurun_out/r06ab
SSD_FAT_DOWN=1 timeout 300 python -m pytest tests/test_real_shapes_gpu.py -q -m gpu -x -k "sixteen_layer" 2>&1 | tail -2
for rep in 1 2 3; do for m in 0 1; do
  SSD_FAT_DOWN=$m timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --no-random-pair --ref-seqs 0 --ttft-samples 2 --no-roofline > gpurun_out/r06ab/f_${m}_$rep.json 2> gpurun_out/r06ab/f_${m}_$rep.err
  python -c "
import json; d=json.loads(open('gpurun_out/r06ab/f_${m}_$rep.json').read().strip().splitlines()[-1]); print('fat_down', $m, 'rep', $rep, d['ms_per_step'], d['mean_accepted_len'], d['cache_hit_rate'])"
done; done
