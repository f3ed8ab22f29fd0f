mkdir -p gpurun_out/r06ab
for rep in 1 2 3; do for m in 1 3 4 0; do
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --no-random-pair --ref-seqs 0 --ttft-samples 2 --no-roofline --tune-deep $m > gpurun_out/r06ab/c_${m}_$rep.json 2> gpurun_out/r06ab/c_${m}_$rep.err
  python -c "
import json; d=json.loads(open('gpurun_out/r06ab/c_${m}_$rep.json').read().strip().splitlines()[-1]); print('mode', $m, 'rep', $rep, d['ms_per_step'], d['mean_accepted_len'], d['cache_hit_rate'])"
done; done
