#!/bin/bash
# the driver's round-end commands on the final tree + one shuffled file order
mkdir -p gpurun_out/r06final
(time python -m pytest tests/ -x -q -m gpu) > gpurun_out/r06final/suite_plain.txt 2>&1; tail -4 gpurun_out/r06final/suite_plain.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06final/smoke.txt 2>&1; tail -1 gpurun_out/r06final/smoke.txt
python bench.py > gpurun_out/r06final/bench_c4.json 2> gpurun_out/r06final/bench_c4.err
python -c "
import json; d=json.loads(open('gpurun_out/r06final/bench_c4.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['ttft_p50_ms'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['cpu_baseline']['value'])"
(time SSD_TEST_SHUFFLE_SEED=4 python -m pytest tests/ -x -q -m gpu) > gpurun_out/r06final/suite_shuffle4.txt 2>&1; tail -4 gpurun_out/r06final/suite_shuffle4.txt
