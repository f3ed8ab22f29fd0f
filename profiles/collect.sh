#!/bin/bash
# Round evidence, run on the GPU box from the repo root:  bash profiles/collect.sh
#   bench lines (c3 = default workload, c2), rocprofv3 kernel-trace summaries of the same commands, and the two separate
#   PMC passes (FETCH_SIZE, WRITE_SIZE -- never combined with a trace domain) for the GEMM family's HBM traffic.
# Everything lands in gpurun_out/evidence/; the summaries worth judging are copied into profiles/ by hand.
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/evidence
mkdir -p $O
python bench.py > $O/bench_c3_n1.json 2> $O/bench_c3.err
python bench.py --workload c2 > $O/bench_c2_n1.json 2> $O/bench_c2.err
rocprofv3 --kernel-trace --stats -d $O/kt_c3 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --ttft-samples 1 > $O/kt_c3.json 2> /dev/null
python profiles/kstats.py "$(ls $O/kt_c3/*/*.db | tail -1)" $O/c3_kernel_stats.csv > $O/c3_kernel_stats.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt_c2 -- python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline --ttft-samples 1 > $O/kt_c2.json 2> /dev/null
python profiles/kstats.py "$(ls $O/kt_c2/*/*.db | tail -1)" $O/c2_kernel_stats.csv > $O/c2_kernel_stats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $O/pmc_$c -- python bench.py --workload c2 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --ttft-samples 1 > /dev/null 2>&1
  python profiles/pmcstats.py "$(ls $O/pmc_$c/*/*.db | tail -1)" > $O/pmc_$c.csv 2>&1
done
rm -rf $O/kt_c3 $O/kt_c2 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
ls -la $O
