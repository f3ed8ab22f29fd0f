#!/bin/bash
# Round-2 evidence, run on the GPU box from the repo root:  bash profiles/collect_r02.sh [quick]
#   bench lines (c4 = the metric's workload and bench.py's default, c3, c2), rocprofv3 kernel-trace summaries of the same
#   commands, and SEPARATE PMC passes on the HEADLINE models (70B + 1B): FETCH_SIZE, WRITE_SIZE (HBM traffic of the GEMM
#   family) and MFMA / VALU utilisation.  PMC passes never carry a trace domain.  Everything lands in gpurun_out/evidence2/;
#   the summaries worth judging are copied into profiles/ by hand.
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/evidence2
mkdir -p $O
S="--steps 20 --warmup 3"
python bench.py $S > $O/bench_c4_n1.json 2> $O/bench_c4.err
python bench.py $S --workload c3 > $O/bench_c3_n1.json 2> $O/bench_c3.err
python bench.py $S --workload c2 > $O/bench_c2_n1.json 2> $O/bench_c2.err
[ "${1:-}" = "quick" ] && exit 0
P="--no-cpu-baseline --ttft-samples 1 --ref-seqs 0"
rocprofv3 --kernel-trace --stats -d $O/kt_c4 -- python bench.py $S $P > $O/kt_c4.json 2> /dev/null
python profiles/kstats.py "$(ls $O/kt_c4/*/*.db | tail -1)" $O/c4_kernel_stats.csv > $O/c4_kernel_stats.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt_c3 -- python bench.py $S $P --workload c3 > $O/kt_c3.json 2> /dev/null
python profiles/kstats.py "$(ls $O/kt_c3/*/*.db | tail -1)" $O/c3_kernel_stats.csv > $O/c3_kernel_stats.txt 2>&1
rocprofv3 -L > $O/counters_available.txt 2>&1
Q="--workload c3 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --ttft-samples 1 --ref-seqs 0"
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "MfmaUtil VALUBusy"; do
  tag=$(echo $c | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $c -d $O/pmc_$tag -- python bench.py $Q > /dev/null 2> $O/pmc_$tag.err
  python profiles/pmcstats.py "$(ls $O/pmc_$tag/*/*.db | tail -1)" > $O/pmc_$tag.csv 2>&1
done
rm -rf $O/kt_c4 $O/kt_c3 $O/pmc_*/
ls -la $O
