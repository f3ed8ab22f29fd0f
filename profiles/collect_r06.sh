#!/bin/bash
# Round-6 evidence, run on the GPU box from the repo root:  bash profiles/collect_r06.sh [trace|pmc|overlap]
#   trace:   rocprofv3 --kernel-trace --stats of the default bench command (c4) -> per-kernel summary + prefill timeline
#   overlap: the same trace, analysed per stream (profiles/overlap_timeline.py): how the co-located draft round shares the GPU with the verify
#   pmc:     SEPARATE counter passes on c4 (FETCH_SIZE, WRITE_SIZE, MfmaUtil / VALUBusy).  PMC passes never carry a trace domain.
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06
mkdir -p $O
what="${1:-trace}"
if [ "$what" = "trace" ] || [ "$what" = "overlap" ]; then
  P="--no-cpu-baseline --ttft-samples 6 --ref-seqs 0 --no-roofline --no-random-pair"
  timeout 500 rocprofv3 --kernel-trace --stats -d $O/kt_c4 -- python bench.py --steps 20 --warmup 3 $P > $O/kt_c4.json 2> /dev/null
  DB="$(ls $O/kt_c4/*/*.db | tail -1)"
  python profiles/kstats.py "$DB" $O/c4_kernel_stats.csv > $O/c4_kernel_stats.txt 2>&1
  python profiles/prefill_timeline.py "$DB" > $O/c4_prefill_timeline.txt 2>&1
  python profiles/overlap_timeline.py "$DB" 0.5 > $O/c4_overlap_timeline.txt 2>&1
  rm -rf $O/kt_c4
else
  Q="--workload c4 --steps 4 --no-random-pair --warmup 1 --no-cpu-baseline --no-roofline --ttft-samples 1 --ref-seqs 0"
  for c in FETCH_SIZE WRITE_SIZE "MfmaUtil VALUBusy"; do
    tag=$(echo $c | tr ' ' '_')
    timeout 420 rocprofv3 --pmc $c -d $O/pmc_$tag -- python bench.py $Q > /dev/null 2> $O/pmc_$tag.err
    python profiles/pmcstats.py "$(ls $O/pmc_$tag/*/*.db | tail -1)" > $O/pmc_$tag.csv 2>&1
    rm -rf $O/pmc_$tag/
  done
fi
ls -la $O | tail -n 12
