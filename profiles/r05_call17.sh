#!/bin/bash
# Round 5, call 17: rocprofv3 --kernel-trace --stats summaries of the c2 and c5t bench commands on the final code
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05
mkdir -p $O
P="--no-cpu-baseline --ttft-samples 4 --ref-seqs 0 --no-roofline --no-random-pair"
for wl in c2 c5t; do
  timeout 500 rocprofv3 --kernel-trace --stats -d $O/kt_$wl -- python bench.py --workload $wl --steps 20 --warmup 3 $P > $O/kt_$wl.json 2> /dev/null
  DB="$(ls $O/kt_$wl/*/*.db | tail -1)"
  python profiles/kstats.py "$DB" $O/${wl}_kernel_stats.csv > $O/${wl}_kernel_stats.txt 2>&1
  rm -rf $O/kt_$wl
  head -8 $O/${wl}_kernel_stats.txt
done
