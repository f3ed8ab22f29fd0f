#!/bin/bash
# rocprofv3 kernel-trace summary of c2 (8B + 1B, sync k = 6) with the resident chain segment, + a FETCH_SIZE pass for it
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04
mkdir -p $O
P="--workload c2 --no-cpu-baseline --ttft-samples 2 --ref-seqs 0 --no-roofline"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt_c2 -- python bench.py --steps 20 --warmup 3 $P > $O/kt_c2.json 2> /dev/null
DB="$(ls $O/kt_c2/*/*.db | tail -1)"
python profiles/kstats.py "$DB" $O/c2_kernel_stats.csv > $O/c2_kernel_stats.txt 2>&1
rm -rf $O/kt_c2
head -30 $O/c2_kernel_stats.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_c2 -- python bench.py --steps 4 --warmup 1 $P > /dev/null 2> $O/pmc_c2.err
python profiles/pmcstats.py "$(ls $O/pmc_c2/*/*.db | tail -1)" > $O/pmc_c2_fetch.csv 2>&1
rm -rf $O/pmc_c2
grep -i "chain\|Name" $O/pmc_c2_fetch.csv | head -5
