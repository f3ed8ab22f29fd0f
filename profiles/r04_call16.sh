#!/bin/bash
mkdir -p gpurun_out/r04
for v in 0 1 2; do
  echo "== SSD_K3_LATE=$v"
  SSD_K3_LATE=$v timeout 200 python profiles/draft_probe.py 6 300 2>/dev/null
done
SSD_K3_LATE=2 timeout 300 python profiles/ktrace_probe.py > gpurun_out/r04/ktrace_1b_v7.txt 2>&1
sed -n 4,25p gpurun_out/r04/ktrace_1b_v7.txt
