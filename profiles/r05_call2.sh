#!/bin/bash
# Round 5, call 2: in-kernel timeline of the M-row segment (glue M = 8, tree step M = 24)
mkdir -p gpurun_out/r05
SSD_TREE_SEG=1 timeout 300 python profiles/ktrace_probe.py > gpurun_out/r05/c2_ktrace_tree_seg.txt 2> gpurun_out/r05/c2_ktrace.err; echo "rc=$?" >> gpurun_out/r05/c2_ktrace_tree_seg.txt
grep -A14 "slot 14\|slot 15" gpurun_out/r05/c2_ktrace_tree_seg.txt | head -80
