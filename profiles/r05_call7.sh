#!/bin/bash
# Round 5, call 7: segment v6 = v5 + ONE agent acquire per edge and cached (plain) loads of the hand-off rows
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_hip_tree_segment.py -x -q -m gpu -s > gpurun_out/r05/c7_tree_seg_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r05/c7_tree_seg_tests.txt
grep -E "passed|failed|error|rc=" gpurun_out/r05/c7_tree_seg_tests.txt | tail -5
timeout 300 python profiles/tree_seg_probe.py 300 > gpurun_out/r05/c7_tree_seg_probe.txt 2> gpurun_out/r05/c7_tree_seg_probe.err; cat gpurun_out/r05/c7_tree_seg_probe.txt
SSD_TREE_SEG=1 timeout 300 python profiles/ktrace_probe.py > gpurun_out/r05/c7_ktrace_tree_seg.txt 2> gpurun_out/r05/c7_ktrace.err; echo "rc=$?" >> gpurun_out/r05/c7_ktrace_tree_seg.txt
grep -A11 "slot 14" gpurun_out/r05/c7_ktrace_tree_seg.txt | tail -13
