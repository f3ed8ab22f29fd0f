#!/bin/bash
# Round 5, call 5: segment v4 = v1 with the polls and the hand-off rows issued into an EMPTY load queue (weight prefetch behind them)
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_hip_tree_segment.py -x -q -m gpu -s > gpurun_out/r05/c5_tree_seg_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r05/c5_tree_seg_tests.txt
grep -E "passed|failed|error|rc=" gpurun_out/r05/c5_tree_seg_tests.txt | tail -5
timeout 300 python profiles/tree_seg_probe.py 300 > gpurun_out/r05/c5_tree_seg_probe.txt 2> gpurun_out/r05/c5_tree_seg_probe.err; cat gpurun_out/r05/c5_tree_seg_probe.txt
SSD_TREE_SEG=1 timeout 300 python profiles/ktrace_probe.py > gpurun_out/r05/c5_ktrace_tree_seg.txt 2> gpurun_out/r05/c5_ktrace.err; echo "rc=$?" >> gpurun_out/r05/c5_ktrace_tree_seg.txt
grep -A11 "slot 14" gpurun_out/r05/c5_ktrace_tree_seg.txt | tail -13
