#!/bin/bash
# Round 5, call 1: first run of the resident M-row layer segment: parity tests, then the A/B probe
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_hip_tree_segment.py -x -q -m gpu -s > gpurun_out/r05/c1_tree_seg_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r05/c1_tree_seg_tests.txt
grep -E "passed|failed|error|rc=" gpurun_out/r05/c1_tree_seg_tests.txt | tail -5
timeout 300 python profiles/tree_seg_probe.py 300 > gpurun_out/r05/c1_tree_seg_probe.txt 2> gpurun_out/r05/c1_tree_seg_probe.err; cat gpurun_out/r05/c1_tree_seg_probe.txt; tail -3 gpurun_out/r05/c1_tree_seg_probe.err
