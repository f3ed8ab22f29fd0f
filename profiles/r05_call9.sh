#!/bin/bash
# Round 5, call 9: the remaining new tests, the one-off full-depth 70B parity run, then the whole GPU suite
mkdir -p gpurun_out/r05
free -g | head -2
timeout 300 python -m pytest tests/test_hip_ops.py -q -m gpu -k "stand_alone or fork" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_real_shapes_gpu.py -q -m gpu -s -k "eight_layer" > gpurun_out/r05/c9_70b_cut.txt 2>&1; grep -E "70B x|passed|failed|Error" gpurun_out/r05/c9_70b_cut.txt | tail -5
timeout 900 python -m pytest tests/test_reference_replays_gpu.py -q -m gpu -s -k "undamped" > gpurun_out/r05/c9_lockstep_undamped.txt 2>&1; grep -E "full size|passed|failed|Error|assert" gpurun_out/r05/c9_lockstep_undamped.txt | tail -6
SSD_FULL_70B=1 timeout 1500 python -m pytest tests/test_real_shapes_gpu.py -q -m gpu -s -k "full_depth_70b" > gpurun_out/r05/c9_full_depth_70b.txt 2>&1; grep -E "70B x|last layer|passed|failed|skipped|Error" gpurun_out/r05/c9_full_depth_70b.txt | tail -6
timeout 1500 python -m pytest tests/ -q -m gpu > gpurun_out/r05/c9_gpu_suite_full.txt 2>&1; tail -5 gpurun_out/r05/c9_gpu_suite_full.txt
