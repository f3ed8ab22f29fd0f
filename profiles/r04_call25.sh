#!/bin/bash
# the full-size engine tests with the chain segment on (before flipping the default)
mkdir -p gpurun_out/r04
SSD_CHAIN_SEG=1 timeout 1200 python -m pytest tests/test_real_shapes_gpu.py tests/test_reference_replays_gpu.py -q -m gpu -k "full or lockstep or 1b" > gpurun_out/r04/chain_fullsize_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r04/chain_fullsize_tests.txt
tail -12 gpurun_out/r04/chain_fullsize_tests.txt
