#!/bin/bash
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_hip_chain.py -x -q -m gpu -s > gpurun_out/r04/chain_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r04/chain_tests.txt
tail -30 gpurun_out/r04/chain_tests.txt
