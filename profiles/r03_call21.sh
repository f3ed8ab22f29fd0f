#!/bin/bash
# 40-token prefill + verify forwards at the real 70B / Qwen3-32B layer shapes with the QKV slabs consumed by the RoPE kernel
mkdir -p gpurun_out/r03
timeout 42 python -m pytest tests/test_real_shapes_gpu.py -x -q -m gpu -k "decoder_layer and (70b or qwen3)" > gpurun_out/r03/rope_parts_real_shapes.txt 2>&1
echo "rc=$?" >> gpurun_out/r03/rope_parts_real_shapes.txt
tail -3 gpurun_out/r03/rope_parts_real_shapes.txt
