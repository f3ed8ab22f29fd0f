#!/bin/bash
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_custom_ar_gpu.py tests/test_tp_one_gpu.py -x -q -m gpu > gpurun_out/r04/c9_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r04/c9_tests.txt
tail -3 gpurun_out/r04/c9_tests.txt
timeout 400 python profiles/tp_shard_probe.py 8 > gpurun_out/r04/tp_shard_per_kind2.txt 2> gpurun_out/r04/tp_shard2.err; echo "tp rc=$?"
grep -E "all-reduce|PROJECTED|xGMI" gpurun_out/r04/tp_shard_per_kind2.txt
SSD_AR_PROTO=flag timeout 400 python profiles/tp_shard_probe.py 8 > gpurun_out/r04/tp_shard_per_kind2_flag.txt 2> gpurun_out/r04/tp_shard2f.err
grep -E "all-reduce|PROJECTED" gpurun_out/r04/tp_shard_per_kind2_flag.txt
