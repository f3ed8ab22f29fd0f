#!/bin/bash
# Round 5, call 12: new entry-point tests, then the driver's own round-end commands on the current tree
mkdir -p gpurun_out/r05
timeout 300 python -m pytest tests/test_hip_ops.py -q -m gpu -k "call_site_names or stand_alone" 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05/c12_smoke.txt 2>&1; tail -2 gpurun_out/r05/c12_smoke.txt
timeout 1800 python -m pytest tests/ -q -m gpu > gpurun_out/r05/c12_gpu_suite_full.txt 2>&1; grep -E "passed|failed" gpurun_out/r05/c12_gpu_suite_full.txt | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/r05/c12_gpu_suite_full.txt | head
timeout 600 python bench.py > gpurun_out/r05/c12_bench_c4.json 2> gpurun_out/r05/c12_bench_c4.err; tail -c 400 gpurun_out/r05/c12_bench_c4.json | head -c 400; echo
