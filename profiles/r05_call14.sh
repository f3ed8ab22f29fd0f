#!/bin/bash
# Round 5, call 14: rocprofv3 --kernel-trace --stats of the default bench command on the final code (the M-row segment stays off beside the co-located verify)
mkdir -p gpurun_out/r05
bash profiles/collect_r05.sh trace > gpurun_out/r05/c14_collect_trace.log 2>&1; head -12 gpurun_out/r05/c4_kernel_stats.txt
