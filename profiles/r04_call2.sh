#!/bin/bash
# Round 4, call 2: the new lock-step parity tests (full-size sync + async k=7 f=3 vs the oracle engine), the full-1B truth-relative
# verify-logits test, smoke() with its accepting leg, and the in-kernel ramp profile of the draft's kernels.
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_reference_replays_gpu.py -x -q -m gpu -s -k "full_size" > gpurun_out/r04/lockstep_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r04/lockstep_tests.txt
tail -12 gpurun_out/r04/lockstep_tests.txt
timeout 600 python -m pytest tests/test_real_shapes_gpu.py -x -q -m gpu -s -k "full_1b" > gpurun_out/r04/full1b_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r04/full1b_tests.txt
tail -6 gpurun_out/r04/full1b_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/smoke.txt 2>&1; echo "rc=$?" >> gpurun_out/r04/smoke.txt
tail -4 gpurun_out/r04/smoke.txt
timeout 300 python profiles/ktrace_probe.py > gpurun_out/r04/ktrace_1b.txt 2>&1; echo "rc=$?" >> gpurun_out/r04/ktrace_1b.txt
tail -5 gpurun_out/r04/ktrace_1b.txt
