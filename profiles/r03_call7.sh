# Round-3 GPU call 7 (single-process only): the c4 bench line with both TTFT definitions, then the engine tests touched since call 4
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03
mkdir -p $O
timeout 500 python bench.py > $O/bench_c4_n1.json 2> $O/bench_c4.err
timeout 500 python -m pytest tests/test_model_gpu.py tests/test_eagle_gpu.py -m gpu -q --no-header -p no:cacheprovider --timeout=300 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|destroy_process" | tail -15 > $O/t7_engine.txt
tail -c 600 $O/bench_c4_n1.json; echo; tail -n 4 $O/t7_engine.txt
