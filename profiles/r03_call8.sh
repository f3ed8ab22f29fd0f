# Round-3 GPU call 8: every single-process GPU test (the three multi-process files run in call 9)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03
mkdir -p $O
timeout 1300 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --timeout=420 \
  --deselect tests/test_bench_gpu.py --deselect tests/test_custom_ar_gpu.py --deselect tests/test_tp_one_gpu.py 2>&1 \
  | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|destroy_process" | tail -40 > $O/t8_single_process.txt
tail -n 6 $O/t8_single_process.txt
