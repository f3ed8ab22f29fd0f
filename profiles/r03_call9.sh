# Round-3 GPU call 9: argmax epilogue without its second barrier (tests + head timing), then the three multi-process test files
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03
mkdir -p $O
timeout 200 python -m pytest tests/test_hip_argmax_parts.py tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider --timeout=200 -k "argmax or tiny_llama or engine_sync or engine_autoregressive or prefill_shape" 2>&1 | tail -5 > $O/t9_argmax.txt
timeout 200 python profiles/micro/draft_probe.py > $O/draft_probe5.txt 2> $O/draft_probe5.err
timeout 700 python -m pytest tests/test_bench_gpu.py tests/test_custom_ar_gpu.py tests/test_tp_one_gpu.py -m gpu -q --no-header -p no:cacheprovider --timeout=300 2>&1 \
  | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|destroy_process" | tail -25 > $O/t9_multi_process.txt
tail -n 3 $O/t9_argmax.txt $O/t9_multi_process.txt; grep "WHOLE\|lm head\|argmax" $O/draft_probe5.txt
