# Round-3 GPU call 5: N = 4 / 5 / 8 bench launches on one shared GPU (each under its own timeout), engine tests after the
# early-stream / fused-attention changes, draft probe small-op rows, c4 TTFT
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03
mkdir -p $O
rm -f $O/bench_tiny_rc.txt
for n in 4 5 8; do
  t0=$(date +%s)
  SSD_DIST_BACKEND=gloo SSD_LOCAL_DEVICE=0 SSD_AR_DEVICE=0 SSD_BENCH_CPU_SECONDS=1 HSA_ENABLE_IPC_MODE_LEGACY=0 \
    timeout 200 python bench.py --gpus $n --steps 6 --warmup 2 --ttft-samples 2 --workload tiny-async --ref-seqs 1 --ref-output-len 64 \
    > $O/bench_tiny_n$n.json 2> $O/bench_tiny_n$n.err
  echo "N=$n rc=$? wall=$(( $(date +%s) - t0 ))s" >> $O/bench_tiny_rc.txt
  grep -v "amdgpu.ids\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" $O/bench_tiny_n$n.err | tail -n 6 >> $O/bench_tiny_rc.txt
done
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_eagle_gpu.py tests/test_hip_attn_oproj.py tests/test_engine_temperature_gpu.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -30 > $O/t5_engine.txt
timeout 200 python profiles/micro/draft_probe.py > $O/draft_probe4.txt 2> $O/draft_probe4.err
timeout 400 python bench.py --steps 10 --warmup 3 --ref-seqs 1 --ref-output-len 128 > $O/bench_c4_quick.json 2> $O/bench_c4_quick.err
cat $O/bench_tiny_rc.txt; grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|destroy_process" $O/t5_engine.txt | tail -n 5
