# Round-3 GPU call 4: fused attention + o_proj, EAGLE glue graph, TTFT attribution, N = 4 / 5 / 8 bench launches (debug)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03
mkdir -p $O
timeout 300 python -m pytest tests/test_hip_attn_oproj.py tests/test_hip_ops.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -30 > $O/t4_ops.txt
timeout 200 python profiles/micro/draft_probe.py > $O/draft_probe3.txt 2> $O/draft_probe3.err
timeout 200 python profiles/async_probe.py llama-3.1-8b > $O/async_probe3.json 2> $O/async_probe3.err
timeout 600 python -m pytest tests/test_eagle_gpu.py tests/test_model_gpu.py tests/test_reference_replays_gpu.py tests/test_hip_fused.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -30 > $O/t4_engine.txt
export SSD_DIST_BACKEND=gloo SSD_LOCAL_DEVICE=0 SSD_AR_DEVICE=0 SSD_BENCH_CPU_SECONDS=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 4 5 8; do
  /usr/bin/time -f "N=$n wall %e s" timeout 240 python bench.py --gpus $n --steps 6 --warmup 2 --ttft-samples 2 --workload tiny-async --ref-seqs 1 --ref-output-len 64 > $O/bench_tiny_n$n.json 2> $O/bench_tiny_n$n.err
  echo "N=$n rc=$?" >> $O/bench_tiny_rc.txt
  tail -n 3 $O/bench_tiny_n$n.err >> $O/bench_tiny_rc.txt
done
unset SSD_DIST_BACKEND SSD_LOCAL_DEVICE SSD_AR_DEVICE SSD_BENCH_CPU_SECONDS
timeout 400 python profiles/ttft_probe.py c4 > $O/ttft_probe.txt 2> $O/ttft_probe.err
for f in t4_ops t4_engine bench_tiny_rc; do echo "== $f"; tail -n 6 $O/$f.txt; done
