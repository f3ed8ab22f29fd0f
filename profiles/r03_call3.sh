# Round-3 GPU call 3: re-validated argmax tails, cache lookup, N = 4/5/8 bench tests, probes, c4 / c2 bench lines, c4 kernel trace
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03
mkdir -p $O
timeout 300 python -m pytest tests/test_hip_argmax_parts.py tests/test_hip_ops.py tests/test_custom_ar_gpu.py tests/test_tp_one_gpu.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -30 > $O/t3_ops.txt
timeout 200 python profiles/micro/draft_probe.py > $O/draft_probe2.txt 2> $O/draft_probe2.err
timeout 200 python profiles/async_probe.py llama-3.1-8b > $O/async_probe2.json 2> $O/async_probe2.err
timeout 300 python profiles/tp_shard_probe.py 8 > $O/tp_shard_per_kind2.txt 2> $O/tp_shard2.err
timeout 400 python profiles/tune_gemm.py 70b_tp4 70b_tp8 > $O/tune_tp.json 2> $O/tune_tp.txt
timeout 900 python -m pytest tests/test_bench_gpu.py -m gpu -q --no-header -p no:cacheprovider -s 2>&1 | tail -40 > $O/t3_bench.txt
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_reference_replays_gpu.py tests/test_engine_temperature_gpu.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -30 > $O/t3_engine.txt
timeout 500 python bench.py --steps 20 --warmup 3 > $O/bench_c4_n1.json 2> $O/bench_c4.err
timeout 300 python bench.py --steps 20 --warmup 3 --workload c2 > $O/bench_c2_n1.json 2> $O/bench_c2.err
P="--no-cpu-baseline --ttft-samples 6 --ref-seqs 0 --no-roofline"
timeout 500 rocprofv3 --kernel-trace --stats -d $O/kt_c4 -- python bench.py --steps 20 --warmup 3 $P > $O/kt_c4.json 2> /dev/null
DB="$(ls $O/kt_c4/*/*.db | tail -1)"
python profiles/kstats.py "$DB" $O/c4_kernel_stats.csv > $O/c4_kernel_stats.txt 2>&1
python profiles/prefill_timeline.py "$DB" > $O/c4_prefill_timeline.txt 2>&1
rm -rf $O/kt_c4
for f in t3_ops t3_bench t3_engine; do echo "== $f"; tail -n 4 $O/$f.txt; done
