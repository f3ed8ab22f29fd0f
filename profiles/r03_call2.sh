cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03
mkdir -p $O
timeout 300 python -m pytest tests/test_hip_argmax_parts.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -40 > $O/t_argmax.txt
timeout 240 python profiles/micro/draft_probe.py > $O/draft_probe.txt 2> $O/draft_probe.err
timeout 240 python profiles/async_probe.py llama-3.1-8b > $O/async_probe.json 2> $O/async_probe.err
timeout 300 python profiles/tp_shard_probe.py 8 > $O/tp_shard_per_kind.txt 2> $O/tp_shard.err
timeout 600 python -m pytest tests/test_reference_replays_gpu.py -m gpu -q --no-header -p no:cacheprovider -s 2>&1 | tail -60 > $O/t_replays.txt
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_model_gpu.py tests/test_hip_fused.py tests/test_real_shapes_gpu.py tests/test_eagle_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -40 > $O/t_core.txt
tail -3 $O/t_argmax.txt $O/t_replays.txt $O/t_core.txt
