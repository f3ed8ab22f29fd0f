# Round-3 GPU call 6a: the whole GPU suite (per-test timeout), draft probes, c4 / c2 bench lines
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --timeout=420 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|destroy_process" | tail -40 > $O/t6_all.txt
timeout 200 python profiles/micro/draft_probe.py > $O/draft_probe4.txt 2> $O/draft_probe4.err
timeout 200 python profiles/async_probe.py llama-3.1-8b > $O/async_probe4.json 2> $O/async_probe4.err
timeout 500 python bench.py > $O/bench_c4_n1.json 2> $O/bench_c4.err
timeout 300 python bench.py --workload c2 > $O/bench_c2_n1.json 2> $O/bench_c2.err
tail -n 5 $O/t6_all.txt
