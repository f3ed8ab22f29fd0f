# Round-3 GPU call 11: smoke(), rocprofv3 kernel trace of the default bench command, separate PMC passes (FETCH_SIZE, WRITE_SIZE, MfmaUtil)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03
mkdir -p $O
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
bash profiles/collect_r03.sh trace > $O/collect_trace.log 2>&1
bash profiles/collect_r03.sh pmc > $O/collect_pmc.log 2>&1
tail -n 2 $O/smoke.txt; head -n 6 $O/c4_kernel_stats.txt | cut -c1-150; head -n 3 $O/c4_prefill_timeline.txt; ls -la $O | grep pmc
