# Round-3 GPU call 14: prefill attention waves per workgroup (TTFT only)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03; mkdir -p $O
Q="--steps 3 --warmup 1 --ref-seqs 0 --no-roofline --no-cpu-baseline"
for wv in 0 4 8; do
  SSD_ATTN_PREFILL_WAVES=$wv timeout 400 python bench.py $Q > $O/ttft_waves$wv.json 2> /dev/null
  python - <<PY
import json
d = json.load(open("$O/ttft_waves$wv.json"))
print("prefill attention waves", $wv, "ttft_p50_ms", d["ttft_p50_ms"], d["ttft_samples_ms"])
PY
done
