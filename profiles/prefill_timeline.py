"""Where one prefill's time goes (TTFT attribution), from a rocprofv3 --kernel-trace rocpd database of
`python bench.py --steps 2 --warmup 1 --ttft-samples 6 --ref-seqs 0 --no-roofline --no-cpu-baseline`:
prefills are the bursts of gemm_pf_kernel dispatches; for the LAST burst (a hipGraph replay, steady state) print the span
from the first kernel of the burst's forward to its last, GPU-busy time, gaps, and the per-kernel totals.
    python profiles/prefill_timeline.py <dir-or-db>"""
import glob
import os
import sqlite3
import sys

p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*.db"), recursive=True))[-1]
c = sqlite3.connect(p)
rows = list(c.execute("select start, end, name from kernels order by start"))
pf = [i for i, r in enumerate(rows) if "gemm_pf_kernel" in r[2]]
assert pf, "no prefill GEMM dispatches in the trace"
# bursts of prefill GEMMs: split where two consecutive ones are > 3 ms apart
bursts, cur = [], [pf[0]]
for a, b in zip(pf, pf[1:]):
    if rows[b][0] - rows[a][1] > 3_000_000:
        bursts.append(cur)
        cur = []
    cur.append(b)
bursts.append(cur)
print(f"{len(bursts)} prefills in the trace ({[len(b) for b in bursts]} prefill-GEMM dispatches each)")
for which in (-1,):
    b = bursts[which]
    # extend the window backwards / forwards over the dispatches that belong to the same forward: gaps < 200 us
    lo, hi = b[0], b[-1]
    while lo > 0 and rows[lo][0] - rows[lo - 1][1] < 200_000:
        lo -= 1
    while hi + 1 < len(rows) and rows[hi + 1][0] - rows[hi][1] < 200_000:
        hi += 1
    win = rows[lo:hi + 1]
    span = win[-1][1] - win[0][0]
    busy = sum(e - s for s, e, _ in win)
    gaps = [win[i + 1][0] - win[i][1] for i in range(len(win) - 1)]
    print(f"last prefill: {len(win)} dispatches, span {span / 1e6:.3f} ms, GPU busy {busy / 1e6:.3f} ms ({100 * busy / span:.1f} %), "
          f"gaps {sum(g for g in gaps if g > 0) / 1e6:.3f} ms (max {max(gaps) / 1e3:.1f} us)")
    agg = {}
    for s, e, n in win:
        a = agg.setdefault(n, [0, 0])
        a[0] += 1
        a[1] += e - s
    for n, (k, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
        print(f"  {n[:84]:84s} n={k:5d} avg={t / k / 1e3:8.2f}us tot={t / 1e6:8.3f}ms {100 * t / span:5.1f}%")
