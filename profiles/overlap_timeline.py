"""Round 6: how the co-located draft round shares the GPU with the target's verify in bench.py's c4 step, from a rocprofv3
--kernel-trace rocpd database.  Kernels are split by the HIP stream / queue they ran on (the verify graph replays on the
engine's stream, the draft server on its side stream); for the steady-state tail of the run it prints, per speculation step:

  * the verify's span (first to last target kernel), the sum of its kernel durations, the sum of the gaps between its consecutive
    kernels -- and how much of those gaps is covered by draft kernels (the target waits for CUs the draft holds) vs empty;
  * per target kernel kind: average duration when a draft kernel overlaps it vs when none does (is the stream being slowed, or only
    delayed?);
  * the draft kernels' total duration, the part of it that lies inside target kernels (hidden) and outside (exposed).

    python profiles/overlap_timeline.py <dir-or-db> [tail_fraction]"""
import glob
import os
import sqlite3
import sys
from bisect import bisect_left, bisect_right

p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*.db"), recursive=True))[-1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
c = sqlite3.connect(p)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print("columns of `kernels`:", cols)
key = next((k for k in ("stream_id", "stream", "queue_id", "queue") if k in cols), None)
rows = list(c.execute(f"select start, end, name{', ' + key if key else ''} from kernels order by start"))
rows = rows[int(len(rows) * (1 - frac)):]
if key is None:
    print("no stream / queue column: cannot split the streams")
    sys.exit(0)
by = {}
for r in rows:
    by.setdefault(r[3], []).append(r)
print({k: len(v) for k, v in by.items()})
# the target's stream is the one that carries the big gate_up GEMM (gemm_wf_kernel<1, 4, 1>, > 100 us)
def is_big(r):
    return "gemm_wf_kernel" in r[2] and r[1] - r[0] > 100_000
tkey = max(by, key=lambda k: sum(1 for r in by[k] if is_big(r)))
T = by[tkey]
D = sorted((r for k, v in by.items() if k != tkey for r in v), key=lambda r: r[0])
print(f"target stream {tkey}: {len(T)} dispatches; other streams: {len(D)} dispatches")
ds, de = [r[0] for r in D], [r[1] for r in D]


def covered(a, b):
    """ns of [a, b) covered by at least one draft kernel (draft kernels on one stream do not overlap each other)."""
    if b <= a or not D:
        return 0
    i = max(0, bisect_right(de, a) - 1)
    tot = 0
    while i < len(D) and ds[i] < b:
        lo, hi = max(a, ds[i]), min(b, de[i])
        if hi > lo:
            tot += hi - lo
        i += 1
    return tot


# split the target stream into verifies: a gap > 1 ms between consecutive target kernels = the host round trip between steps
steps, cur = [], [T[0]]
for a, b in zip(T, T[1:]):
    if b[0] - a[1] > 1_000_000 or (b[0] - a[1] > 200_000 and covered(a[1], b[0]) == 0):
        steps.append(cur)
        cur = []
    cur.append(b)
steps.append(cur)
steps = [s for s in steps if len(s) > 400]          # a 70B verify graph is ~560 launches
print(f"{len(steps)} verifies in the window")
tot = dict(span=0, busy=0, gaps=0, gaps_cov=0, n=0)
kind = {}
for s in steps:
    span = s[-1][1] - s[0][0]
    busy = sum(r[1] - r[0] for r in s)
    gaps = [(a[1], b[0]) for a, b in zip(s, s[1:]) if b[0] > a[1]]
    g = sum(b - a for a, b in gaps)
    gc = sum(covered(a, b) for a, b in gaps)
    tot["span"] += span; tot["busy"] += busy; tot["gaps"] += g; tot["gaps_cov"] += gc; tot["n"] += 1
    for r in s:
        ov = covered(r[0], r[1])
        k = kind.setdefault(r[2][:60], dict(n_ov=0, t_ov=0, n_no=0, t_no=0, ov=0))
        if ov > 0.05 * (r[1] - r[0]):
            k["n_ov"] += 1; k["t_ov"] += r[1] - r[0]; k["ov"] += ov
        else:
            k["n_no"] += 1; k["t_no"] += r[1] - r[0]
n = max(1, tot["n"])
print(f"per verify: span {tot['span'] / n / 1e6:.3f} ms = kernels {tot['busy'] / n / 1e6:.3f} ms + gaps {tot['gaps'] / n / 1e6:.3f} ms "
      f"(of which {tot['gaps_cov'] / n / 1e6:.3f} ms while a draft kernel runs, {(tot['gaps'] - tot['gaps_cov']) / n / 1e6:.3f} ms empty)")
print("target kernel kind: avg us with a draft kernel beside it (n) | alone (n) | share of its time the draft overlaps")
for name, k in sorted(kind.items(), key=lambda kv: -(kv[1]["t_ov"] + kv[1]["t_no"]))[:10]:
    a = k["t_ov"] / k["n_ov"] / 1e3 if k["n_ov"] else float("nan")
    b = k["t_no"] / k["n_no"] / 1e3 if k["n_no"] else float("nan")
    print(f"  {name:60s} {a:8.2f} ({k['n_ov']:5d}) | {b:8.2f} ({k['n_no']:5d}) | {k['ov'] / max(1, k['t_ov']):.2f}")
# draft side: inside the verify windows
dt = hidden = 0
ts, te = [r[0] for r in T], [r[1] for r in T]
for r in D:
    if not steps or r[0] < steps[0][0][0] or r[1] > steps[-1][-1][1] + 10_000_000:
        continue
    dt += r[1] - r[0]
    i = max(0, bisect_right(te, r[0]) - 1)
    while i < len(T) and ts[i] < r[1]:
        lo, hi = max(r[0], ts[i]), min(r[1], te[i])
        if hi > lo:
            hidden += hi - lo
        i += 1
print(f"draft kernels in the window: {dt / n / 1e6:.3f} ms per step, {hidden / n / 1e6:.3f} ms of it while a target kernel runs, "
      f"{(dt - hidden) / n / 1e6:.3f} ms with the target's stream idle")
