"""Busy time vs launch gaps inside hipGraph replays, from a rocprofv3 rocpd kernel trace:
python profiles/ktimeline.py <dir-or-db> [tail_fraction]   (looks at the last fraction of dispatches = steady state)"""
import glob
import os
import sqlite3
import sys

p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*.db"), recursive=True))[-1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
c = sqlite3.connect(p)
rows = list(c.execute("select start, end, name from kernels order by start"))
rows = rows[int(len(rows) * (1 - frac)):]
busy = sum(e - s for s, e, _ in rows)
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
small = [g for g in gaps if g < 30000]
span = rows[-1][1] - rows[0][0]
print(f"{len(rows)} dispatches, span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms ({100 * busy / span:.1f}%), "
      f"in-graph gaps {sum(small) / 1e6:.3f} ms (avg {sum(small) / max(1, len(small)) / 1e3:.2f} us over {len(small)}), "
      f"long gaps {sum(g for g in gaps if g >= 30000) / 1e6:.3f} ms")
agg = {}
for s, e, n in rows:
    a = agg.setdefault(n, [0, 0])
    a[0] += 1
    a[1] += e - s
for n, (k, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{n[:70]:70s} n={k:6d} avg={t / k / 1e3:7.2f}us tot={t / 1e6:8.3f}ms {100 * t / busy:5.1f}%")
