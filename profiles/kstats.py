"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel table: python tools_kstats.py <db> [out.csv]"""
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
rows = list(c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
lines = ["name,calls,total_ns,avg_ns,min_ns,max_ns,percent"]
for r in rows:
    lines.append(f'"{r[0]}",{r[1]},{r[2]},{r[3]:.1f},{r[4]},{r[5]},{100 * r[2] / tot:.2f}')
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("\n".join(lines) + "\n")
print(f"total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
for r in rows[:28]:
    print(f"{r[0][:84]:84s} n={r[1]:6d} tot={r[2] / 1e6:9.3f}ms avg={r[3] / 1e3:8.2f}us min={r[4] / 1e3:7.2f} max={r[5] / 1e3:8.2f} {100 * r[2] / tot:5.1f}%")
