"""Per-kernel PMC summary of a rocprofv3 --pmc run (rocpd database): python profiles/pmcstats.py <db>
Groups by (kernel name, grid size) so that launches of one kernel on different matrix shapes stay apart; sorted by the
counter's total over all launches."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select kernel_name, grid_size, workgroup_size, counter_name, count(*), avg(value), avg(duration) "
                      "from counters_collection group by kernel_name, grid_size, workgroup_size, counter_name"))
rows.sort(key=lambda r: -r[4] * r[5])
print("kernel,grid,wg,counter,launches,avg_value,avg_duration_ns")
for r in rows:
    print(f'"{r[0][:60]}",{r[1]},{r[2]},{r[3]},{r[4]},{r[5]:.1f},{r[6]:.0f}')
