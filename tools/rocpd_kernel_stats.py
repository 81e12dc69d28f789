#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average / min / max ns, share) from a rocprofv3 rocpd database (rocprofv3 7.2 writes
<name>_results.db for --kernel-trace --stats): usage  rocpd_kernel_stats.py bench_results.db out.csv"""
import csv, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], round(r[3], 3), round(100 * r[2] / tot, 2), r[4], r[5]])
for r in rows[:8]:
    print(r[1], round(r[3] / 1e3, 1), "us", round(100 * r[2] / tot, 1), "%", r[0][:90])
