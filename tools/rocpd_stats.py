#!/usr/bin/env python3
"""rocprofv3 (rocpd SQLite output of `--kernel-trace --stats`) -> the per-kernel stats CSV kept under profiles/.
Usage: rocpd_stats.py <results.db> <out.csv>"""
import csv, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for n, c, t, a, mn, mx in rows:
        w.writerow([n, c, t, f"{a:.3f}", f"{100.0 * t / tot:.4f}", mn, mx])
print(f"{len(rows)} kernels, {tot / 1e6:.1f} ms of kernel time")
