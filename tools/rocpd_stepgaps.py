#!/usr/bin/env python3
"""rocprofv3 rocpd SQLite (`--kernel-trace`) -> per replayed step: span, summed kernel time, the idle gap in front of the step (host-side launch gaps between
hipGraph replays show up here, not inside a step).   Usage: rocpd_stepgaps.py <results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.execute("select * from kernels limit 1"); cols = [d[0] for d in cur.description]
pick = lambda *names: next((n for n in names if n in cols), None)
c_start, c_end, c_name = pick("start", "start_timestamp"), pick("end", "end_timestamp"), pick("name", "kernel_name")
rows = db.execute(f"select {c_name}, {c_start}, {c_end} from kernels order by {c_start}").fetchall()
marks = [i for i, r in enumerate(rows) if "clamp_adam_k" in r[0]]
out = []
for k in range(len(marks) - 1):
    a, b = marks[k] + 1, marks[k + 1] + 1
    step = rows[a:b]
    span = (step[-1][2] - step[0][1]) / 1e6
    busy = sum(r[2] - r[1] for r in step) / 1e6
    gap = (step[0][1] - rows[marks[k]][2]) / 1e6
    big = sorted(((step[i][1] - max(r[2] for r in step[:i])) / 1e3, i, step[i][0][:50]) for i in range(1, len(step)))[-3:]
    out.append((span, busy, gap, len(step), big))
for span, busy, gap, n, big in out[-12:]:
    print(f"step: {n} dispatches span {span:.3f} ms kernel {busy:.3f} ms gap-before {gap:.3f} ms  largest inner gaps (us): " + ", ".join(f"{g:.1f}@{i}" for g, i, _ in big))
