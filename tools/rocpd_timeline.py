#!/usr/bin/env python3
"""rocprofv3 rocpd SQLite (`--kernel-trace`) -> the dispatch timeline of ONE replayed training step (between two clamp_adam_k launches):
seq, kernel, grid, start (us from the step's first kernel), duration (us), idle gap before it (us).   Usage: rocpd_timeline.py <results.db> <out.csv>"""
import csv, re, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
cols = [c[1] for c in db.execute("pragma table_info(kernels)").fetchall()]
if not cols:          # a view: take the names from a row
    cur = db.execute("select * from kernels limit 1"); cols = [d[0] for d in cur.description]
pick = lambda *names: next((n for n in names if n in cols), None)
c_start, c_end, c_name = pick("start", "start_timestamp"), pick("end", "end_timestamp"), pick("name", "kernel_name")
c_grid = pick("grid_size_x", "grid_x", "grid_size")
c_wg = pick("workgroup_size_x", "workgroup_x", "workgroup_size")
sel = ", ".join(c for c in (c_name, c_start, c_end, c_grid, c_wg) if c)
rows = db.execute(f"select {sel} from kernels order by {c_start}").fetchall()
marks = [i for i, r in enumerate(rows) if "clamp_adam_k" in r[0]]
if len(marks) < 3:
    sys.exit("need at least three optimizer launches in the trace")
# the fastest complete step of the trace = a hipGraph replay (the trace also holds eager warm-up and instrumented steps)
spans = [(rows[marks[k + 1]][2] - rows[marks[k] + 1][1], marks[k] + 1, marks[k + 1] + 1) for k in range(len(marks) - 1)]
_, a, b = min(spans)
step = rows[a:b]
t0 = step[0][1]
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", n)).split("(")[0][:90]
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["seq", "kernel", "grid", "workgroup", "start_us", "dur_us", "gap_us"])
    prev_end = None
    for i, r in enumerate(step):
        name, s, e = r[0], r[1], r[2]
        grid = r[3] if c_grid else ""
        wg = r[4] if (c_grid and c_wg) else ""
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        w.writerow([i, short(name), grid, wg, f"{(s - t0) / 1e3:.2f}", f"{(e - s) / 1e3:.2f}", f"{gap:.2f}"])
        prev_end = e if prev_end is None else max(prev_end, e)
busy = sum(r[2] - r[1] for r in step) / 1e6
print(f"{len(step)} dispatches, step span {(step[-1][2] - t0) / 1e6:.3f} ms, kernel time {busy:.3f} ms")
