#!/usr/bin/env python3
"""Per-dispatch HBM bytes of ONE eager training step from the two PMC passes (FETCH_SIZE, WRITE_SIZE counter_collection.csv files of tools/prof_pmc.sh), in launch
order, for joining with the dispatch timeline of a replayed step (tools/rocpd_timeline.py: same launches in the same order).
Usage: pmc_step_dispatches.py <fetch.csv> <write.csv> <out.csv>"""
import csv, re, sys

def short(n):
    return re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", n)).split("(")[0][:90]

def load(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    key = "Dispatch_Id" if "Dispatch_Id" in rows[0] else None
    if key:
        rows.sort(key=lambda r: int(r[key]))
    return rows

F, W = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
def last_step(rows):
    marks = [i for i, r in enumerate(rows) if "clamp_adam_k" in r["Kernel_Name"]]
    return rows[marks[-2] + 1: marks[-1] + 1]
f, w = last_step(F), last_step(W)
assert len(f) == len(w), (len(f), len(w))
with open(sys.argv[3], "w", newline="") as o:
    wr = csv.writer(o)
    wr.writerow(["seq", "kernel", "grid", "hbm_MB"])
    for i, (a, b) in enumerate(zip(f, w)):
        assert short(a["Kernel_Name"]) == short(b["Kernel_Name"])
        wr.writerow([i, short(a["Kernel_Name"]), a.get("Grid_Size", ""), f"{(2.0 * float(a['Counter_Value']) + float(b['Counter_Value'])) * 1024 / 1e6:.3f}"])
print(len(f), "dispatches")
