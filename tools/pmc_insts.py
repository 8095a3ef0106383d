#!/usr/bin/env python3
"""rocprofv3 --pmc counter_collection.csv files -> one row per (kernel, grid): launches and the mean of every counter per launch.
Usage: pmc_insts.py <out.csv> <counter_collection.csv> [...]"""
import csv, collections, re, sys

def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", n))
    return n.split("(")[0][:100]

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
names = []
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        key = (short(r["Kernel_Name"]), r.get("Grid_Size", ""))
        c = r["Counter_Name"]
        if c not in names:
            names.append(c)
        a = acc[key][c]
        a[0] += float(r["Counter_Value"]); a[1] += 1
with open(sys.argv[1], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid", "launches"] + names)
    for key, cs in sorted(acc.items(), key=lambda kv: -sum(v[0] for v in kv[1].values())):
        n = max(v[1] for v in cs.values())
        w.writerow([key[0], key[1], n] + [f"{cs[c][0] / max(cs[c][1], 1):.0f}" if c in cs else "" for c in names])
print(f"{len(acc)} (kernel, grid) rows, counters: {names}")
