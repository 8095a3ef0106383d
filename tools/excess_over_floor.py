#!/usr/bin/env python3
"""Where a replayed step spends time above its memory floor: joins the per-dispatch HBM bytes of one eager step (tools/pmc_step_dispatches.py, PMC) with
the per-dispatch durations of one replayed step (tools/rocpd_timeline.py) - same launches, same order - and ranks kernels by
    excess = duration - bytes / 5 TB/s - 2 us          (5 TB/s: what the streaming kernels of this repo reach; 2 us: a graph node's floor + ramp)
Usage: excess_over_floor.py <dispatches.csv> <timeline.csv>"""
import collections, csv, difflib, sys

a = list(csv.DictReader(open(sys.argv[1]))); b = list(csv.DictReader(open(sys.argv[2])))
ka = [r["kernel"].split("<")[0] for r in a]; kb = [r["kernel"].split("<")[0] for r in b]
pairs = []
for tag, i1, i2, j1, j2 in difflib.SequenceMatcher(None, ka, kb, autojunk=False).get_opcodes():
    if tag == "equal":
        pairs += [(i1 + k, j1 + k) for k in range(i2 - i1)]
rows = []
for i, j in pairs:
    mb, us = float(a[i]["hbm_MB"]), float(b[j]["dur_us"])
    rows.append((us - mb / 5.0 - 2.0, us, mb, b[j]["kernel"][:64], b[j]["grid"], j))
print(f"{len(pairs)} dispatches matched; step {sum(r[1] for r in rows) / 1e3:.2f} ms, {sum(r[2] for r in rows) / 1e3:.1f} GB; "
      f"time above the floor {sum(max(r[0], 0) for r in rows) / 1e3:.2f} ms")
fam = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for r in rows:
    f = fam[r[3].split("<")[0]]; f[0] += 1; f[1] += r[1]; f[2] += r[2]; f[3] += max(r[0], 0)
print(f"{'kernel':28s} {'n':>4s} {'ms':>7s} {'GB':>7s} {'TB/s':>6s} {'excess ms':>10s}")
for k, f in sorted(fam.items(), key=lambda kv: -kv[1][3])[:20]:
    print(f"{k:28s} {f[0]:4d} {f[1] / 1e3:7.3f} {f[2] / 1e3:7.2f} {f[2] / f[1]:6.2f} {f[3] / 1e3:10.3f}")
print()
for r in sorted(rows, reverse=True)[:30]:
    print(f"excess {r[0]:6.1f} us  dur {r[1]:7.1f} us  {r[2]:8.1f} MB  {r[2] / r[1]:5.2f} TB/s  {r[3]} grid {r[4]} #{r[5]}")
