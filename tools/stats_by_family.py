#!/usr/bin/env python3
"""Per-step kernel time by kernel family from a rocpd_stats CSV (tools/prof_bench.sh): stats_by_family.py <csv> [steps]
steps defaults to the call count of clamp_adam_k (one launch per step)."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else next(int(r["Calls"]) for r in rows if "clamp_adam_k" in r["Name"])
agg, calls = collections.Counter(), collections.Counter()
for r in rows:
    n = r["Name"]
    if "spin_kernel" in n:
        continue
    m = re.search(r"::(\w+)", n)
    k = m.group(1) if m else n[:40]
    if "f32f_t" in n:
        k += "<f32f>"
    elif "<float" in n:
        k += "<float>"
    agg[k] += int(r["TotalDurationNs"]); calls[k] += int(r["Calls"])
tot = sum(agg.values())
for k, v in agg.most_common(30):
    print(f"{k:40s} {v / steps / 1e6:8.3f} ms/step  {calls[k] / steps:7.1f} launches/step")
print(f"{'total':40s} {tot / steps / 1e6:8.3f} ms/step  {sum(calls.values()) / steps:7.1f} launches/step   ({steps} steps)")
