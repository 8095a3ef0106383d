#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) into per-kernel HBM traffic per launch.
Units/corrections as MI355X_MICROARCH.md §HBM prescribes: counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of
wide (16 B/lane) coalesced reads, so the read side is doubled; WRITE_SIZE is taken as is (uncalibrated)."""
import csv, collections, json, re, sys

def family(name):
    m = re.search(r"\(anonymous namespace\)::([A-Za-z_]\w*)", name) or re.search(r"([A-Za-z_][\w:]*)\s*[<(]", name.replace("void ", ""))
    return m.group(1) if m else name[:40]

def load(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = agg[family(r["Kernel_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    return agg

if __name__ == "__main__":
    fcsv, wcsv, out = sys.argv[1], sys.argv[2], sys.argv[3]
    commit = sys.argv[4] if len(sys.argv) > 4 else None
    F = load(fcsv, "FETCH_SIZE")
    Wr = load(wcsv, "WRITE_SIZE")
    res = {}
    for k, (f, n) in sorted(F.items(), key=lambda kv: -kv[1][0]):
        w, wn = Wr.get(k, [0.0, 0])
        res[k] = {"launches": n, "fetch_KiB_raw_per_launch": round(f / n, 1), "write_KiB_per_launch": round(w / max(wn, 1), 1),
                  "hbm_bytes_per_launch": int((2.0 * f / n + w / max(wn, 1)) * 1024)}
    # the traced run is bench.py --no-graph --steps 2 --warmup 1 plus its instrumented / eager warm-up steps: STEPS eager training steps in all
    skip = ("FillFunctor", "elementwise", "spin_kernel", "pack_weight", "pack_patch")
    steps = max(1, res.get("clamp_adam_k", {}).get("launches", 1))
    total = sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in res.items() if not any(s_ in k for s_ in skip))
    json.dump({"note": "FETCH_SIZE doubled (gfx950 half-count of 128-B requests), WRITE_SIZE as reported; bench.py --no-graph --steps 2 --warmup 1, shipped tile table "
                       "(pn2/tuned_gfx950.json); per-launch figures are averages over all launches of a kernel family in the traced run",
               "config": {"model": "res2net", "batch": 32, "size": 352, "dtype": "bf16"}, "commit": commit, "traced_steps": steps,
               "step_total_bytes": int(total / steps), "kernels": res}, open(out, "w"), indent=1)
    print(f"steps {steps}  HBM bytes / step {total / steps / 1e9:.2f} GB")
    for k, v in list(res.items())[:16]:
        print(f"{k:28s} n={v['launches']:5d} hbm/launch={v['hbm_bytes_per_launch'] / 1e6:9.2f} MB (fetch raw {v['fetch_KiB_raw_per_launch'] / 1024:8.2f} MiB, write {v['write_KiB_per_launch'] / 1024:8.2f} MiB)")
