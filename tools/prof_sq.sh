#!/bin/bash
# SQ counters per kernel (GPU box): MFMA busy cycles, wave cycles and their wait buckets, LDS conflicts - one pass (8 SQ slots), eager launches.
# prof_sq.sh <out.csv> [bench args]   -> gpurun_out/<out.csv> (per kernel name: launches, sum of every counter)
out=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_sq
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pmc_sq -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-line --no-extras "$@" > /tmp/psq.log 2>&1
tail -2 /tmp/psq.log | cut -c1-300
F=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1)
python3 - "$F" "$GRAFT_REPO_ROOT/gpurun_out/$out" <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    d = (r["Dispatch_Id"])
    if d not in seen:
        seen.add(d); n[k] += 1
names = sorted({c for v in acc.values() for c in v})
with open(sys.argv[2], "w") as f:
    f.write("kernel,launches," + ",".join(names) + "\n")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0)):
        f.write(f"\"{k}\",{n[k]}," + ",".join(f"{v.get(c, 0):.0f}" for c in names) + "\n")
PY
head -25 $GRAFT_REPO_ROOT/gpurun_out/$out
