#!/bin/bash
# HBM bytes of the depth-wise kernels (tools/dw_one.py) from two separate PMC passes; prints KiB per launch per kernel (FETCH_SIZE raw: x2 for 16 B/lane reads on gfx950)
cd /tmp && export TMPDIR=/tmp
for pipe in 0; do
  export PN2_DW_PIPE=$pipe
  rm -rf /tmp/dwf /tmp/dww
  rocprofv3 --pmc FETCH_SIZE -d /tmp/dwf -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dw_one.py > /tmp/dwf.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d /tmp/dww -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dw_one.py > /tmp/dww.log 2>&1
  echo "PN2_DW_PIPE=$pipe"
  python3 - <<'PY'
import csv, glob, collections
for d, c in (("/tmp/dwf", "FETCH_SIZE"), ("/tmp/dww", "WRITE_SIZE")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and "dwconv" in r["Kernel_Name"]:
            import re
            k = re.search(r"dwconv3x3_\w+<[^>]*>", r["Kernel_Name"]).group(0)
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    for k, (v, n) in agg.items():
        print(f"  {c} {k:60s} n={n} {v / n / 1024:9.1f} MiB per launch (raw)")
PY
done
