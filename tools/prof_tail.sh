#!/bin/bash
# rocprofv3 kernel stats of tools/tail_micro.py (GPU box): prof_tail.sh <out.csv> [tail_micro args]
out=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tail
rocprofv3 --kernel-trace --stats -d /tmp/prof_tail -- python3 $GRAFT_REPO_ROOT/tools/tail_micro.py "$@" > /tmp/tm.log 2>&1
grep "^N=\|relmax" /tmp/tm.log | grep -v "relmax [0-9.]*e-0[78]$"
DB=$(find /tmp/prof_tail -name "*.db" | head -1)
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB $GRAFT_REPO_ROOT/gpurun_out/$out
python3 - <<PY
import csv
for r in csv.reader(open("$GRAFT_REPO_ROOT/gpurun_out/$out")):
    if "tail" in r[0] or "loss_" in r[0] or "Name" in r[0]: print(r[0][:58].ljust(58), r[1:4], r[5:7])
PY
