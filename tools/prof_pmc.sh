#!/bin/bash
# HBM traffic per kernel family from rocprofv3 PMC counters (GPU box): two separate passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass, and
# gpurun refuses --pmc combined with trace domains), eager launches (--no-graph) so that every kernel is its own dispatch, tiles from the shipped table.
# prof_pmc.sh <out.json> <commit>   -> gpurun_out/<out.json>
out=$1; commit=$2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_f -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-line --no-extras > /tmp/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_w -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-line --no-extras > /tmp/pw.log 2>&1
F=$(find /tmp/pmc_f -name "*counter_collection.csv" | head -1); W=$(find /tmp/pmc_w -name "*counter_collection.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $F $W $GRAFT_REPO_ROOT/gpurun_out/$out $commit
python3 $GRAFT_REPO_ROOT/tools/pmc_step_dispatches.py $F $W $GRAFT_REPO_ROOT/gpurun_out/${out%.json}_dispatches.csv
