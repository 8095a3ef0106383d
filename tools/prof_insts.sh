#!/bin/bash
# Instruction mix / stall counters per kernel and grid (GPU box): prof_insts.sh <out.csv>  -> gpurun_out/<out.csv>
# Two PMC passes of an eager step (every kernel its own dispatch), tiles from the shipped table.  No trace domains together with --pmc.
out=$1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pi_a /tmp/pi_b
ARGS="--no-graph --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-line"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM -d /tmp/pi_a -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /tmp/pia.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d /tmp/pi_b -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /tmp/pib.log 2>&1
A=$(find /tmp/pi_a -name "*counter_collection.csv" | head -1); B=$(find /tmp/pi_b -name "*counter_collection.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/pmc_insts.py $GRAFT_REPO_ROOT/gpurun_out/$out $A $B
