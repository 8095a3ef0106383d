#!/bin/bash
# Instruction / cycle counters of the fused-tail kernels (GPU box): prof_tail_pmc.sh <out.csv> [PN2_LIB path]  -> gpurun_out/<out.csv>
# Two PMC passes of tools/tail_micro.py (no trace domains together with --pmc).
out=$1
[ -n "$2" ] && export PN2_LIB=$2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt_a /tmp/pt_b
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES -d /tmp/pt_a -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/tail_micro.py 32 352 0 > /tmp/pta.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d /tmp/pt_b -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/tail_micro.py 32 352 0 > /tmp/ptb.log 2>&1
A=$(find /tmp/pt_a -name "*counter_collection.csv" | head -1); B=$(find /tmp/pt_b -name "*counter_collection.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/pmc_insts.py $GRAFT_REPO_ROOT/gpurun_out/$out $A $B
grep "tail\|loss_fin\|kernel" $GRAFT_REPO_ROOT/gpurun_out/$out
