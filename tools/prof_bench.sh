#!/bin/bash
# rocprofv3 kernel stats of a bench.py command (GPU box): prof_bench.sh <out.csv> [bench.py args]; the CSV lands in gpurun_out/.
# The shipped tile table (pn2/tuned_gfx950.json) covers every conv shape of the benchmark configurations, so the trace holds no tuner launches and no
# cache-evicting fills: the per-kernel averages are those of the steady-state step.
out=$1; shift
# one rank only: `bench.py --gpus N` would start torchrun from a process the profiler has already initialised the GPU in (bench.py refuses that too)
prev=""; for a in "$@"; do if [ "$prev" = "--gpus" ] && [ "$a" != "1" ]; then echo "prof_bench.sh: profile ONE rank (--gpus $a refused); run multi-rank jobs with the launcher outside the profiler" >&2; exit 2; fi; case "$a" in --gpus=1) ;; --gpus=*) echo "prof_bench.sh: profile ONE rank ($a refused)" >&2; exit 2;; esac; prev="$a"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > /tmp/pb.log 2>&1
tail -1 /tmp/pb.log | cut -c1-260
DB=$(find /tmp/prof_bench -name "*.db" | head -1)
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB $GRAFT_REPO_ROOT/gpurun_out/$out
python3 $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB $GRAFT_REPO_ROOT/gpurun_out/${out%.csv}_timeline.csv
python3 $GRAFT_REPO_ROOT/tools/rocpd_stepgaps.py $DB
