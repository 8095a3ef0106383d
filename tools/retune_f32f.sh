#!/bin/bash
# Re-tune the fp32fast entries (conv GEMM 'g' and weight-gradient 'w' keys ending in 'f32f') of the shipped table pn2/tuned_gfx950.json (GPU box) after a change to the
# register-staged kernels; everything else is kept.  Result: gpurun_out/tuned_f32f.json, and an A/B of the step with the old and the new table.
set -e
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import json
t = json.load(open("pranet-v2_amd/pn2/tuned_gfx950.json"))
keep = {k: v for k, v in t.items() if "'f32f'" not in k}
json.dump(keep, open("/tmp/tune_cache.json", "w"))
print(len(t), "->", len(keep), "entries kept")
PY
mkdir -p gpurun_out
echo "old table: $(python3 bench.py --dtype fp32fast --no-cpu-baseline --no-fp32-line --no-extras --steps 12 --warmup 4 | tail -1 | cut -c90-180)"
PN2_TUNE_TABLE=0 PN2_TUNE_CACHE=/tmp/tune_cache.json PN2_TUNE_REPS=7 python3 bench.py --dtype fp32fast --no-cpu-baseline --no-fp32-line --no-extras --steps 5 | tail -1 | cut -c90-180
cp /tmp/tune_cache.json gpurun_out/tuned_f32f.json
python3 - <<'PY'
import json
old = json.load(open("pranet-v2_amd/pn2/tuned_gfx950.json")); new = json.load(open("gpurun_out/tuned_f32f.json"))
ch = [k for k in new if "'f32f'" in k and old.get(k) != new[k]]
print(len(new), "entries;", len(ch), "fp32fast entries changed of", sum(1 for k in new if "'f32f'" in k))
PY
cp gpurun_out/tuned_f32f.json /tmp/new_table.json
echo "new table: $(PN2_TUNE_TABLE=0 PN2_TUNE_CACHE=/tmp/new_table.json python3 bench.py --dtype fp32fast --no-cpu-baseline --no-fp32-line --no-extras --steps 12 --warmup 4 | tail -1 | cut -c90-180)"
echo "old table: $(python3 bench.py --dtype fp32fast --no-cpu-baseline --no-fp32-line --no-extras --steps 12 --warmup 4 | tail -1 | cut -c90-180)"
