#!/bin/bash
# Re-tune ONLY the weight-gradient entries ('w' keys) of the shipped table pn2/tuned_gfx950.json (GPU box): the conv tile entries are kept, the wgrad
# entries are dropped and tuned again (more candidates than when the table was made), for the three benchmark configurations.  Result: gpurun_out/tuned_new.json
set -e
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import json, ast
t = json.load(open("pranet-v2_amd/pn2/tuned_gfx950.json"))
keep = {k: v for k, v in t.items() if ast.literal_eval(k)[0] != "w"}
json.dump(keep, open("/tmp/tune_cache.json", "w"))
print(len(t), "->", len(keep), "entries kept")
PY
export PN2_TUNE_TABLE=0 PN2_TUNE_CACHE=/tmp/tune_cache.json PN2_TUNE_REPS=7
python3 bench.py --no-cpu-baseline --no-fp32-line --no-extras --steps 5 | tail -1 | cut -c1-200
python3 bench.py --no-cpu-baseline --no-fp32-line --no-extras --steps 5 --model pvt --batch 16 | tail -1 | cut -c1-200
python3 bench.py --no-cpu-baseline --no-fp32-line --no-extras --steps 5 --model emcad --batch 16 --size 512 | tail -1 | cut -c1-200
mkdir -p gpurun_out
cp /tmp/tune_cache.json gpurun_out/tuned_new.json
python3 - <<'PY'
import json, ast, collections
t = json.load(open("gpurun_out/tuned_new.json"))
c = collections.Counter(v[0] for k, v in t.items() if ast.literal_eval(k)[0] == "w")
print(len(t), "entries; wgrad kernel codes:", dict(c))
PY
