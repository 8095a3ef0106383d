#!/bin/bash
# Re-tune the conv GEMM entries ('g' keys: forward / dgrad / dgrad + BatchNorm-backward epilogue) of the shipped table pn2/tuned_gfx950.json (GPU box) after
# a kernel change; the weight-gradient entries ('w') are kept.  Three benchmark configurations.  Result: gpurun_out/tuned_gemm.json
set -e
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import json, ast
t = json.load(open("pranet-v2_amd/pn2/tuned_gfx950.json"))
keep = {k: v for k, v in t.items() if ast.literal_eval(k)[0] != "g"}
json.dump(keep, open("/tmp/tune_cache.json", "w"))
print(len(t), "->", len(keep), "entries kept")
PY
export PN2_TUNE_TABLE=0 PN2_TUNE_CACHE=/tmp/tune_cache.json PN2_TUNE_REPS=7
python3 bench.py --no-cpu-baseline --no-fp32-line --no-extras --steps 5 | tail -1 | cut -c1-200
python3 bench.py --no-cpu-baseline --no-fp32-line --no-extras --steps 5 --model pvt --batch 16 | tail -1 | cut -c1-200
python3 bench.py --no-cpu-baseline --no-fp32-line --no-extras --steps 5 --model emcad --batch 16 --size 512 | tail -1 | cut -c1-200
mkdir -p gpurun_out
cp /tmp/tune_cache.json gpurun_out/tuned_gemm.json
python3 - <<'PY'
import json, ast
old = json.load(open("pranet-v2_amd/pn2/tuned_gfx950.json")); new = json.load(open("gpurun_out/tuned_gemm.json"))
ch = [k for k in new if ast.literal_eval(k)[0] == "g" and old.get(k) != new[k]]
print(len(new), "entries;", len(ch), "conv entries changed of", sum(1 for k in new if ast.literal_eval(k)[0] == "g"))
PY
