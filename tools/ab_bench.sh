#!/bin/bash
# usage: tools/ab_bench.sh outdir "NAME1:ENV=1 ENV2=2" "NAME2:..."  -> one bench.py run per configuration, prints images/s and ms/step
out=$1; shift
mkdir -p $out
for cfg in "$@"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python bench.py --no-cpu-baseline --no-fp32-line --no-extras --steps 60 > $out/$name.log 2>&1
  python - "$out/$name.log" "$name" "$envs" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith('{"metric"')]
if not line:
    print(sys.argv[2], "FAILED", open(sys.argv[1]).read()[-600:])
else:
    j = json.loads(line[-1]); k = j["kernels"]
    print(f"{sys.argv[2]:24s} {j['value']:8.1f} img/s  {j['ms_per_step']:7.3f} ms   [{sys.argv[3]}]")
PY
done
