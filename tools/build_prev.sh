#!/bin/bash
# Build the library of the last COMMIT next to the working-tree one (pranet-v2_amd/csrc/libpn2_prev.so) for an A/B on one box:
#   tools/build_prev.sh && gpurun -- 'tools/ab_bench.sh out "new:PN2_X=0" "prev:PN2_LIB=$PWD/pranet-v2_amd/csrc/libpn2_prev.so"'
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/pn2_prev && mkdir -p /tmp/pn2_prev
(cd "$R" && git archive HEAD pranet-v2_amd/csrc include) | tar -x -C /tmp/pn2_prev
make -C /tmp/pn2_prev/pranet-v2_amd/csrc -j6 > /tmp/pn2_prev/make.log 2>&1
cp /tmp/pn2_prev/pranet-v2_amd/csrc/libpn2_hip.so "$R/pranet-v2_amd/csrc/libpn2_prev.so"
ls -la "$R"/pranet-v2_amd/csrc/*.so
