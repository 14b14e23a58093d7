#!/bin/bash
# Build-flag sweep of mlp_h3 on the GPU box: bash tools/micro/mlp_sweep.sh "<defs1>" "<defs2>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/spherical-dyffusion_amd/csrc
for D in "$@"; do
  rm -f mlp_h3.o
  make CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $D" mlp_h3.o > /dev/null 2>&1
  make > /dev/null 2>&1
  echo "== $D"
  (cd $R && python tools/mlp_stamps.py 2>&1 | grep -E "drop=|fc1|chain|fc2|epilogue|store|LDS|total" | awk '{printf "%s | ", $0} /total/{print ""}'; python tools/mlp_bench.py 25 2>&1 | tail -2 | cut -c1-60)
done
