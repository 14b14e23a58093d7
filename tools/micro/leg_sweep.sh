#!/bin/bash
# Build-flag sweep of leg_par on the GPU box: bash tools/micro/leg_sweep.sh "<defs1>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/spherical-dyffusion_amd/csrc
for D in "$@"; do
  rm -f leg_par.o
  make CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $D" leg_par.o > /dev/null 2>&1
  make > /dev/null 2>&1
  echo "== $D"
  (cd $R && python tools/stage_bench.py 25 2>&1 | grep -E "legendre"; python tools/stage_bench.py 25 2>&1 | grep -E "legendre")
done
