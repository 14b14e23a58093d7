#!/bin/bash
# Ablation sweep of dh_h3 build flags on the GPU box: bash tools/micro/dh_sweep.sh "<defs1>" "<defs2>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/spherical-dyffusion_amd/csrc
for D in "$@"; do
  rm -f dh_h3.o
  make CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $D" dh_h3.o > /dev/null 2>&1
  make > /dev/null 2>&1
  echo "== $D"
  (cd $R && python tools/dh_stamps.py 2>&1 | grep -E "^ms|stamp 2|stamp 3|stamp 5")
done
