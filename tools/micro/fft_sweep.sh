#!/bin/bash
# Sweep fft360 build parameters on the GPU box: bash tools/micro/fft_sweep.sh "<defs1>" "<defs2>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/spherical-dyffusion_amd/csrc
for D in "$@"; do
  rm -f fft360.o
  make CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $D" fft360.o > /dev/null 2>&1
  make > /dev/null 2>&1
  echo "== $D"
  (cd $R && python tools/stage_bench.py 25 2>&1 | grep -E "fft")
done
