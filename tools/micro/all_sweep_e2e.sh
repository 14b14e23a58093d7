#!/bin/bash
# End-to-end bench with extra build flags for ALL kernels on the GPU box: bash tools/micro/all_sweep_e2e.sh "<defs1>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/spherical-dyffusion_amd/csrc
for D in "$@"; do
  make clean > /dev/null 2>&1
  make -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $D" > /dev/null 2>&1
  echo "== $D"
  (cd $R && for i in 1 2; do python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"])"; done)
done
