#!/bin/bash
# Build-flag sweep of dh_h3 on the GPU box: bash tools/micro/dh_sweep.sh "<defs1>" "<defs2>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for D in "$@"; do
  cd $R/spherical-dyffusion_amd/csrc
  rm -f dh_h3.o
  make CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $D" dh_h3.o > /dev/null 2>&1
  make > /dev/null 2>&1
  echo "== $D"
  cd $R
  python -m pytest tests/test_gpu_ops.py -q -k dhconv 2>&1 | tail -1
  python tools/stage_bench.py 25 2>&1 | grep -E "dhconv \(frag"
  python tools/stage_bench.py 25 2>&1 | grep -E "dhconv \(frag"
done
