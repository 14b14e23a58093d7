#!/bin/bash
# Build-flag sweep of conv_h3 on the GPU box: bash tools/micro/conv_sweep.sh "<defs1>" "<defs2>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for D in "$@"; do
  cd $R/spherical-dyffusion_amd/csrc
  rm -f conv_h3.o
  make CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $D" conv_h3.o > /dev/null 2>&1
  make > /dev/null 2>&1
  echo "== $D"
  cd $R
  python -m pytest tests/test_gpu_ops.py -q -k "conv256 or conv_cin" 2>&1 | tail -2
  SDY_CONV_STAMPS=1 python tools/conv_bench.py 2>&1 | grep -E "conv_h3|tile 1"
  python tools/conv_bench.py 2>&1 | grep -E "conv_h3"
done
