#!/bin/bash
# Build-flag sweep of conv_h3 inside the network: bash tools/micro/conv_sweep_e2e.sh "<defs1>" "<defs2>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for D in "$@"; do
  cd $R/spherical-dyffusion_amd/csrc
  rm -f conv_h3.o
  make CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $D" conv_h3.o > /dev/null 2>&1
  make > /dev/null 2>&1
  echo "== $D"
  cd $R
  python bench.py --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', d['value'])
for k in d['roofline']['kernels']:
    if 'conv' in k['name']: print('  ', k['name'], k['ms'], k.get('frac_hbm'))
"
done
