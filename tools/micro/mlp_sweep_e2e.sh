#!/bin/bash
# Build-flag sweep of mlp_h3 inside the network: bash tools/micro/mlp_sweep_e2e.sh "<defs1>" "<defs2>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for D in "$@"; do
  cd $R/spherical-dyffusion_amd/csrc
  rm -f mlp_h3.o
  make CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $D" mlp_h3.o > /dev/null 2>&1
  make > /dev/null 2>&1
  echo "== $D"
  cd $R
  python bench.py --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', d['value'], ' '.join('%s %.4f' % (k['name'], k['ms']) for k in d['roofline']['kernels'] if 'mlp' in k['name'] or 'inner' in k['name']))
"
done
