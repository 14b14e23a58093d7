#!/bin/bash
# A/B of mlp_h3 builds on one device, interleaved rounds (cdna_hip_programming.md rule 24): bash tools/ab_mlp.sh <out> name=lib ...
OUT=$1; shift
mkdir -p $(dirname $OUT)
: > $OUT
for round in 1 2 3; do
  for kv in "$@"; do
    n=${kv%%=*}; l=${kv#*=}
    echo "== round $round $n" >> $OUT
    SDY_AMD_LIB=$l timeout 300 python tools/mlp_bench.py 25 >> $OUT 2>&1
  done
done
grep -E "==|fused" $OUT | sed 's/| two.*//'
