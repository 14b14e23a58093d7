#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes as MI355X_MICROARCH.md prescribes) of the conv GEMMs at B=25.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for CTR in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d $R/gpurun_out/pmc_$CTR -- python3 $R/tools/${PMC_CASE:-pmc_case25.py} > $R/gpurun_out/pmc_$CTR.log 2>&1
  echo "rc=$? $CTR"
  F=$(find $R/gpurun_out/pmc_$CTR -name "*counter_collection.csv" | head -1)
  python3 - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if any(t in r["Kernel_Name"] for t in ("mlp", "gemm", "fft", "instnorm", "leg_", "dh_h3", "conv_h3")):
        agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in agg.items():
    print(f"{k:62s} {c:12s} avg {sum(v)/len(v):14.1f} KB over {len(v)} dispatches")
PY
done
