#!/bin/bash
# PMC pass over a small conv-GEMM run (GPU box).  usage: bash tools/pmc_probe.sh <script.py> [args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for CTRS in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_MFMA"; do
  TAG=$(echo $CTRS | md5sum | cut -c1-6)
  rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $R/gpurun_out/pmc_$TAG -- python3 $R/$1 ${@:2} > $R/gpurun_out/pmc_$TAG.log 2>&1
  echo "rc=$? $CTRS"
  F=$(find $R/gpurun_out/pmc_$TAG -name "*counter_collection.csv" | head -1)
  python3 - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if not any(t in k for t in ("gemm", "fft", "mlp", "dh_h3", "leg_h3", "conv_h3")): continue
    print(k)
    for c, v in d.items(): print(f"    {c:28s} {v / cnt[(k, c)]:16.1f} (avg per dispatch, {cnt[(k,c)]} dispatches)")
PY
done
