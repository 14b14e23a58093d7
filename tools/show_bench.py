"""Prints the headline fields and the per-stage table of a bench.py JSON line (argument: file)."""
import json
import sys

d = json.load(open(sys.argv[1]))
print({k: v for k, v in d.items() if k not in ("roofline", "config", "dtype", "metric")})
r = d.get("roofline")
if r:
    print("roofline:", r["frac"], r["ms_per_launch"], r.get("hbm_bound_kernels"))
    for k in r["kernels"]:
        print(f"  {k['name']:24s} x{k['launches_per_step']:4d} {k['ms']:8.4f} ms  share {k['share']:.4f}  "
              f"hbm {k['frac_hbm']}  mfma {k['frac_mfma']}")
