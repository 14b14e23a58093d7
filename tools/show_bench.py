"""Prints the headline fields of a bench.py JSON line (argument: file)."""
import json
import sys

d = json.load(open(sys.argv[1]))
print({k: v for k, v in d.items() if k != "roofline"})
r = d["roofline"]
print("roofline:", r["frac"], r["ms_per_launch"], r.get("hbm_bound_kernels"))
