#!/usr/bin/env python3
"""Assemble profiles/<tag>/pmc_summary.txt from what tools/profile_round.sh leaves in gpurun_out/<tag>/: the per-counter
averages (pmc_*.txt: kernel prefix (64 chars), counter, average per dispatch) and the rocprofv3 --stats kernel_stats.csv.

    python tools/pmc_summary.py gpurun_out/r3a > profiles/r3a/pmc_summary.txt

HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KB; gfx950 reports half the bytes of wide coalesced
streaming reads: MI355X_MICROARCH.md 'HBM'); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs);
clock = GRBM_GUI_ACTIVE / 8 / the kernel's average duration in kernel_stats.csv."""
import csv
import os
import re
import sys

d = sys.argv[1]
tag = os.path.basename(os.path.normpath(d))
cnt = {}
for fn in os.listdir(d):
    if not (fn.startswith("pmc_") and fn.endswith(".txt")) or fn == "pmc_summary.txt":
        continue
    for line in open(os.path.join(d, fn)):
        m = re.match(r"(.{66})\s+(\S+)\s+avg\s+([0-9.]+) over (\d+) dispatches", line)
        if m:
            cnt.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(3))
dur = {}
for r in csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))):
    dur[r["Name"][:64]] = float(r["AverageNs"]) * 1e-6
print(f"Round {tag}: one interpolator (dropout on) + one forecaster forward at B = 25 (tools/pmc_forward25.py), rocprofv3 --kernel-trace --pmc,")
print("one counter group per pass (tools/profile_round.sh; assembled by tools/pmc_summary.py).  FETCH_SIZE / WRITE_SIZE are reported in KB;")
print("FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced streaming reads, MI355X_MICROARCH.md 'HBM').  MFMA busy =")
print("SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); clock = GRBM_GUI_ACTIVE / 8 / (average duration of the kernel")
print(f"in profiles/{tag}/kernel_stats.csv, the bench command under rocprofv3 --kernel-trace --stats).\n")
print("%-56s %9s %9s %9s %9s %8s %7s" % ("kernel", "fetch GB", "write GB", "total GB", "MFMA busy", "clk GHz", "avg ms"))
for k in sorted(cnt):
    c = cnt[k]
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    ms = next((v for n, v in dur.items() if n.startswith(k[:60])), None)
    if ms is None or ms < 0.2:
        continue
    f, w = 2.0 * c["FETCH_SIZE"] * 1024 / 1e9, c["WRITE_SIZE"] * 1024 / 1e9
    gui = c.get("GRBM_GUI_ACTIVE")
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES")
    mb = "%8.1f%%" % (100.0 * busy / 1024 / (gui / 8)) if (busy is not None and gui) else "       - "
    clk = "%8.2f" % (gui / 8 / (ms * 1e-3) / 1e9) if gui else "      - "
    print("%-56s %9.3f %9.3f %9.3f %9s %8s %7.3f" % (k[:56], f, w, f + w, mb, clk, ms))
