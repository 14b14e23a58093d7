#!/usr/bin/env python3
"""Instruction census of the largest loop of every kernel in a hipcc --save-temps .s file.

usage: isa_census.py file.s [kernel-substring]
Prints, per kernel, the instruction classes inside its largest backward-branch loop (the persistent tile loop) with an
issue-cycle estimate (VALU 4, transcendental / packed / 64-bit 8, quarter-rate integer multiplies 16, MFMA 8 issue).
"""
import collections
import re
import sys

TRANS = ("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")
QUARTER = ("v_mul_hi_u32", "v_mul_lo_u32", "v_mad_u64_u32", "v_mad_i64_i32", "v_mul_hi_i32")
F64 = ("_f64",)


def classify(m):
    if m.startswith("v_mfma"):
        return "mfma", 8
    if m.startswith(TRANS):
        return "valu.trans", 8
    if m.startswith(QUARTER):
        return "valu.quarter", 16
    if m.startswith("v_pk_"):
        return "valu.packed", 8
    if m.startswith("v_") and m.endswith(("_f64", "_b64", "_u64", "_i64")) or "_f64" in m:
        return "valu.64", 8
    if m.startswith("v_accvgpr") or m.startswith("v_mov"):
        return "valu.mov", 4
    if m.startswith("v_cvt"):
        return "valu.cvt", 4
    if m.startswith("v_"):
        return "valu.other", 4
    if m.startswith("ds_"):
        return "lds", 4
    if m.startswith(("global_load", "buffer_load", "flat_load")):
        return "vmem.load", 4
    if m.startswith(("global_store", "buffer_store", "flat_store", "global_atomic")):
        return "vmem.store", 4
    if m.startswith("scratch_"):
        return "scratch", 4
    if m.startswith("s_waitcnt"):
        return "s_waitcnt", 1
    if m.startswith("s_barrier"):
        return "s_barrier", 1
    if m.startswith("s_"):
        return "salu", 1
    return "other", 1


def kernels(lines):
    start, name = None, None
    for i, ln in enumerate(lines):
        mm = re.match(r"^(_Z\w+):", ln)
        if mm and start is None:
            start, name = i, mm.group(1)
        if ln.strip() == "s_endpgm" and start is not None:
            yield name, start, i
            start = None


def main():
    lines = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, a, b in kernels(lines):
        if want not in name:
            continue
        labels = {}
        for i in range(a, b):
            mm = re.match(r"^(\.LBB\d+_\d+):", lines[i])
            if mm:
                labels[mm.group(1)] = i
        best = (0, None, None)
        for i in range(a, b):
            mm = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", lines[i])
            if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                if i - labels[mm.group(1)] > best[0]:
                    best = (i - labels[mm.group(1)], labels[mm.group(1)], i)
        _, la, lb = best
        if la is None:
            continue
        cnt, cyc = collections.Counter(), collections.Counter()
        mn = collections.Counter()
        for i in range(la, lb + 1):
            mm = re.match(r"^\s+([a-z_0-9]+)\b", lines[i])
            if not mm or lines[i].lstrip().startswith((".", ";")):
                continue
            c, w = classify(mm.group(1))
            cnt[c] += 1
            cyc[c] += w
            mn[mm.group(1)] += 1
        print(f"== {name}  loop lines {la - a}..{lb - a} of {b - a}")
        tot = sum(cyc.values())
        for c, n in sorted(cnt.items(), key=lambda kv: -cyc[kv[0]]):
            print(f"  {c:14s} {n:6d} instr  ~{cyc[c]:6d} issue cycles ({100 * cyc[c] / tot:4.1f} %)")
        print(f"  total ~{tot} issue cycles per wave per loop trip; matrix pipe busy {cnt['mfma'] * 32}")
        if "-v" in sys.argv:
            for m, n in mn.most_common(40):
                print(f"     {m:28s} {n}")


main()
