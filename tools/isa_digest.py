"""Condenses a hipcc -save-temps .s file to the memory / MFMA / wait skeleton of one kernel (dev aid)."""
import re, sys
pat = re.compile(r"^\s+(s_waitcnt|s_barrier|v_mfma\w*|global_load\w*|global_store\w*|scratch_\w+|s_cbranch\w*|ds_read\w*|ds_write\w*|buffer_\w+)\b(.*)")
prev, cnt, out = None, 0, []
for line in open(sys.argv[1]):
    if re.match(r"^\.LBB", line):
        key = line.strip()
    else:
        m = pat.match(line)
        if not m: continue
        key = m.group(1)
        if key == "s_waitcnt": key += " " + m.group(2).strip()
        if key.startswith("v_mfma"): key = "mfma"
    if key == prev: cnt += 1
    else:
        if prev: out.append(f"{cnt}x{prev}" if cnt > 1 else prev)
        prev, cnt = key, 1
out.append(f"{cnt}x{prev}")
w = 0
for o in out:
    if w + len(o) > 150: print(); w = 0
    print(o, end=" | "); w += len(o) + 3
print()
