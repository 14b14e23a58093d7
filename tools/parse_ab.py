import re,collections,sys
rows=collections.defaultdict(lambda: collections.defaultdict(list)); vals=collections.defaultdict(list)
mode=None; modes=[]
for line in open(sys.argv[1]):
    m=re.match(r"== round (\d) (\w+)",line)
    if m:
        mode=m.group(2)
        if mode not in modes: modes.append(mode)
        continue
    m=re.match(r"\{'value': ([0-9.]+)",line)
    if m: vals[mode].append(float(m.group(1))); continue
    m=re.match(r"\s+(.+?)\s+x\s+(\d+)\s+([0-9.]+) ms",line)
    if m: rows[m.group(1)][mode].append((int(m.group(2)),float(m.group(3))))
print({k:round(sum(v)/len(v),2) for k,v in vals.items()})
base=modes[0]
for name,d in rows.items():
    if not d[base] or any(not d[mo] for mo in modes[1:]): continue   # a stage only one side runs (e.g. the drop-path copy)
    c=d[base][0][0]; h=sum(x[1] for x in d[base])/len(d[base])
    out=f"{name:26s} x{c:4d} {base} {h:.4f}"
    for mo in modes[1:]:
        n=sum(x[1] for x in d[mo])/len(d[mo]); out+=f" | {mo} {n:.4f} ({(n-h)*c:+.2f} ms/pass)"
    print(out)
