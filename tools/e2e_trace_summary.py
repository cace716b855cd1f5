#!/usr/bin/env python3
"""Summary of a `rocprofv3 --kernel-trace` of tools/e2e_trace.py: per shape, the mean duration of every kernel of one
operator call, the mean idle gap in front of it (previous kernel's end -> this kernel's start) and the call period.
usage: e2e_trace_summary.py <trace dir> <stdout of e2e_trace.py>"""
import csv, glob, os, re, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
names = [l.split(" ", 1)[1].strip() for l in open(sys.argv[2]) if l.startswith("SHAPE ")]
def short(n):
    m = re.search(r"(\w+)<", n) or re.search(r"(\w+)\(", n) or re.search(r"(\w+)", n)
    return m.group(1)
# cut at the float64 fill markers
segs, cur = [], None
for s, e, n in rows:
    if "FillFunctor<double>" in n:
        cur = []; segs.append(cur); continue
    if cur is not None: cur.append((s, e, short(n)))
segs = segs[1:]  # the first fill is the marker tensor's own initialisation
print("| shape | kernel (in call order) | mean us | gap in front us |")
print("|---|---|---|---|")
for name, seg in zip(names, segs):
    cut = next((i for i, x in enumerate(seg) if not x[2].endswith("_kernel") or "elementwise" in x[2]), len(seg))
    seg = seg[:cut]  # what follows the timed calls: the next shape's random inputs and warm-ups
    # kernels per call = period of the name sequence
    seq = [x[2] for x in seg]
    per = next(p for p in range(1, len(seq)) if seq[:p] == seq[p:2 * p])
    ncall = next((c for c in range(1, len(seq) // per + 1) if seq[c * per:(c + 1) * per] != seq[:per]), len(seq) // per)
    seg = seg[:ncall * per]  # (the warm-up calls of the next operator on the same inputs follow)
    tot = (seg[-1][1] - seg[per - 1][1]) / (ncall - 1) / 1e3  # steady-state period: end of call 0 -> end of the last call
    for i in range(per):
        d = [seg[c * per + i][1] - seg[c * per + i][0] for c in range(1, ncall)]
        g = [seg[c * per + i][0] - seg[c * per + i - 1][1] for c in range(1, ncall)]
        print(f"| {name} | {seq[i]} | {sum(d) / len(d) / 1e3:.1f} | {sum(g) / len(g) / 1e3:.1f} |")
    print(f"| {name} | **call period** | {tot:.1f} | |")
