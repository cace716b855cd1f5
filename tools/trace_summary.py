#!/usr/bin/env python3
"""rocprofv3 --kernel-trace csv -> per (attention kernel instantiation, grid) the launches, mean / min duration and the
TFLOP/s of the sweep shape that grid belongs to ((4,32,N,D) non-causal: grid = 128 * N / (32 * waves)).
usage: trace_summary.py <dir with *kernel_trace.csv> [skip]   (skip: warm-up launches dropped per group, default 3)"""
import csv, glob, os, re, sys
from collections import defaultdict
d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"attn_i8_kernel<(\d+), (\d+), (\w+), (\w+), (\w+), (\w+), (\w+)>", r["Kernel_Name"])
        if not m:
            continue
        D, nw, causal, kth, vbf, fp8, mask = m.groups()
        wgs = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
        rows[(int(D), int(nw), fp8 == "true", causal == "true", wgs)].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
print("| head_dim | PV | waves | workgroups | keys N | launches | mean us | min us | TFLOP/s (mean) | TFLOP/s (min) |")
print("|---|---|---|---|---|---|---|---|---|---|")
for (D, nw, fp8, causal, wgs), v in sorted(rows.items()):
    v = [dur for _, dur in sorted(v)][skip:]
    if not v:
        continue
    N = wgs * 32 * nw // 128            # B*H = 128 heads
    fl = 4.0 * 128 * N * N * D / (2 if causal else 1)
    mean, mn = sum(v) / len(v), min(v)
    print(f"| {D} | {'fp8' if fp8 else 'fp16'}{' causal' if causal else ''} | {nw} | {wgs} | {N} | {len(v)} | {mean / 1e3:.1f} | {mn / 1e3:.1f} | {fl / mean / 1e3:.0f} | {fl / mn / 1e3:.0f} |")
