#!/usr/bin/env python3
"""Per-wave instruction counts of the attention kernel from tools/attn_fixed_cost.py's PMC run (pmc_dispatches.py output
filtered to attn_i8): per-tile cost = (count(2N) - count(N)) / extra tiles, fixed cost = count(N) - tiles(N) * per-tile.
usage: fixed_cost_table.py <attn.txt>"""
import ast, re, sys
rows = []
for l in open(sys.argv[1]):
    m = re.match(r"\((\d+), '([^']*)', '(\d+)'\) (\{.*\})", l.strip())
    d = ast.literal_eval(m.group(4))
    k = re.search(r"attn_i8_kernel<(\d+), (\d+)", m.group(2))
    rows.append((int(m.group(1)), int(k.group(1)), int(k.group(2)), d))
rows.sort()
# order of attn_fixed_cost.py: per D: N = 512, 1024, 2048, 4096; per N: fp16 then fp8
print("| head_dim | PV | N | waves/wg | VALU/wave | SALU/wave | per tile VALU | SALU | fixed VALU | fixed SALU |")
print("|---|---|---|---|---|---|---|---|---|---|")
i = 0
for D in (64, 128):
    per = {}
    for N in (512, 1024, 2048, 4096):
        for pv in ("fp16", "fp8"):
            _, d_, nw, c = rows[i]; i += 1
            assert d_ == D
            per[(pv, N)] = (nw, c["SQ_INSTS_VALU"] / c["SQ_WAVES"], c["SQ_INSTS_SALU"] / c["SQ_WAVES"])
    for pv in ("fp16", "fp8"):
        for N in (512, 1024, 2048):
            nw, v, s = per[(pv, N)]
            nw2, v2, s2 = per[(pv, 2 * N)]
            if nw != nw2: continue
            t = N // 64
            pv_t, ps_t = (v2 - v) / t, (s2 - s) / t
            print(f"| {D} | {pv} | {N} | {nw} | {v:.0f} | {s:.0f} | {pv_t:.1f} | {ps_t:.1f} | {v - t * pv_t:.0f} | {s - t * ps_t:.0f} |")
