#!/usr/bin/env python3
"""Run-to-run determinism sweep over the public entry points (many kernel variants x shapes), N launches each."""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sageattention_amd as sa
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 25
torch.manual_seed(7)
bad = 0
cases = []
for D, causal, pv, gran, dt, layout in itertools.product((64, 128), (False, True), ("fp16", "fp8"), ("per_thread", "per_warp"),
                                                          (torch.float16, torch.bfloat16), ("HND", "NHD")):
    cases.append((4, 16, 8, 1536 + 64 * (D == 64), 1536 + 37, D, causal, pv, gran, dt, layout))
for (B, Hq, Hk, M, N, D, causal, pv, gran, dt, layout) in cases:
    if causal:
        N = M
    shp = (lambda h, n: (B, h, n, D)) if layout == "HND" else (lambda h, n: (B, n, h, D))
    q = torch.randn(shp(Hq, M), dtype=dt, device="cuda")
    k = torch.randn(shp(Hk, N), dtype=dt, device="cuda") + 1
    v = torch.randn(shp(Hk, N), dtype=dt, device="cuda")
    fn = sa.sageattn_qk_int8_pv_fp16_cuda if pv == "fp16" else sa.sageattn_qk_int8_pv_fp8_cuda
    o0, l0 = fn(q, k, v, tensor_layout=layout, is_causal=causal, qk_quant_gran=gran, return_lse=True)
    nd = 0
    for _ in range(runs):
        o, l = fn(q, k, v, tensor_layout=layout, is_causal=causal, qk_quant_gran=gran, return_lse=True)
        nd += int(not (torch.equal(o, o0) and torch.equal(l, l0)))
    if nd:
        bad += 1
        print("NONDETERMINISTIC", (B, Hq, Hk, M, N, D, causal, pv, gran, str(dt), layout), nd, "/", runs)
# varlen and attn_mask entry points
lens = [700, 37, 1200, 64, 999]
cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
T = sum(lens)
for D, causal in ((64, False), (64, True), (128, True)):
    q = torch.randn(T, 8, D, dtype=torch.float16, device="cuda"); k = torch.randn(T, 4, D, dtype=torch.float16, device="cuda"); v = torch.randn(T, 4, D, dtype=torch.float16, device="cuda")
    o0 = sa.sageattn_varlen(q, k, v, cu, cu, max(lens), max(lens), is_causal=causal)
    nd = sum(int(not torch.equal(sa.sageattn_varlen(q, k, v, cu, cu, max(lens), max(lens), is_causal=causal), o0)) for _ in range(runs))
    if nd: bad += 1; print("NONDETERMINISTIC varlen", D, causal, nd)
for D in (64, 128):
    q = torch.randn(2, 4, 500, D, dtype=torch.float16, device="cuda"); k = torch.randn(2, 4, 777, D, dtype=torch.float16, device="cuda"); v = torch.randn(2, 4, 777, D, dtype=torch.float16, device="cuda")
    m = torch.rand(1, 1, 500, 777, device="cuda") > 0.3
    o0 = sa.sageattn_qk_int8_pv_fp16_triton(q, k, v, attn_mask=m)
    nd = sum(int(not torch.equal(sa.sageattn_qk_int8_pv_fp16_triton(q, k, v, attn_mask=m), o0)) for _ in range(runs))
    if nd: bad += 1; print("NONDETERMINISTIC mask", D, nd)
print("cases", len(cases) + 5, "nondeterministic", bad)
