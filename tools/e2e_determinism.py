#!/usr/bin/env python3
"""End-to-end run-to-run determinism of one configuration, many launches, with the components isolated."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sageattention_amd as sa
from sageattention_amd import core
B, H, N, D, causal = 4, 32, 2048, 64, True
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
torch.manual_seed(23)
q = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
k = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
v = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
for fused in (True, False):
    core.FUSE_Q_QUANT = fused
    o0, l0 = sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v, is_causal=causal, return_lse=True)
    km0 = sa.quant.k_mean(k)
    k80, ks0 = core._quant_k(k, km0, "HND", "per_thread")
    nd_o = nd_l = nd_km = nd_k8 = 0
    first = None
    for it in range(runs):
        o, l = sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v, is_causal=causal, return_lse=True)
        km = sa.quant.k_mean(k)
        k8, ks = core._quant_k(k, km, "HND", "per_thread")
        eo, el = torch.equal(o, o0), torch.equal(l, l0)
        nd_o += int(not eo); nd_l += int(not el)
        nd_km += int(not torch.equal(km, km0)); nd_k8 += int(not (torch.equal(k8, k80) and torch.equal(ks, ks0)))
        if (not eo or not el) and first is None:
            d = (o.float() - o0.float()).abs()
            bad = (d > 0).nonzero()
            dl = (l - l0).abs(); badl = (dl > 0).nonzero()
            first = (it, d.max().item(), bad.shape[0], bad[:3].tolist(), sorted(set(bad[:, 2].tolist()))[:8], dl.max().item(), badl.shape[0], badl[:3].tolist())
    print(f"fused_q={fused}: runs {runs}  o differs {nd_o}  lse differs {nd_l}  km differs {nd_km}  k8/ks differs {nd_k8}  first: {first}", flush=True)
