#!/usr/bin/env python3
"""Kernel-only launches of the two attention kernels over the sweep's shapes, for `rocprofv3 --kernel-trace`:
(4,32,N,D), D in {64,128}, N in 1K..16K, FP16 PV and FP8 PV, non-causal; `iters` back-to-back launches per shape after
3 warm-up launches (steady state), or with --isolated a device synchronisation and a 3 ms pause before every launch (the
figure a tracer quotes for one launch on an idle, cooled chip).  tools/trace_summary.py turns the trace into a table."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sageattention_amd as sa
from sageattention_amd import _lib as L, _qattn, core
iters = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30
isolated = "--isolated" in sys.argv
only = [a for a in sys.argv[1:] if a.startswith("--only=")]
shapes = [(4, 32, N, D) for D in (64, 128) for N in (1024, 2048, 4096, 8192, 16384)]
if only:
    keep = only[0][7:].split(",")
    shapes = [s for s in shapes if f"{s[2]}x{s[3]}" in keep]
for (B, H, N, D) in shapes:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.float16, device="cuda") for _ in range(3))
    km = sa.quant.k_mean(k)
    q8, qs, k8, ks, _ = core._quant_qk(q, k, km, "HND", "per_thread", D ** -0.5, 32, False, H, H)
    v8, vsc, _ = sa.quant.per_channel_fp8(v, smooth_v=False)
    o = torch.empty_like(q)
    f16 = lambda: _qattn._attn_f16(q8, k8, v, o, qs, ks, None, 1, 0, L.GRAN_PER_THREAD, D ** -0.5, 0)
    f8 = lambda: _qattn._attn_f8(q8, k8, v8, o, qs, ks, vsc, None, 1, 0, L.GRAN_PER_THREAD, D ** -0.5, 0)
    for fn in (f16, f8):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        for _ in range(iters):
            if isolated:
                torch.cuda.synchronize(); time.sleep(0.003)
            fn()
        torch.cuda.synchronize()
    print("done", (B, H, N, D), flush=True)
