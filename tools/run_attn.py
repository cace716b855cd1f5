#!/usr/bin/env python3
"""Launch the attention kernel alone a few times (for rocprofv3).  usage: run_attn.py [workload] [iters] [nwaves]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sageattention_amd as sa
from sageattention_amd import _lib as L, _qattn, core

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nw = int(sys.argv[3]) if len(sys.argv) > 3 else 0
B, H, N, D, causal = {"c2": (4, 32, 2048, 64, False), "c3": (4, 32, 8192, 128, False),
                      "c3c": (4, 32, 8192, 128, True), "c2c": (4, 32, 2048, 64, True),
                      "c4": (4, 32, 16384, 128, True),
                      # one rank's launch of the gather schedule at configs[4] (8 ranks): its 8192 query rows x all 65536 keys
                      "c5r": (1, 32, 65536, 128, False)}[wl]
fp8 = wl in ("c4", "c5r")  # configs[3], configs[4]: INT8 QK^T + FP8 PV
M = 8192 if wl == "c5r" else N
L.lib().sage_set_tuning(0, nw)
torch.manual_seed(0)
q = torch.randn(B, H, M, D, dtype=torch.float16, device="cuda")
k = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
v = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
km = sa.quant.k_mean(k)
q8, qs, k8, ks, _ = core._quant_qk(q, k, km, "HND", "per_thread", D ** -0.5, 32, False, H, H)
o = torch.empty_like(q)
if fp8:
    v8, vsc, _ = sa.quant.per_channel_fp8(v, smooth_v=False)
    vpad = int(os.environ.get("VPAD", "0"))  # experiment: V^T rows `vpad` bytes further apart (power-of-two row strides alias in L2)
    if vpad:
        buf = torch.empty(v8.shape[:-1] + (v8.shape[-1] + vpad,), dtype=v8.dtype, device="cuda")
        buf[..., :v8.shape[-1]].copy_(v8)
        v8 = buf[..., :v8.shape[-1]]
for _ in range(iters):
    if fp8:
        _qattn._attn_f8(q8, k8, v8, o, qs, ks, vsc, None, 1, int(causal), 3, D ** -0.5, 0)
    else:
        _qattn._attn_f16(q8, k8, v, o, qs, ks, None, 1, int(causal), 3, D ** -0.5, 0)
torch.cuda.synchronize()
print("done", wl, iters)
