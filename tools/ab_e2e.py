#!/usr/bin/env python3
"""In-process A/B of the end-to-end operator with and without the fused Q quantizer."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sageattention_amd as sa
from sageattention_amd import core
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
B, H, N, D, causal, fn = {"c3": (4, 32, 8192, 128, False, sa.sageattn_qk_int8_pv_fp16_cuda),
                          "c3c": (4, 32, 8192, 128, True, sa.sageattn_qk_int8_pv_fp16_cuda),
                          "c2": (4, 32, 2048, 64, False, sa.sageattn_qk_int8_pv_fp16_cuda),
                          "c4": (4, 32, 16384, 128, True, sa.sageattn_qk_int8_pv_fp8_cuda)}[wl]
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, N, D, dtype=torch.float16, device="cuda") for _ in range(3))
def run(fused, iters):
    core.FUSE_Q_QUANT = fused
    core.FUSE_Q_MAX_SEQ = 1 << 30  # compare the two paths at every length
    for _ in range(2): fn(q, k, v, is_causal=causal)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn(q, k, v, is_causal=causal)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
res = {True: [], False: []}
for r in range(7):
    for f in (True, False):
        res[f].append(run(f, 10))
fl = 4.0 * B * H * N * N * D / (2 if causal else 1)
for f in (True, False):
    m = statistics.median(res[f])
    print(f"{wl} fused={f}: median {m:.4f} ms -> {fl/m/1e9:.1f} TFLOPS")
