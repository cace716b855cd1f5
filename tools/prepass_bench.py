#!/usr/bin/env python3
"""Pre-pass kernels alone: microseconds and GB/s of algorithmic bytes for K (mean + quantizer: 2 B read twice, 1 B
written -> 5 B/element), Q (3 B/element) and the FP8 V quantizer (5 B/element)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sageattention_amd as sa
from sageattention_amd import _lib as L, quant
from sageattention_amd.quant import _quant, k_mean

def timeit(f, n=30):
    for _ in range(3): f()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n)
    return statistics.median(ts) * 1e3

print("| shape | K mean+quant us | GB/s | Q quant us | GB/s | V fp8 quant us | GB/s |")
print("|---|---|---|---|---|---|---|")
for (B, H, N, D) in [(4, 32, 1024, 64), (4, 32, 2048, 64), (4, 32, 2048, 128), (4, 32, 8192, 128), (4, 32, 16384, 128), (1, 32, 65536, 128)]:
    torch.manual_seed(0)
    q = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
    tk = timeit(lambda: _quant(k, "HND", L.GRAN_PER_THREAD, True, 64, 64, 1.0, L.ROUND_TRITON, mean=k_mean(k)))
    tq = timeit(lambda: _quant(q, "HND", L.GRAN_PER_THREAD, False, 128, 32, 1.0, L.ROUND_TRITON))
    tv = timeit(lambda: quant.per_channel_fp8(v, smooth_v=False))
    n = k.numel()
    print(f"| ({B},{H},{N},{D}) | {tk:.1f} | {5*n/tk/1e3:.0f} | {tq:.1f} | {3*n/tq/1e3:.0f} | {tv:.1f} | {5*n/tv/1e3:.0f} |", flush=True)
