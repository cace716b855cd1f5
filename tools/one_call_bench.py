#!/usr/bin/env python3
"""End-to-end time of the public operators through the one-call C-ABI operator (core.ONE_CALL, the default) and through
the multi-call path (3-4 ctypes crossings, 6-9 allocations), interleaved in one process; FA2-ROCm beside them."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import sageattention_amd as sa
from sageattention_amd import core
shapes = [(4, 32, 1024, 64, False), (4, 32, 2048, 64, False), (4, 32, 1024, 128, False), (4, 32, 2048, 128, False),
          (4, 32, 4096, 128, True), (4, 32, 8192, 128, False)]
print("| shape | operator | one call us | multi call us | speed-up | FA2-ROCm us | one call x FA2 |")
print("|---|---|---|---|---|---|---|")
for (B, H, N, D, causal) in shapes:
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.float16, device="cuda") for _ in range(3))
    fl = 4.0 * B * H * N * N * D / (2 if causal else 1)
    n = max(5, int(30e-3 / (fl / 0.9e15)))
    def timed(fn):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    with torch.nn.attention.sdpa_kernel(torch.nn.attention.SDPBackend.FLASH_ATTENTION):
        t_fa = statistics.median(timed(lambda: F.scaled_dot_product_attention(q, k, v, is_causal=causal)) for _ in range(3))
    for name, fn in (("fp16", sa.sageattn_qk_int8_pv_fp16_cuda), ("fp8", sa.sageattn_qk_int8_pv_fp8_cuda)):
        res = {True: [], False: []}
        for rnd in range(5):
            for one in (True, False):
                core.ONE_CALL = one
                res[one].append(timed(lambda: fn(q, k, v, is_causal=causal)))
        core.ONE_CALL = True
        a, b = statistics.median(res[True]), statistics.median(res[False])
        print(f"| ({B},{H},{N},{D}){' causal' if causal else ''} | {name} | {a:.1f} | {b:.1f} | {b / a:.3f} | {t_fa:.1f} | {t_fa / a:.2f} |", flush=True)
