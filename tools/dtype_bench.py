#!/usr/bin/env python3
"""fp16 vs bf16 inputs, end to end (public entry points) and FA2-ROCm on the same tensors: markdown table on stdout
(`profiles/r02_dtype.md`).  Video models run in bf16; since the end of round 2 a bf16 V is multiplied as bf16
(v_mfma_f32_32x32x16_bf16) instead of being converted to fp16 tile by tile inside the kernel."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sageattention_amd as sa
from torch.nn.attention import SDPBackend, sdpa_kernel


def timeit(f, flop):
    for _ in range(3): f()
    n = max(3, min(200, int(30e-3 / (flop / 1.0e15))))
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return statistics.median(ts)


shapes = [(4, 32, 2048, 64, False), (4, 32, 8192, 64, False), (1, 30, 8866, 64, False), (4, 32, 2048, 128, False),
          (4, 32, 8192, 128, False), (4, 32, 8192, 128, True), (2, 48, 8192, 128, False)]
print("| shape (B,H,N,D) | causal | dtype | FA2-ROCm | INT8/FP16-PV (16-bit PV) | x FA2 | INT8/FP8-PV | x FA2 |")
print("|---|---|---|---|---|---|---|---|")
torch.manual_seed(0)
for (B, H, N, D, causal) in shapes:
    flop = 4.0 * B * H * N * N * D / (2 if causal else 1)
    for dt, name in ((torch.float16, "fp16"), (torch.bfloat16, "bf16")):
        q, k, v = (torch.randn(B, H, N, D, dtype=dt, device="cuda") for _ in range(3))
        with sdpa_kernel(SDPBackend.FLASH_ATTENTION):
            fa = flop / timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=causal), flop) / 1e9
        a = flop / timeit(lambda: sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v, is_causal=causal), flop) / 1e9
        b = flop / timeit(lambda: sa.sageattn_qk_int8_pv_fp8_cuda(q, k, v, is_causal=causal), flop) / 1e9
        print(f"| ({B},{H},{N},{D}) | {int(causal)} | {name} | {fa:.0f} | {a:.0f} | {a / fa:.2f} | {b:.0f} | {b / fa:.2f} |", flush=True)
