#!/usr/bin/env python3
"""end-to-end TFLOPS of the INT8-QK/FP16-PV and INT8-QK/FP8-PV operators and of FA2-ROCm (torch SDPA,
flash backend) at head_dim {64,128}, seqlen 1K..16K, causal and not; B*H = 128 (B = 4, H = 32), random fp16 inputs.
Writes a markdown table to stdout."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sageattention_amd as sa
from torch.nn.attention import SDPBackend, sdpa_kernel

def timeit(f, n):
    for _ in range(3): f()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n)
    return statistics.median(ts)

B, H = 4, 32
print("| head_dim | seqlen | causal | FA2-ROCm | INT8/FP16-PV | x FA2 | of 3333 | INT8/FP8-PV | x FA2 | of 5000 |")
print("|---|---|---|---|---|---|---|---|---|---|")
for D in (64, 128):
    for N in (1024, 2048, 4096, 8192, 16384):
        for causal in (False, True):
            torch.manual_seed(0)
            q, k, v = (torch.randn(B, H, N, D, dtype=torch.float16, device="cuda") for _ in range(3))
            fl = 4.0 * B * H * N * N * D / (2 if causal else 1)
            n = max(3, min(50, int(2e12 / fl * 20)))
            with sdpa_kernel(SDPBackend.FLASH_ATTENTION):
                t_fa = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=causal), n)
            t16 = timeit(lambda: sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v, is_causal=causal), n)
            t8 = timeit(lambda: sa.sageattn_qk_int8_pv_fp8_cuda(q, k, v, is_causal=causal), n)
            tf = lambda t: fl / t / 1e9
            print(f"| {D} | {N} | {int(causal)} | {tf(t_fa):.0f} | {tf(t16):.0f} | {t_fa/t16:.2f} | {tf(t16)/3333:.2f} | {tf(t8):.0f} | {t_fa/t8:.2f} | {tf(t8)/5000:.2f} |", flush=True)
