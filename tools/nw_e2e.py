#!/usr/bin/env python3
"""End-to-end operator time by workgroup geometry (4 or 8 waves) -- the kernel-only crossover of sage_attn.hip::run_attn was
measured on pre-quantized operands; with the Q quantizer folded into the prologue the prologue weighs more."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sageattention_amd as sa
from sageattention_amd import _lib as L
shapes = [(4, 32, 2048, 128, False), (4, 32, 3072, 128, False), (4, 32, 4096, 128, False), (4, 32, 6144, 128, False), (4, 32, 8192, 128, False),
          (4, 32, 8192, 128, True), (4, 32, 16384, 128, True), (2, 32, 16384, 128, False)]
for (B, H, N, D, causal) in shapes:
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.float16, device="cuda") for _ in range(3))
    res = {4: [], 8: []}
    fl = 4.0 * B * H * N * N * D / (2 if causal else 1)
    n = max(3, int(30e-3 / (fl / 1.2e15)))
    for rnd in range(5):
        for nw in (4, 8):
            L.lib().sage_set_tuning(0, nw)
            for _ in range(2): sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v, is_causal=causal)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v, is_causal=causal)
            e1.record(); torch.cuda.synchronize()
            res[nw].append(e0.elapsed_time(e1) / n)
    L.lib().sage_set_tuning(0, 0)
    a, b = statistics.median(res[4]), statistics.median(res[8])
    print(f"{(B,H,N,D,causal)} fp16 PV end to end: 4 waves {a*1e3:8.1f} us ({fl/a/1e9:.0f} TF)  8 waves {b*1e3:8.1f} us ({fl/b/1e9:.0f} TF)  4w/8w {a/b:.3f}", flush=True)
