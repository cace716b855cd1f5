#!/usr/bin/env python3
"""End-to-end time of the operators with the Q quantizer folded into the attention kernel or as its own launch
(core.FUSE_Q_MAX_SEQ decides per call): where is the crossover?"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sageattention_amd as sa
from sageattention_amd import core
shapes = [(4, 32, 2048, 128, False), (4, 32, 4096, 128, False), (4, 32, 8192, 128, False), (4, 32, 8192, 128, True),
          (4, 32, 16384, 128, True), (4, 32, 8192, 64, False), (4, 32, 16384, 64, False)]
for (B, H, N, D, causal) in shapes:
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.float16, device="cuda") for _ in range(3))
    for name, fn in (("fp16", sa.sageattn_qk_int8_pv_fp16_cuda), ("fp8", sa.sageattn_qk_int8_pv_fp8_cuda)):
        res = {}
        for rnd in range(5):
            for mode, lim in (("fused", 1 << 30), ("separate", 0)):
                core.FUSE_Q_MAX_SEQ = lim
                for _ in range(2): fn(q, k, v, is_causal=causal)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = max(3, int(20e-3 / (4.0 * B * H * N * N * D / (2 if causal else 1) / 1.2e15)))
                e0.record()
                for _ in range(n): fn(q, k, v, is_causal=causal)
                e1.record(); torch.cuda.synchronize()
                res.setdefault(mode, []).append(e0.elapsed_time(e1) / n)
        f, s = statistics.median(res["fused"]), statistics.median(res["separate"])
        print(f"{(B,H,N,D,causal)} {name}: fused {f*1e3:8.1f} us  separate {s*1e3:8.1f} us  fused/separate {f/s:.3f}", flush=True)
