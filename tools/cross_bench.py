#!/usr/bin/env python3
"""Cross-attention shapes (many query rows, few keys: Wan / Hunyuan text and image conditioning): fused-Q on/off."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sageattention_amd as sa
from sageattention_amd import core
def timeit(f, n=20):
    for _ in range(3): f()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n)
    return statistics.median(ts)
keep = core.FUSE_Q_MAX_SEQ
for (B, H, M, N, D) in [(1, 40, 32760, 512, 128), (1, 40, 32760, 257, 128), (2, 24, 16384, 256, 64), (1, 40, 32760, 2048, 128),
                        (1, 40, 32760, 4096, 128), (4, 32, 8192, 1024, 128)]:
    q = torch.randn(B, H, M, D, dtype=torch.float16, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
    fl = 4.0 * B * H * M * N * D
    out = []
    for name, fn in (("fp16", sa.sageattn_qk_int8_pv_fp16_cuda), ("fp8", sa.sageattn_qk_int8_pv_fp8_cuda)):
        core.FUSE_Q_MAX_SEQ = 1 << 30
        t1 = timeit(lambda: fn(q, k, v))
        core.FUSE_Q_MAX_SEQ = 0
        t0 = timeit(lambda: fn(q, k, v))
        out.append(f"{name}: unfused {t0*1e3:.0f} us ({fl/t0/1e9:.0f} TF) fused {t1*1e3:.0f} us ({fl/t1/1e9:.0f} TF) {t0/t1:.2f}x")
    with torch.nn.attention.sdpa_kernel(torch.nn.attention.SDPBackend.FLASH_ATTENTION):
        tf = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v))
    print((B, H, M, N, D), " | ".join(out), f"| FA2 {tf*1e3:.0f} us", flush=True)
core.FUSE_Q_MAX_SEQ = keep
