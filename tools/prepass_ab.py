#!/usr/bin/env python3
"""K pre-pass (smoothing mean + INT8 quantizer) of short sequences: the one-launch form (k_onepass_kernel, default where it
applies) against the two-launch form (SAGE_K_ONEPASS=0), each in its own process: prepass_ab.py runs itself twice."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import statistics, torch
    from sageattention_amd import _lib as L
    from sageattention_amd.quant import k_smooth_quant
    for (B, H, N, D) in [(4, 32, 512, 64), (4, 32, 1024, 64), (4, 32, 2048, 64), (4, 32, 512, 128), (4, 32, 1024, 128), (8, 32, 1024, 64), (2, 48, 1024, 128)]:
        k = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
        ts = []
        for r in range(7):
            for _ in range(3): k_smooth_quant(k, "HND", L.GRAN_PER_THREAD, L.ROUND_TRITON)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): k_smooth_quant(k, "HND", L.GRAN_PER_THREAD, L.ROUND_TRITON)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 50 * 1e3)
        print(f"({B},{H},{N},{D}) {statistics.median(ts):.1f} us", flush=True)
else:
    for mode in ("1", "0"):
        print(f"SAGE_K_ONEPASS={mode}", flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, SAGE_K_ONEPASS=mode), check=True)
