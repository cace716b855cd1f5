#!/usr/bin/env python3
"""The public operators end to end on short shapes, for `rocprofv3 --kernel-trace`: e2e_trace.py [calls]
Each shape runs `calls` back-to-back operator calls (after 3 warm-ups) bracketed by a marker kernel (a 1-element fill) so
that tools/e2e_trace_summary.py can cut the trace into shapes.  Prints the shape list in the order it ran."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sageattention_amd as sa
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 40
shapes = [(4, 32, 1024, 64, False), (4, 32, 2048, 64, False), (4, 32, 1024, 128, False), (4, 32, 2048, 128, False),
          (4, 32, 1024, 128, True), (4, 32, 4096, 128, True)]
marker = torch.zeros(1, device="cuda", dtype=torch.float64)
for (B, H, N, D, causal) in shapes:
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.float16, device="cuda") for _ in range(3))
    for name, fn in (("fp16", sa.sageattn_qk_int8_pv_fp16_cuda), ("fp8", sa.sageattn_qk_int8_pv_fp8_cuda)):
        for _ in range(3): fn(q, k, v, is_causal=causal)
        torch.cuda.synchronize()
        marker.fill_(1.0)  # float64 fill: the only such kernel in the trace
        for _ in range(calls): fn(q, k, v, is_causal=causal)
        torch.cuda.synchronize()
        print(f"SHAPE ({B},{H},{N},{D}){' causal' if causal else ''} {name}", flush=True)
