#!/usr/bin/env python3
"""Does the operator capture into a HIP graph, and what does replay save on a launch-bound shape?  (C2 by default)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sageattention_amd as sa
B, H, N, D = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4,32,2048,64").split(","))
pv = sys.argv[2] if len(sys.argv) > 2 else "fp16"
fn = sa.sageattn_qk_int8_pv_fp16_cuda if pv == "fp16" else sa.sageattn_qk_int8_pv_fp8_cuda
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, N, D, dtype=torch.float16, device="cuda") for _ in range(3))
for _ in range(3): o_eager = fn(q, k, v)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    o_graph = fn(q, k, v)
g.replay(); torch.cuda.synchronize()
print("graph output equals eager:", torch.equal(o_graph, o_eager))
def timeit(f, n=50):
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n)
    return statistics.median(ts)
te, tg = timeit(lambda: fn(q, k, v)), timeit(g.replay)
fl = 4.0 * B * H * N * N * D
print(f"eager {te*1e3:.1f} us ({fl/te/1e9:.0f} TFLOPS)   graph replay {tg*1e3:.1f} us ({fl/tg/1e9:.0f} TFLOPS)")
