import os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch
import sageattention_amd as sa
from sageattention_amd import _lib as L, _qattn, core
B, H, M, N, D = 1, 32, 8192, 65536, 128
torch.manual_seed(0)
q = torch.randn(B, H, M, D, dtype=torch.float16, device="cuda")
k = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
v = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
km = sa.quant.k_mean(k)
q8, qs, k8, ks, _ = core._quant_qk(q, k, km, "HND", "per_thread", D ** -0.5, 32, False, H, H)
o = torch.empty_like(q)
v8, vsc, _ = sa.quant.per_channel_fp8(v, smooth_v=False)
res = {4: [], 8: []}
for rnd in range(5):
    for nw in (4, 8):
        L.lib().sage_set_tuning(0, nw)
        for _ in range(2): _qattn._attn_f8(q8, k8, v8, o, qs, ks, vsc, None, 1, 0, 3, D ** -0.5, 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): _qattn._attn_f8(q8, k8, v8, o, qs, ks, vsc, None, 1, 0, 3, D ** -0.5, 0)
        e1.record(); torch.cuda.synchronize()
        res[nw].append(e0.elapsed_time(e1) / 5)
L.lib().sage_set_tuning(0, 0)
fl = 4.0 * B * H * M * N * D
for nw in (4, 8): print(f"c5r per-rank launch (1,32,M=8192,N=65536,128) fp8, {nw} waves: median {statistics.median(res[nw]):.3f} ms = {fl / statistics.median(res[nw]) / 1e9:.0f} TFLOP/s")
