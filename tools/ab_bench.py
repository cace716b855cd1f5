#!/usr/bin/env python3
"""In-process A/B of several builds of libsageattn_hip.so (cdna guide rule 24: interleaved rounds in ONE process).
usage: ab_bench.py [--wl c3] [--rounds 7] [--iters 5] lib_a.so lib_b.so ...   (NWAVES via name suffix @4/@8)"""
import argparse, ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sageattention_amd as sa
from sageattention_amd import _lib as L, core

ap = argparse.ArgumentParser()
ap.add_argument("--wl", default="c3")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--data", default="randn", choices=["randn", "zeros"], help="zeros: DVFS probe (cdna guide rule 25)")
ap.add_argument("--pv", default="fp16", choices=["fp16", "fp8"])
ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"], help="element type of q/k/v/o (bf16: the kernel converts V tiles to fp16 on the fly)")
ap.add_argument("--share-kv", action="store_true", help="latency probe: every head reads the K/V of head 0 (stride 0): all tiles L2 hits")
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
B, H, N, D, causal = {"c2": (4, 32, 2048, 64, False), "c3": (4, 32, 8192, 128, False), "c3c": (4, 32, 8192, 128, True),
                      "c2c": (4, 32, 2048, 64, True), "c16k": (2, 32, 16384, 128, False),
                      "d64_8k": (4, 32, 8192, 64, False), "c3s": (1, 32, 8192, 128, False), "c3xs": (1, 8, 8192, 128, False),
                      "c3l": (8, 32, 8192, 128, False), "d128_1k": (16, 32, 1024, 128, False), "d128_1536": (8, 32, 1536, 128, False), "d128_2k": (8, 32, 2048, 128, False), "d128_3k": (4, 32, 3072, 128, False), "d128_4kc": (4, 32, 4096, 128, True), "d128_4k": (4, 32, 4096, 128, False), "d128_6k": (4, 32, 6144, 128, False), "c16kc": (2, 32, 16384, 128, True), "d128_2kc": (8, 32, 2048, 128, True),
                      "c4": (4, 32, 16384, 128, True), "c32k": (1, 32, 32768, 128, False), "c64k": (1, 16, 65536, 128, False)}[a.wl]
torch.manual_seed(0)
DT = torch.float16 if a.dtype == "fp16" else torch.bfloat16
EL = 0 if a.dtype == "fp16" else 1   # SAGE_F16 / SAGE_BF16
q = torch.randn(B, H, N, D, dtype=DT, device="cuda")
k = torch.randn(B, H, N, D, dtype=DT, device="cuda")
v = torch.randn(B, H, N, D, dtype=DT, device="cuda")
if a.data == "zeros":
    q.zero_(); k.zero_(); v.zero_()
km = sa.quant.k_mean(k)
q8, qs, k8, ks, _ = core._quant_qk(q, k, km, "HND", "per_thread", D ** -0.5, 32, False, H, H)
o = torch.empty_like(q)
if a.share_kv:
    k8 = k8[:, :1].expand(B, H, N, D)
    ks = ks[:, :1].expand(B, H, ks.shape[-1]).contiguous()
    v = v[:, :1].expand(B, H, N, D)
if a.pv == "fp8":
    v8, vs, _ = sa.quant.per_channel_fp8(v.contiguous(), tensor_layout="HND", smooth_v=False)
    if a.share_kv:
        v8 = v8[:, :1].expand(B, H, v8.shape[2], v8.shape[3])
    vd = L.SageTensor(v8.data_ptr(), v8.stride(0), v8.stride(1), v8.stride(2))
ref = None
variants = []
for spec in a.libs:
    path, nw = (spec.split("@") + [""])[:2]   # lib.so[@nwaves]
    l = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in L.SIGNATURES.items():
        if hasattr(l, name):
            fn = getattr(l, name); fn.restype, fn.argtypes = res, args
    variants.append((spec, l, int(nw) if nw else 0))
st = torch.cuda.current_stream().cuda_stream
def run(l, nw):
    l.sage_set_tuning(0, nw)
    if a.pv == "fp8":
        r = l.sage_attn_qk_int8_pv_f8(L.desc(q8, "HND"), L.desc(k8, "HND"), vd, L.desc(o, "HND"), EL, qs.data_ptr(),
                                      ks.data_ptr(), vs.data_ptr(), None, None, B, H, H, N, N, D, int(causal), 3, 128, 32,
                                      D ** -0.5, 0, st)
        assert r == 0, r
        return
    r = l.sage_attn_qk_int8_pv_f16(L.desc(q8, "HND"), L.desc(k8, "HND"), L.desc(v, "HND"), EL, L.desc(o, "HND"), EL,
                                   qs.data_ptr(), ks.data_ptr(), None, None, B, H, H, N, N, D, int(causal), 3, 128, 32,
                                   D ** -0.5, 0, st)
    assert r == 0, r
times = {s: [] for s, _, _ in variants}
for s, l, nw in variants:  # warmup + consistency
    run(l, nw); torch.cuda.synchronize()
    cur = o.float().clone()
    if ref is None: ref = cur
    print(s, "max|o - first variant| =", (cur - ref).abs().max().item())
for r in range(a.rounds):
    for s, l, nw in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters): run(l, nw)
        e1.record(); torch.cuda.synchronize()
        times[s].append(e0.elapsed_time(e1) / a.iters)
fl = 4.0 * B * H * N * N * D / (2 if causal else 1)
for s, t in times.items():
    print(f"{s:50s} median {statistics.median(t):.4f} ms  min {min(t):.4f} ms  -> {fl/statistics.median(t)/1e9:.1f} TFLOPS (median)")
