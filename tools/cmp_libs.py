#!/usr/bin/env python3
"""Compare the attention output of several library builds against fp32 SDPA on one workload (debug aid)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sageattention_amd as sa
from sageattention_amd import _lib as L, core
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="4,32,2048,64")
ap.add_argument("--causal", type=int, default=1)
ap.add_argument("--pv", default="fp16")
ap.add_argument("--runs", type=int, default=30)
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
B, H, N, D = (int(x) for x in a.shape.split(","))
causal = bool(a.causal)
torch.manual_seed(0)
q = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
k = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
v = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
km = sa.quant.k_mean(k)
q8, qs, k8, ks, _ = core._quant_qk(q, k, km, "HND", "per_thread", D ** -0.5, 32, False, H, H)
ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float(), is_causal=causal)
st = torch.cuda.current_stream().cuda_stream
if a.pv == "fp8":
    v8, vs, _ = sa.quant.per_channel_fp8(v, tensor_layout="HND", smooth_v=False)
    vd = L.SageTensor(v8.data_ptr(), v8.stride(0), v8.stride(1), v8.stride(2))
def call(l, o):
    if a.pv == "fp8":
        return l.sage_attn_qk_int8_pv_f8(L.desc(q8, "HND"), L.desc(k8, "HND"), vd, L.desc(o, "HND"), 0, qs.data_ptr(),
                                         ks.data_ptr(), vs.data_ptr(), None, None, B, H, H, N, N, D, int(causal), 3, 128, 32,
                                         D ** -0.5, 0, st)
    return l.sage_attn_qk_int8_pv_f16(L.desc(q8, "HND"), L.desc(k8, "HND"), L.desc(v, "HND"), 0, L.desc(o, "HND"), 0,
                                      qs.data_ptr(), ks.data_ptr(), None, None, B, H, H, N, N, D, int(causal), 3, 128, 32,
                                      D ** -0.5, 0, st)
for path in a.libs:
    l = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in L.SIGNATURES.items():
        if hasattr(l, name):
            fn = getattr(l, name); fn.restype, fn.argtypes = res, args
    o = torch.empty_like(q)
    r = call(l, o)
    torch.cuda.synchronize()
    first = o.clone()
    nd = 0
    for it in range(a.runs):
        o.zero_()
        call(l, o)
        torch.cuda.synchronize()
        if not torch.equal(o, first):
            dd = (o.float() - first.float()).abs()
            bad = (dd > 0).nonzero()
            nd += 1
            if nd <= 3:
                print("  run", it, "differs: max", dd.max().item(), "n", bad.shape[0], "first idx", bad[0].tolist(), "rows", sorted(set(bad[:, 2].tolist()))[:10])
    print("  nondeterministic runs:", nd, "/", a.runs)
    d = (o.float() - ref).abs()
    idx = (d == d.max()).nonzero()[0].tolist()
    bad = (d > 0.05).nonzero()
    print(path, "status", r, "max|o-ref|", d.max().item(), "at", idx, "count>0.05:", bad.shape[0],
          "rows:", sorted(set(bad[:, 2].tolist()))[:12], "nan:", torch.isnan(o).sum().item())
