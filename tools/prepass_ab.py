#!/usr/bin/env python3
"""K (and K+V) pre-pass of several builds of the library, interleaved in one process, outputs compared bit for bit with the
first build: prepass_ab.py [--dtype fp16|bf16] [--gran thread|block|warp] lib_a.so lib_b.so ...
Times sage_k_smooth_quant (the FP16-PV operators' K pre-pass) and sage_kv_prepare_fp8 (the FP8-PV operator's K + V pre-pass)."""
import argparse, ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sageattention_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
ap.add_argument("--gran", default="thread", choices=["thread", "block", "warp"])
ap.add_argument("--once", action="store_true", help="one call per build and shape, no timing (for rocprofv3 --pmc)")
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
DT = torch.float16 if a.dtype == "fp16" else torch.bfloat16
gran, rounding = {"thread": (L.GRAN_PER_THREAD, L.ROUND_TRITON), "block": (L.GRAN_PER_BLOCK, L.ROUND_TRITON),
                  "warp": (L.GRAN_PER_BLOCK, L.ROUND_CUDA)}[a.gran]  # (per_warp quantizes K per block, CUDA numerics)
libs = []
for path in a.libs:
    l = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in L.SIGNATURES.items():
        if hasattr(l, name):
            fn = getattr(l, name); fn.restype, fn.argtypes = res, args
    libs.append((path, l))
st = torch.cuda.current_stream().cuda_stream
shapes = [(4, 32, 1024, 64), (4, 32, 2048, 64), (4, 32, 1024, 128), (4, 32, 2048, 128), (4, 32, 4096, 128), (4, 32, 8192, 128),
          (4, 32, 16384, 128), (2, 30, 1000, 64)]
print("| shape | " + " | ".join(f"{os.path.basename(p)} K us | K+V fp8 us" for p, _ in libs) + " | identical |")
print("|---|" + "---|---|" * len(libs) + "---|")
for (B, H, N, D) in shapes:
    torch.manual_seed(0)
    k = torch.randn(B, H, N, D, dtype=DT, device="cuda") + torch.randn(B, H, 1, D, dtype=DT, device="cuda")
    v = torch.randn(B, H, N, D, dtype=DT, device="cuda")
    G = (N + 63) // 64 * (4 if gran == L.GRAN_PER_THREAD else 1)
    npad = (N + 63) // 64 * 64
    outs, times = [], []
    for path, l in libs:
        k8 = torch.zeros(B, H, N, D, dtype=torch.int8, device="cuda")
        ks = torch.zeros(B, H, G, dtype=torch.float32, device="cuda")
        km = torch.zeros(B, H, D, dtype=DT, device="cuda")
        v8 = torch.zeros(B, H, D, npad, dtype=torch.uint8, device="cuda")
        vs = torch.zeros(B, H, D, dtype=torch.float32, device="cuda")
        ws = torch.empty(max(1, l.sage_kv_prepare_fp8_workspace_bytes(B, H, N, D) // 4), dtype=torch.float32, device="cuda")
        vd = L.SageTensor(v8.data_ptr(), v8.stride(0), v8.stride(1), v8.stride(2))
        def run_k(l=l, k8=k8, ks=ks, km=km, ws=ws):   # (defaults bind THIS build's objects: a plain closure would see the last build's)
            r = l.sage_k_smooth_quant(L.desc(k, "HND"), L.dtype_code(DT), B, H, N, D, L.desc(k8, "HND"), ks.data_ptr(),
                                      km.data_ptr(), gran, rounding, ws.data_ptr(), st)
            assert r == 0, r
        def run_kv(l=l, k8=k8, ks=ks, km=km, ws=ws, vd=vd, vs=vs):
            r = l.sage_kv_prepare_fp8(L.desc(k, "HND"), L.desc(v, "HND"), L.dtype_code(DT), B, H, N, D, L.desc(k8, "HND"),
                                      ks.data_ptr(), km.data_ptr(), gran, rounding, vd, vs.data_ptr(), 448.0, ws.data_ptr(), st)
            assert r == 0, r
        run_k(); torch.cuda.synchronize()
        o1 = (k8.clone(), ks.clone(), km.clone())
        k8.zero_(); ks.zero_(); km.zero_()
        run_kv(); torch.cuda.synchronize()
        outs.append(o1 + (k8.clone(), ks.clone(), km.clone(), v8.clone(), vs.clone()))
        times.append((run_k, run_kv, [], []))
    names = ("k8", "k_scale", "km", "kv:k8", "kv:k_scale", "kv:km", "kv:v8", "kv:v_scale")
    diff = [f"{os.path.basename(libs[i + 1][0])}:{n}" for i, o in enumerate(outs[1:]) for n, x, y in zip(names, outs[0], o) if not torch.equal(x, y)]
    same = not diff if not diff else "DIFFERENT: " + ",".join(diff)
    if a.once:
        print(f"| ({B},{H},{N},{D}) | {same} |", flush=True)
        continue
    n = max(10, int(2e9 / (B * H * N * D * 5)))
    for rnd in range(5):
        for run_k, run_kv, tk, tkv in times:
            for f, acc in ((run_k, tk), (run_kv, tkv)):
                for _ in range(3): f()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n): f()
                e1.record(); torch.cuda.synchronize()
                acc.append(e0.elapsed_time(e1) / n * 1e3)
    print(f"| ({B},{H},{N},{D}) | " + " | ".join(f"{statistics.median(tk):.1f} | {statistics.median(tkv):.1f}" for _, _, tk, tkv in times)
          + f" | {same} |", flush=True)
