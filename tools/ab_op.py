#!/usr/bin/env python3
"""The one-call operators (sage_sageattn_pv_f16 / _f8: K pre-pass + fused-Q attention) of several builds of the library,
interleaved in one process, HIP events, outputs compared bit for bit with the first build:
  ab_op.py [--dtype fp16|bf16] [--gran per_thread|per_warp] [--rounds 7] lib_a.so lib_b.so ..."""
import argparse, ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sageattention_amd import _lib as L
from sageattention_amd.core import _GRAN_CODE

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
ap.add_argument("--gran", default="per_thread", choices=["per_thread", "per_warp"])
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
DT = torch.float16 if a.dtype == "fp16" else torch.bfloat16
libs = []
for path in a.libs:
    l = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in L.SIGNATURES.items():
        if hasattr(l, name):
            fn = getattr(l, name); fn.restype, fn.argtypes = res, args
    libs.append((os.path.basename(path), l))
st = torch.cuda.current_stream().cuda_stream
shapes = [(4, 32, 1024, 64, False), (4, 32, 2048, 64, False), (4, 32, 1024, 128, False), (4, 32, 1024, 128, True),
          (4, 32, 2048, 128, False), (4, 32, 4096, 128, True), (4, 32, 8192, 128, False), (2, 30, 1000, 64, False)]
print("| shape | PV | " + " | ".join(f"{n} us" for n, _ in libs) + " | identical |")
print("|---|---|" + "---|" * len(libs) + "---|")
for (B, H, N, D, causal) in shapes:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, dtype=DT, device="cuda") for _ in range(3))
    k = k + torch.randn(B, H, 1, D, dtype=DT, device="cuda")
    fl = 4.0 * B * H * N * N * D / (2 if causal else 1)
    n = max(10, int(20e-3 / (fl / 0.8e15)))
    for pv_fp8 in (False, True):
        runs, outs = [], []
        for name, l in libs:
            opts = L.OpOpts(_GRAN_CODE[a.gran], 32, 1, 1, 0)
            nbytes = l.sage_sageattn_workspace_bytes(int(pv_fp8), B, H, H, N, N, D, 0, opts)
            ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            o = torch.zeros_like(q)
            def run(l=l, opts=opts, ws=ws, o=o, nbytes=nbytes):
                if pv_fp8:
                    r = l.sage_sageattn_pv_f8(L.desc(q, "HND"), L.desc(k, "HND"), L.desc(v, "HND"), L.dtype_code(DT), L.desc(o, "HND"),
                                              None, B, H, H, N, N, D, int(causal), D ** -0.5, 448.0, opts, ws.data_ptr(), nbytes, st)
                else:
                    r = l.sage_sageattn_pv_f16(L.desc(q, "HND"), L.desc(k, "HND"), L.desc(v, "HND"), L.dtype_code(DT), L.desc(o, "HND"),
                                               None, B, H, H, N, N, D, int(causal), D ** -0.5, opts, ws.data_ptr(), nbytes, st)
                assert r == 0, r
            run(); torch.cuda.synchronize()
            outs.append(o.clone())
            runs.append((run, []))
        for rnd in range(a.rounds):
            for run, acc in runs:
                for _ in range(3): run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n): run()
                e1.record(); torch.cuda.synchronize()
                acc.append(e0.elapsed_time(e1) / n * 1e3)
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        print(f"| ({B},{H},{N},{D}){' causal' if causal else ''} | {'fp8' if pv_fp8 else 'fp16'} | "
              + " | ".join(f"{statistics.median(acc):.1f}" for _, acc in runs) + f" | {same} |", flush=True)
