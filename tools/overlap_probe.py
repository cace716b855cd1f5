#!/usr/bin/env python3
"""Can the HBM-bound pre-pass of one half of the heads hide under the compute-bound attention kernel of the other half?
Probe: attention (pre-quantized operands, heads [0,H/2)) on the main stream and the K/Q quantizers of heads [H/2,H) on a
side stream, against the same two pieces run back to back."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sageattention_amd as sa
from sageattention_amd import _lib as L, _qattn, core
B, H, N, D = 4, 32, 8192, 128
pv = sys.argv[1] if len(sys.argv) > 1 else "fp16"
q, k, v = (torch.randn(B, H, N, D, dtype=torch.float16, device="cuda") for _ in range(3))
h2 = H // 2
qa, ka, va = q[:, :h2], k[:, :h2], v[:, :h2]
qb, kb, vb = q[:, h2:], k[:, h2:], v[:, h2:]
km = sa.quant.k_mean(ka.contiguous())
q8, qs, k8, ks, _ = core._quant_qk(qa.contiguous(), ka.contiguous(), km, "HND", "per_thread", D ** -0.5, 32, False, h2, h2)
o = torch.empty(B, h2, N, D, dtype=torch.float16, device="cuda")
vac = va.contiguous()
if pv == "fp8":
    v8, vsc, _ = sa.quant.per_channel_fp8(vac, smooth_v=False)
def attn():
    if pv == "fp16":
        _qattn._attn_f16(q8, k8, vac, o, qs, ks, None, 1, 0, L.GRAN_PER_THREAD, D ** -0.5, 0)
    else:
        _qattn._attn_f8(q8, k8, v8, o, qs, ks, vsc, None, 1, 0, L.GRAN_PER_THREAD, D ** -0.5, 0)
qbc, kbc, vbc = qb.contiguous(), kb.contiguous(), vb.contiguous()
def prepass():
    kmb = sa.quant.k_mean(kbc)
    core._quant_qk(qbc, kbc, kmb, "HND", "per_thread", D ** -0.5, 32, False, h2, h2)
    if pv == "fp8":
        sa.quant.per_channel_fp8(vbc, smooth_v=False)
side = torch.cuda.Stream()
def timed(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)
def serial():
    attn(); prepass()
def overlapped():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        prepass()
    attn()
    main.wait_stream(side)
def overlapped_pre_first():   # side work submitted first, so its workgroups are resident before the attention grid arrives
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        prepass()
    attn()
    main.wait_stream(side)
ta, tp = timed(attn), timed(prepass)
ts, to = timed(serial), timed(overlapped)
print(f"{pv}: attention {ta*1e3:.0f} us, pre-pass {tp*1e3:.0f} us, back to back {ts*1e3:.0f} us, overlapped {to*1e3:.0f} us "
      f"({(ts-to)/tp*100:.0f} % of the pre-pass hidden)")
