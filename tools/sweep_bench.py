#!/usr/bin/env python3
"""TFLOPS of the INT8-QK/FP16-PV and INT8-QK/FP8-PV operators and of FA2-ROCm (torch SDPA, flash
backend) at head_dim {64,128}, seqlen 1K..16K, causal and not; B*H = 128 (B = 4, H = 32) as the reference's benches
(bench/bench_baseline.py:9-11), plus the reference's video shapes: (1,30,8866,64) (utils/reado.py:9, CogVideoX-2b per
rank, ragged) and (2,48,8192,128) without and with per-channel K outliers (k += 4*randn(1,H,1,D)).

Two measurements per operator, as two markdown tables on stdout:
  kernel-only  the attention kernel alone on pre-quantized operands -- the form of the reference's published numbers
               (bench/bench_qk_int8_pv_fp16_cuda.py:36-59; "excluding the quantization and smoothing", bench/README.md:63).
               The operands are the real quantizer outputs of the same random fp16 tensors, not randint.
  end-to-end   the public entry point: K mean, quantizers (Q folded into the kernel up to 4096 rows), attention.
TFLOPS = 4*B*H*N^2*D/t (/2 causal), bench/bench_baseline.py:31.  Speedups are against FA2-ROCm on the same fp16 tensors.
``--flush``: every timed call starts from cold caches (see timeit)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sageattention_amd as sa
from sageattention_amd import _lib as L, _qattn, core
from torch.nn.attention import SDPBackend, sdpa_kernel


FLUSH = "--flush" in sys.argv   # cold caches: overwrite a 512 MiB buffer (L2 + 256 MiB Infinity Cache) before EVERY timed call,
_flush_buf = None               # as the reference's bench helper does (bench/utils.py:9-12); one event pair per call


def timeit(f, n):
    global _flush_buf
    if FLUSH:
        if _flush_buf is None:
            _flush_buf = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
        for _ in range(2): f()
        ts = []
        for _ in range(max(5, min(n, 15))):
            _flush_buf.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return statistics.median(ts)
    for _ in range(3): f()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n)
    return statistics.median(ts)


def measure(B, H, N, D, causal, kbias=False):
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.float16, device="cuda") for _ in range(3))
    if kbias:
        k = k + 4 * torch.randn(1, H, 1, D, dtype=torch.float16, device="cuda")
    fl = 4.0 * B * H * N * N * D / (2 if causal else 1)
    n = max(3, min(50, int(2e12 / fl * 20)))
    with sdpa_kernel(SDPBackend.FLASH_ATTENTION):
        t_fa = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=causal), n)
    e16 = timeit(lambda: sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v, is_causal=causal), n)
    e8 = timeit(lambda: sa.sageattn_qk_int8_pv_fp8_cuda(q, k, v, is_causal=causal), n)
    km = sa.quant.k_mean(k)
    q8, qs, k8, ks, _ = core._quant_qk(q, k, km, "HND", "per_thread", D ** -0.5, 32, False, H, H)
    o = torch.empty_like(q)
    k16 = timeit(lambda: _qattn._attn_f16(q8, k8, v, o, qs, ks, None, 1, int(causal), L.GRAN_PER_THREAD, D ** -0.5, 0), n)
    v8, vsc, _ = sa.quant.per_channel_fp8(v, smooth_v=False)
    k8t = timeit(lambda: _qattn._attn_f8(q8, k8, v8, o, qs, ks, vsc, None, 1, int(causal), L.GRAN_PER_THREAD, D ** -0.5, 0), n)
    return fl, t_fa, k16, k8t, e16, e8


GRID = [(4, 32, N, D, c, False) for D in (64, 128) for N in (1024, 2048, 4096, 8192, 16384) for c in (False, True)]
VIDEO = [(1, 30, 8866, 64, False, False), (2, 48, 8192, 128, False, False), (2, 48, 8192, 128, False, True)]
rows = []
for cfg in GRID + VIDEO:
    rows.append((cfg, measure(*cfg)))
    print("#", cfg, ["%.4f" % x for x in rows[-1][1][1:]], file=sys.stderr, flush=True)
tf = lambda fl, t: fl / t / 1e9
for title, i16, i8 in (("kernel-only (pre-quantized operands)", 2, 3), ("end-to-end (quantizers included)", 4, 5)):
    print(f"\n### {title}\n")
    print("| shape (B,H,N,D) | causal | FA2-ROCm | INT8/FP16-PV | x FA2 | of 3333 | INT8/FP8-PV | x FA2 | of 5000 |")
    print("|---|---|---|---|---|---|---|---|---|")
    for (B, H, N, D, c, kb), m in rows:
        fl, t_fa = m[0], m[1]
        name = f"({B},{H},{N},{D})" + (" K outliers" if kb else "")
        print(f"| {name} | {int(c)} | {tf(fl, t_fa):.0f} | {tf(fl, m[i16]):.0f} | {t_fa / m[i16]:.2f} | {tf(fl, m[i16]) / 3333:.2f} | "
              f"{tf(fl, m[i8]):.0f} | {t_fa / m[i8]:.2f} | {tf(fl, m[i8]) / 5000:.2f} |")
