#!/usr/bin/env python3
"""Generate golden vectors by RUNNING the reference's own Triton kernels on CPU.

Usage (in the build container only; /root/reference does not exist on the GPU box):

    cd /root/repo && TRITON_INTERPRET=1 PYTHONDONTWRITEBYTECODE=1 \
        PYTHONPATH=/root/reference python oracle/gen_golden.py

Writes ``tests/golden/<case>.npz``.  A fixture is DATA only: the seeded inputs and the outputs
the reference produced for them (int8 tensors, fp32 scales, fp16/bf16 outputs stored as
uint16 bit patterns, base-2 LSE).  Upstream pairings only:
  per_block quantizer  <-> attn_qk_int8_per_block{,_causal}.forward
  per_thread quantizer <-> attn_qk_int8_per_thread.forward (non-causal only)
Reference call sites restated: sageattention/core.py:279-318.
"""
import io
import os
import sys
import contextlib

import numpy as np
import torch

assert os.environ.get("TRITON_INTERPRET") == "1", "run with TRITON_INTERPRET=1"

from sageattention.triton.quant_per_block import per_block_int8  # noqa: E402
from sageattention.triton.quant_per_thread import per_thread_int8  # noqa: E402
from sageattention.triton.attn_qk_int8_per_block import forward as attn_block  # noqa: E402
from sageattention.triton.attn_qk_int8_per_block_causal import forward as attn_block_causal  # noqa: E402
from sageattention.triton.attn_qk_int8_per_thread import forward as attn_thread  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

# name, B, Hq, Hk, M, N, D, layout, dtype, causal, k_bias
CASES = [
    ("c1_hnd", 2, 8, 8, 128, 128, 64, "HND", "fp16", False, 0.0),   # BASELINE configs[0]
    ("c1_nhd", 2, 8, 8, 128, 128, 64, "NHD", "fp16", False, 0.0),
    ("c1_causal", 2, 4, 4, 128, 128, 64, "HND", "fp16", True, 0.0),
    ("d128_ragged", 1, 2, 2, 192, 192, 128, "HND", "fp16", False, 2.0),  # N not a multiple of 128
    ("gqa_causal_320", 1, 8, 2, 320, 320, 64, "HND", "fp16", True, 1.0),
    ("cross_100x200", 1, 2, 2, 100, 200, 64, "HND", "fp16", False, 0.0),  # M != N, ragged both
    ("bf16_d128", 1, 2, 2, 256, 256, 128, "NHD", "bf16", False, 3.0),
    # round 2: head_dim 128 causal with several K tiles per q-block, and a bf16 / GQA / NHD / causal combination
    ("d128_causal_384", 1, 4, 2, 384, 384, 128, "HND", "fp16", True, 2.0),
    ("bf16_gqa_causal_nhd", 1, 4, 2, 256, 256, 64, "NHD", "bf16", True, 1.0),
]


def bits(x: torch.Tensor) -> np.ndarray:
    if x.dtype in (torch.float16, torch.bfloat16):
        return x.contiguous().view(torch.int16).numpy().view(np.uint16)
    return x.contiguous().numpy()


def run_quiet(fn, *a, **kw):
    # the reference's per-thread forward prints shapes (attn_qk_int8_per_thread.py:203-204)
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


def gen_varlen():
    """sageattn_varlen pairing (core.py:459-471): quant_per_block_varlen + attn_*_varlen."""
    from sageattention.triton.quant_per_block_varlen import per_block_int8 as pbv
    from sageattention.triton.attn_qk_int8_block_varlen import forward as attn_false_varlen
    from sageattention.triton.attn_qk_int8_per_block_causal_varlen import forward as attn_true_varlen
    torch.manual_seed(4242)
    lens = [100, 37, 200, 64]
    Hq, Hk, D = 4, 2, 64
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    T = sum(lens)
    q = torch.randn(T, Hq, D).half()
    k = (torch.randn(T, Hk, D) + 1.5 * torch.randn(1, Hk, D)).half()
    v = torch.randn(T, Hk, D).half()
    km = k.mean(dim=0, keepdim=True)      # core.py:461
    k2 = k - km
    sm = 1.0 / (D ** 0.5)
    q8, qs, k8, ks, cuqs, cuks = pbv(q, k2, cu, cu, max(lens), max(lens), sm_scale=sm)
    o = attn_false_varlen(q8, k8, v, cu, cu, max(lens), qs, ks, cuqs, cuks, output_dtype=torch.float16)
    oc = attn_true_varlen(q8, k8, v, cu, cu, max(lens), qs, ks, cuqs, cuks, output_dtype=torch.float16)
    path = os.path.join(OUT, "varlen", "varlen_d64.npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, q=bits(q), k=bits(k), v=bits(v), cu=cu.numpy(), q8=q8.numpy(), k8=k8.numpy(),
                        qs=qs.numpy(), ks=ks.numpy(), o=bits(o), o_causal=bits(oc))
    print(f"varlen_d64: wrote {os.path.getsize(path)/1024:.0f} KiB", flush=True)


def gen_masked():
    """attn_mask pairing of sageattn_qk_int8_pv_fp16_triton (core.py:306-318, upstream per-block pairing):
    bool mask (False -> -1e6, all-False tiles skipped) and additive mask in q's dtype."""
    torch.manual_seed(777)
    B, H, M, N, D = 1, 2, 200, 300, 64
    q = torch.randn(B, H, M, D).half()
    k = torch.randn(B, H, N, D).half()
    v = torch.randn(B, H, N, D).half()
    km = k.mean(dim=2, keepdim=True)
    sm = 1.0 / (D ** 0.5)
    qb, qsb, kb, ksb = per_block_int8(q, k, km=km, sm_scale=sm, tensor_layout="HND")
    mb = torch.rand(1, 1, M, N) > 0.3
    mb[..., :128, 64:192] = False            # whole 128x64 tiles masked out (tile skipping)
    mb[..., 5, :] = False                    # a fully masked row
    mb = mb.expand(B, H, M, N)
    mf = (torch.randn(B, 1, M, N) * 2).half().expand(B, H, M, N)
    ob, lb = attn_block(qb, kb, v, qsb, ksb, tensor_layout="HND", attn_mask=mb, output_dtype=torch.float16, return_lse=True)
    of, lf = attn_block(qb, kb, v, qsb, ksb, tensor_layout="HND", attn_mask=mf, output_dtype=torch.float16, return_lse=True)
    path = os.path.join(OUT, "masked", "masked_d64.npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, q=bits(q), k=bits(k), v=bits(v), km=bits(km), q8=qb.numpy(), k8=kb.numpy(), qs=qsb.numpy(),
                        ks=ksb.numpy(), mask_bool=mb[0, 0].numpy(), mask_float=bits(mf[:, 0].contiguous()),
                        o_bool=bits(ob), lse2_bool=lb.numpy(), o_float=bits(of), lse2_float=lf.numpy())
    print(f"masked_d64: wrote {os.path.getsize(path)/1024:.0f} KiB", flush=True)


def gen_masked_thread():
    """What the FORK runs for ``sageattn_qk_int8_pv_fp16_triton(quantization_backend="triton", attn_mask=...)``
    (core.py:295-318): per_thread_int8 + attn_qk_int8_per_thread.forward(..., attn_mask=bool | additive), non-causal.
    Three cases: head_dim 64 fp16 (M != N, ragged), head_dim 128 fp16 with GQA, head_dim 64 bf16 in NHD layout (V is
    converted to fp16 first, core.py:289-290); every bool mask has whole 128x64 tiles off (the kernel's tile skip,
    attn_qk_int8_per_thread.py:40-48) and one fully masked row."""
    cases = [("pt_masked_d64", 1, 2, 2, 200, 300, 64, "HND", torch.float16, 778),
             ("pt_masked_d128_gqa", 1, 4, 2, 160, 256, 128, "HND", torch.float16, 779),
             ("pt_masked_bf16_nhd", 2, 2, 2, 130, 192, 64, "NHD", torch.bfloat16, 780)]
    for name, B, Hq, Hk, M, N, D, layout, dtype, seed in cases:
        torch.manual_seed(seed)
        shp = (lambda h, n: (B, h, n, D)) if layout == "HND" else (lambda h, n: (B, n, h, D))
        q = torch.randn(shp(Hq, M)).to(dtype)
        k = (torch.randn(shp(Hk, N)) + 1.5 * torch.randn((1, Hk, 1, D) if layout == "HND" else (1, 1, Hk, D))).to(dtype)
        v = torch.randn(shp(Hk, N)).to(dtype)
        km = k.mean(dim=2 if layout == "HND" else 1, keepdim=True)   # core.py:280
        sm = 1.0 / (D ** 0.5)
        v16 = v.to(torch.float16)                                    # core.py:289-290
        q8, qs, k8, ks = per_thread_int8(q, k, km=km, sm_scale=sm, tensor_layout=layout)
        mb = torch.rand(B, 1, M, N) > 0.3
        mb[..., :128, 64:192] = False            # whole 128x64 tiles masked out (tile skipping)
        mb[..., 5, :] = False                    # a fully masked row (undefined in the reference; excluded by the tests)
        mb = mb.expand(B, Hq, M, N)
        mf = (torch.randn(B, 1, M, N) * 2).to(dtype).expand(B, Hq, M, N)
        ob, lb = run_quiet(attn_thread, q8, k8, v16, qs, ks, sm, tensor_layout=layout, attn_mask=mb, output_dtype=dtype,
                           return_lse=True)
        of, lf = run_quiet(attn_thread, q8, k8, v16, qs, ks, sm, tensor_layout=layout, attn_mask=mf, output_dtype=dtype,
                           return_lse=True)
        meta = dict(B=B, Hq=Hq, Hk=Hk, M=M, N=N, D=D, layout=layout, dtype="fp16" if dtype == torch.float16 else "bf16",
                    sm_scale=sm)
        path = os.path.join(OUT, "masked", f"{name}.npz")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        np.savez_compressed(path, q=bits(q), k=bits(k), v=bits(v), km=bits(km), q8=q8.numpy(), k8=k8.numpy(),
                            qs=qs.numpy(), ks=ks.numpy(), mask_bool=mb[:, 0].contiguous().numpy(),
                            mask_float=bits(mf[:, 0].contiguous()), o_bool=bits(ob), lse2_bool=lb.numpy(),
                            o_float=bits(of), lse2_float=lf.numpy(), meta=np.array([repr(meta)]))
        print(f"{name}: wrote {os.path.getsize(path)/1024:.0f} KiB", flush=True)


def main():
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])   # optional: names of the cases to (re)generate; seeds depend on the case's index only
    if "pt_masked" in only:
        gen_masked_thread()
        return
    if not only:
        gen_varlen()
        gen_masked()
        gen_masked_thread()
    for i, (name, B, Hq, Hk, M, N, D, layout, dt, causal, kbias) in enumerate(CASES):
        if only and name not in only:
            continue
        torch.manual_seed(1000 + i)
        dtype = torch.float16 if dt == "fp16" else torch.bfloat16
        shp = (lambda h, n: (B, h, n, D)) if layout == "HND" else (lambda h, n: (B, n, h, D))
        q = torch.randn(shp(Hq, M)).to(dtype)
        k = torch.randn(shp(Hk, N))
        if kbias:
            bias_shape = (1, Hk, 1, D) if layout == "HND" else (1, 1, Hk, D)
            k = k + kbias * torch.randn(bias_shape)  # channel outliers: exercises smooth_k
        k = k.to(dtype)
        v = torch.randn(shp(Hk, N)).to(dtype)
        seq_dim = 2 if layout == "HND" else 1
        km = k.mean(dim=seq_dim, keepdim=True)  # core.py:280
        sm_scale = 1.0 / (D ** 0.5)
        v16 = v.to(torch.float16)  # core.py:289-290

        out = {"q": bits(q), "k": bits(k), "v": bits(v), "km": bits(km)}
        # --- per-block pairing
        qb, qsb, kb, ksb = per_block_int8(q, k, km=km, sm_scale=sm_scale, tensor_layout=layout)
        fwd = attn_block_causal if causal else attn_block
        ob, lb = fwd(qb, kb, v16, qsb, ksb, tensor_layout=layout, output_dtype=dtype, return_lse=True)
        out.update(pb_q8=qb.numpy(), pb_qs=qsb.numpy(), pb_k8=kb.numpy(), pb_ks=ksb.numpy(),
                   pb_o=bits(ob), pb_lse2=lb.numpy())
        # --- per-thread pairing (non-causal only: SURVEY 3.3)
        qt, qst, kt, kst = per_thread_int8(q, k, km=km, sm_scale=sm_scale, tensor_layout=layout)
        out.update(pt_q8=qt.numpy(), pt_qs=qst.numpy(), pt_k8=kt.numpy(), pt_ks=kst.numpy())
        if not causal:
            ot, lt = run_quiet(attn_thread, qt, kt, v16, qst, kst, sm_scale, tensor_layout=layout,
                               output_dtype=dtype, return_lse=True)
            out.update(pt_o=bits(ot), pt_lse2=lt.numpy())
        meta = dict(B=B, Hq=Hq, Hk=Hk, M=M, N=N, D=D, layout=layout, dtype=dt, causal=int(causal),
                    sm_scale=sm_scale)
        out["meta"] = np.array([repr(meta)])
        path = os.path.join(OUT, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(f"{name}: wrote {os.path.getsize(path)/1024:.0f} KiB", flush=True)


if __name__ == "__main__":
    main()
