"""attn_mask of sageattn_qk_int8_pv_fp16_triton (reference core.py:249-251,302-318).  Golden vectors:
tests/golden/masked/masked_d64.npz from the reference's per-block Triton kernel with a bool and an additive mask."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, LSE2_TOL_FP32_P, LSE2_TOL_ROUNDED_P, calc_diff


def _load():
    z = np.load(os.path.join(GOLDEN_DIR, "masked", "masked_d64.npz"), allow_pickle=False)
    f16 = lambda a: torch.from_numpy(a.view(np.int16).copy()).view(torch.float16)
    g = {k: f16(z[k]) for k in ("q", "k", "v", "km", "mask_float", "o_bool", "o_float")}
    g.update({k: torch.from_numpy(z[k].copy()) for k in ("q8", "k8", "qs", "ks", "mask_bool", "lse2_bool", "lse2_float")})
    return g


def _rows_with_keys(mask_bool):
    return mask_bool.any(dim=-1)  # [M]: rows that keep at least one key (others are undefined in the reference)


def test_masked_oracle_vs_reference():
    from oracle import sage_oracle as O
    g = _load()
    M, N = g["q8"].shape[2], g["k8"].shape[2]
    qrows, kcols = O.expand_q_scale(g["qs"], M, "per_block"), O.expand_k_scale(g["ks"], N, "per_block")
    ok = _rows_with_keys(g["mask_bool"])
    for key, mask in (("bool", g["mask_bool"].view(1, 1, M, N).expand(1, 2, M, N)),
                      ("float", g["mask_float"].view(1, 1, M, N).expand(1, 2, M, N))):
        o, lse2 = O.attn_tile_loop(g["q8"], g["k8"], g["v"], qrows, kcols, logit_mult=1.0, flavor="triton", attn_mask=mask)
        ref, rl = g[f"o_{key}"].float(), g[f"lse2_{key}"]
        sel = ok if key == "bool" else torch.ones_like(ok)
        assert ((o.float() - ref).abs() <= 2 * 2.0 ** -10 * ref.abs().clamp(min=0.25))[:, :, sel].all()
        assert (lse2 - rl)[:, :, sel].abs().max() < 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["bool", "float"])
def test_masked_hip_vs_reference(kind):
    """The HIP kernel fed the int8 tensors and scales of the fixture, against the reference output (|do| <= 4e-3, as
    for the unmasked kernel) and its base-2 LSE; fully masked rows excluded (undefined in the reference)."""
    from sageattention_amd import _lib as L
    import ctypes
    g = _load()
    M, N, D = g["q8"].shape[2], g["k8"].shape[2], 64
    mask = (g["mask_bool"] if kind == "bool" else g["mask_float"][0]).cuda().view(1, 1, M, N).expand(1, 2, M, N)
    o = torch.empty(1, 2, M, D, dtype=torch.float16, device="cuda")
    lse = torch.empty(1, 2, M, dtype=torch.float32, device="cuda")
    q8, k8, v = g["q8"].cuda(), g["k8"].cuda(), g["v"].cuda()
    qs, ks = g["qs"].cuda(), g["ks"].cuda()
    st = (ctypes.c_int64 * 4)(*mask.stride())
    for nw in (8, 4):
        L.lib().sage_set_tuning(0, nw)
        L.check(L.lib().sage_attn_qk_int8_pv_f16_masked(
            L.desc(q8, "HND"), L.desc(k8, "HND"), L.desc(v, "HND"), 0, L.desc(o, "HND"), 0, qs.data_ptr(), ks.data_ptr(),
            mask.data_ptr(), 1 if kind == "bool" else 2, st, lse.data_ptr(), 1, 2, 2, M, N, D, 1, 128, 128, D ** -0.5, 1,
            torch.cuda.current_stream().cuda_stream), "masked")
        torch.cuda.synchronize()
        L.lib().sage_set_tuning(0, 0)
        sel = _rows_with_keys(g["mask_bool"]) if kind == "bool" else torch.ones(M, dtype=torch.bool)
        ref, rl = g[f"o_{kind}"].float(), g[f"lse2_{kind}"]
        assert (o.cpu().float() - ref)[:, :, sel].abs().max() < 4e-3
        assert calc_diff(o.cpu().float()[:, :, sel], ref[:, :, sel]) < 1e-5
        # head_dim 64: l is summed from the fp16-rounded P (MFMA row sums, as the reference's CUDA kernel), the
        # reference Triton kernel sums the fp32 p: one rounding instance, bound derived in conftest
        assert (lse.cpu() - rl)[:, :, sel].abs().max() < LSE2_TOL_ROUNDED_P
        # the same call with V given as bf16 (multiplied as bf16, P rounded to bf16): the fixture's fp16 V loses 3 bits on
        # the way, P another 3 -- within 3e-2 of the reference output; the LSE does not depend on V
        o_b = torch.empty_like(o)
        lse_b = torch.empty_like(lse)
        vb = v.bfloat16()
        L.lib().sage_set_tuning(0, nw)
        L.check(L.lib().sage_attn_qk_int8_pv_f16_masked(
            L.desc(q8, "HND"), L.desc(k8, "HND"), L.desc(vb, "HND"), 1, L.desc(o_b, "HND"), 0, qs.data_ptr(), ks.data_ptr(),
            mask.data_ptr(), 1 if kind == "bool" else 2, st, lse_b.data_ptr(), 1, 2, 2, M, N, D, 1, 128, 128, D ** -0.5, 1,
            torch.cuda.current_stream().cuda_stream), "masked bf16")
        torch.cuda.synchronize()
        L.lib().sage_set_tuning(0, 0)
        assert (o_b.cpu().float() - ref)[:, :, sel].abs().max() < 3e-2
        assert (lse_b.cpu() - rl)[:, :, sel].abs().max() < LSE2_TOL_FP32_P   # bf16 P: fp32 sums of the unrounded p


@pytest.mark.gpu
def test_masked_api_end_to_end():
    """Public entry point with broadcastable masks (bool [M,N], additive [B,1,1,N] padding mask), NHD layout, bf16, GQA."""
    import sageattention_amd as sa
    from oracle import sage_oracle as O
    torch.manual_seed(11)
    B, Hq, Hk, M, N, D = 2, 4, 2, 150, 260, 128
    q = torch.randn(B, M, Hq, D).bfloat16(); k = torch.randn(B, N, Hk, D).bfloat16(); v = torch.randn(B, N, Hk, D).bfloat16()
    mb = torch.rand(M, N) > 0.4
    mb[:, 0] = True
    pad = torch.zeros(B, 1, 1, N); pad[0, ..., 200:] = -1e4; pad = pad.bfloat16()
    for mask in (mb, pad):
        o = sa.sageattn_qk_int8_pv_fp16_triton(q.cuda(), k.cuda(), v.cuda(), tensor_layout="NHD", attn_mask=mask.cuda())
        torch.cuda.synchronize()
        full = mask.expand(B, Hq, M, N)
        s = torch.einsum("bmhd,bnhd->bhmn", q.float(), k.float().repeat_interleave(2, dim=2)) * D ** -0.5
        s = s + (torch.where(full, 0.0, float("-inf")) if full.dtype == torch.bool else full.float() * 0.6931471805599453)
        ref = torch.einsum("bhmn,bnhd->bmhd", torch.softmax(s, -1), v.float().repeat_interleave(2, dim=2))
        assert o.shape == q.shape and o.dtype == torch.bfloat16
        assert (o.cpu().float() - ref).abs().max() < 0.08
    with pytest.raises(AssertionError):
        sa.sageattn_qk_int8_pv_fp16_triton(q.cuda(), k.cuda(), v.cuda(), tensor_layout="NHD", attn_mask=mb.cuda(), is_causal=True)
    with pytest.raises(AssertionError):
        sa.sageattn_qk_int8_pv_fp16_triton(q.cuda(), k.cuda(), v.cuda(), tensor_layout="NHD", attn_mask=torch.ones(3, 7, dtype=torch.bool).cuda())


# ---- the pairing the FORK runs for quantization_backend="triton" (core.py:295-318): per-thread INT8 scales +
#      attn_qk_int8_per_thread.forward(..., attn_mask=...).  Fixtures: oracle/gen_golden.py:gen_masked_thread.
PT_CASES = ["pt_masked_d64", "pt_masked_d128_gqa", "pt_masked_bf16_nhd"]


def _load_pt(name):
    z = np.load(os.path.join(GOLDEN_DIR, "masked", f"{name}.npz"), allow_pickle=False)
    meta = eval(str(z["meta"][0]), {"__builtins__": {}}, {})
    dt = torch.float16 if meta["dtype"] == "fp16" else torch.bfloat16
    el = lambda a: torch.from_numpy(a.view(np.int16).copy()).view(dt)
    g = {k: el(z[k]) for k in ("q", "k", "v", "km", "mask_float", "o_bool", "o_float")}
    g.update({k: torch.from_numpy(z[k].copy()) for k in ("q8", "k8", "qs", "ks", "mask_bool", "lse2_bool", "lse2_float")})
    return g, meta, dt


def _hnd(x, layout):
    return x if layout == "HND" else x.transpose(1, 2)


@pytest.mark.parametrize("name", PT_CASES)
def test_masked_per_thread_oracle_vs_reference(name):
    """The oracle's tile loop with per-thread scale maps and the reference's own arithmetic ("triton" flavor: V as fp16,
    per-tile fp16 PV rounding) against the reference Triton kernel's output for bool and additive masks."""
    from oracle import sage_oracle as O
    g, meta, dt = _load_pt(name)
    B, Hq, M, N, D, layout = meta["B"], meta["Hq"], meta["M"], meta["N"], meta["D"], meta["layout"]
    q8, k8, v = _hnd(g["q8"], layout), _hnd(g["k8"], layout), _hnd(g["v"], layout).to(torch.float16)
    qrows, kcols = O.expand_q_scale(g["qs"], M, "per_thread"), O.expand_k_scale(g["ks"], N, "per_thread")
    ok = _rows_with_keys(g["mask_bool"])  # [B,M]
    for key in ("bool", "float"):
        mask = g[f"mask_{key}"].view(B, 1, M, N).expand(B, Hq, M, N)
        o, lse2 = O.attn_tile_loop(q8, k8, v, qrows, kcols, logit_mult=meta["sm_scale"] * 1.4426950408889634,
                                   flavor="triton", attn_mask=mask, out_dtype=dt)
        ref, rl = _hnd(g[f"o_{key}"], layout).float(), g[f"lse2_{key}"]
        sel = (ok if key == "bool" else torch.ones_like(ok)).view(B, 1, M).expand(B, Hq, M)
        ulp = 2.0 ** (-10 if dt == torch.float16 else -7)
        assert ((o.float() - ref).abs() <= 2 * ulp * ref.abs().clamp(min=0.25))[sel].all()
        assert (lse2 - rl)[sel].abs().max() < 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["bool", "float"])
@pytest.mark.parametrize("name", PT_CASES)
def test_masked_per_thread_hip_vs_reference(name, kind):
    """KTHREAD x HAS_MASK instantiations of the attention kernel: sage_attn_qk_int8_pv_f16_masked(qk_gran = per_thread)
    fed the fixture's int8 tensors and per-thread scales, both workgroup geometries, against the reference kernel's output
    and base-2 LSE with the tolerances of the per-block masked test.  The bf16 fixture runs twice: V converted to fp16
    first (the reference's own arithmetic, core.py:289-290: same tolerances) and V multiplied as bf16 (this library's
    default for bf16 callers: P carries 8 significant bits, bound 3e-2 as in the per-block test)."""
    from sageattention_amd import _lib as L
    import ctypes
    g, meta, dt = _load_pt(name)
    B, Hq, Hk, M, N, D, layout = meta["B"], meta["Hq"], meta["Hk"], meta["M"], meta["N"], meta["D"], meta["layout"]
    el = 0 if dt == torch.float16 else 1
    mask = (g["mask_bool"] if kind == "bool" else g["mask_float"]).cuda().view(B, 1, M, N).expand(B, Hq, M, N)
    q8, k8, qs, ks = g["q8"].cuda(), g["k8"].cuda(), g["qs"].cuda(), g["ks"].cuda()
    st = (ctypes.c_int64 * 4)(*mask.stride())
    ok = _rows_with_keys(g["mask_bool"])
    sel = (ok if kind == "bool" else torch.ones_like(ok)).view(B, 1, M).expand(B, Hq, M)
    ref, rl = _hnd(g[f"o_{kind}"], layout).float(), g[f"lse2_{kind}"]
    v_forms = [(g["v"].to(torch.float16).cuda(), 0, 4e-3 if dt == torch.float16 else 1.6e-2)]  # bf16 output: one bf16 ulp at |o| ~ 2
    if dt == torch.bfloat16:
        v_forms.append((g["v"].cuda(), 1, 3e-2))
    for v, v_el, tol in v_forms:
        for nw in (8, 4):
            o = torch.empty(g["q"].shape, dtype=dt, device="cuda")
            lse = torch.empty(B, Hq, M, dtype=torch.float32, device="cuda")
            L.lib().sage_set_tuning(0, nw)
            L.check(L.lib().sage_attn_qk_int8_pv_f16_masked(
                L.desc(q8, layout), L.desc(k8, layout), L.desc(v, layout), v_el, L.desc(o, layout), el, qs.data_ptr(),
                ks.data_ptr(), mask.data_ptr(), 1 if kind == "bool" else (2 if dt == torch.float16 else 3), st,
                lse.data_ptr(), B, Hq, Hk, M, N, D, 3, 128, 32, meta["sm_scale"], 0,
                torch.cuda.current_stream().cuda_stream), "masked per_thread")
            torch.cuda.synchronize()
            L.lib().sage_set_tuning(0, 0)
            got = _hnd(o, layout).cpu().float()
            assert (got - ref)[sel].abs().max() < tol, (name, kind, nw, v_el)
            if dt == torch.float16:
                assert calc_diff(got[sel], ref[sel]) < 1e-5
            assert (lse.cpu() - rl)[sel].abs().max() < (LSE2_TOL_ROUNDED_P if (D == 64 and v_el == 0) else LSE2_TOL_FP32_P)


@pytest.mark.gpu
def test_triton_entry_point_honours_quantization_backend():
    """sageattn_qk_int8_pv_fp16_triton: "triton" -> per-thread quantization + per-thread kernel for non-causal calls, with
    or without attn_mask (the fork, core.py:295-318); "cuda" -> per-block (core.py:299); causal calls keep the upstream
    per-block pairing (the fork's causal pairing mismatches scale shapes)."""
    import sageattention_amd as sa
    from oracle import sage_oracle as O
    g, meta, dt = _load_pt("pt_masked_d64")
    q, k, v = g["q"].cuda(), g["k"].cuda(), g["v"].cuda()
    M, N = meta["M"], meta["N"]
    mb = g["mask_bool"].view(1, 1, M, N).cuda()
    ok = _rows_with_keys(g["mask_bool"]).view(1, 1, M).expand(1, 2, M)
    o_t, lse_t = sa.sageattn_qk_int8_pv_fp16_triton(q, k, v, attn_mask=mb, quantization_backend="triton", return_lse=True)
    o_c = sa.sageattn_qk_int8_pv_fp16_triton(q, k, v, attn_mask=mb, quantization_backend="cuda")
    torch.cuda.synchronize()
    # end to end against the reference's output for the same fp16 inputs (its quantizer is bit-exact here, test_gpu_parity)
    assert (o_t.cpu().float() - g["o_bool"].float())[ok].abs().max() < 4e-3
    corr = O.lse_correction(g["q"], g["km"], "HND") * meta["sm_scale"]
    assert (lse_t.cpu() - (g["lse2_bool"] / 1.44269504 + corr))[ok].abs().max() < 2e-3
    # the two backends quantize differently (per-thread vs per-block scales): close, not identical
    d = (o_t.float() - o_c.float()).cpu()[ok].abs().max().item()  # (fully masked rows are undefined)
    assert 0 < d < 3e-2, d
    # unmasked, non-causal "triton" == the per-thread operator; causal == the per-block pairing
    a = sa.sageattn_qk_int8_pv_fp16_triton(q[:, :, :128], k[:, :, :128], v[:, :, :128])
    b = sa.sageattn_qk_int8_pv_fp16_cuda(q[:, :, :128], k[:, :, :128], v[:, :, :128], qk_quant_gran="per_thread")
    assert torch.equal(a, b)
    with pytest.raises(ValueError):
        sa.sageattn_qk_int8_pv_fp16_triton(q, k, v, quantization_backend="nope")
