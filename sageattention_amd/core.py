"""Host-side mirror of ``sageattention/core.py`` (reference) for MI355X / gfx950.

Same public names, arguments, defaults, return values and error behaviour as the reference entry points
(``sageattn`` core.py:80, ``sageattn_qk_int8_pv_fp16_triton`` :161, ``sageattn_qk_int8_pv_fp16_cuda`` :480,
``sageattn_qk_int8_pv_fp8_cuda`` :656, ``sageattn_qk_int8_pv_fp8_cuda_sm90`` :908).  Everything below the
argument handling runs as hand-written HIP through the C ABI (include/sageattn_hip.h): K mean, INT8 quantizers
(with the LSE correction fused), V fp16/fp8 preparation and the fused attention kernel.  No Triton, no
rocWMMA, no debug dumps (the fork's torch.save side effects, core.py:320-352,845-881, are not reproduced)."""
import ctypes
import warnings
from typing import Any, Optional

import os

import torch

from . import _lib as L
from . import _qattn
from .quant import _quant, k_mean, k_smooth_quant, kv_prepare_fp8, per_channel_fp8, sub_mean

__all__ = ["sageattn", "sageattn_qk_int8_pv_fp16_cuda", "sageattn_qk_int8_pv_fp16_triton",
           "sageattn_qk_int8_pv_fp8_cuda", "sageattn_qk_int8_pv_fp8_cuda_sm90", "sageattn_varlen"]


def _common_checks(q, k, v):
    dtype = q.dtype
    assert q.is_cuda, "Input tensors must be on cuda."
    assert dtype in [torch.float16, torch.bfloat16], "Input tensors must be in dtype of torch.float16 or torch.bfloat16"
    assert q.device == k.device == v.device, "All tensors must be on the same device."
    assert q.dtype == k.dtype == v.dtype, "All tensors must have the same dtype."
    return dtype


def _pad_head_dim(q, k, v):
    """core.py:590-601"""
    head_dim_og = q.size(-1)
    if head_dim_og < 64:
        pad = 64 - head_dim_og
    elif 64 < head_dim_og < 128:
        pad = 128 - head_dim_og
    elif head_dim_og > 128:
        raise ValueError(f"Unsupported head_dim: {head_dim_og}")
    else:
        pad = 0
    if pad:
        q = torch.nn.functional.pad(q, (0, pad))
        k = torch.nn.functional.pad(k, (0, pad))
        v = torch.nn.functional.pad(v, (0, pad))
    assert q.stride(-1) == 1 and k.stride(-1) == 1 and v.stride(-1) == 1, "Last dim of qkv must be contiguous."
    return q, k, v, head_dim_og


def _quant_qk(q, k, km, tensor_layout, qk_quant_gran, sm_scale, WARPQ, want_lse_corr, Hq, Hk):
    """Quantize Q and K with a given mean (core.py:621-624): the pairing tables below, so tests and tools that need
    pre-quantized operands use exactly what the operators use."""
    gran, rnd = _k_pairing(qk_quant_gran)
    k8, ks, _ = _quant(k, tensor_layout, gran, True, 64, 64, 1.0, rnd, mean=km, dense_heads=True)
    q8, qs, corr = _quant_q(q, km, tensor_layout, qk_quant_gran, sm_scale, WARPQ, want_lse_corr, Hq, Hk)
    return q8, qs, k8, ks, corr


def _finish_lse(lse2, corr, sm_scale):
    """core.py:651: lse / 1.44269504 + lse_correction * sm_scale"""
    out = torch.empty_like(lse2)
    L.check(L.lib().sage_finish_lse(lse2.data_ptr(), L.ptr(corr), float(sm_scale), out.data_ptr(), lse2.numel(),
                                    L.stream_ptr(lse2.device)), "sage_finish_lse")
    return out


_GRAN_CODE = {"per_block": L.GRAN_PER_BLOCK, "per_block_cuda": L.GRAN_PER_BLOCK, "per_warp": L.GRAN_PER_WARP,
              "per_thread": L.GRAN_PER_THREAD}

# Fold the Q quantizer into the attention kernel (sage_attn_fusedq_*): same bits, one launch and one pass over Q less.
# Re-measured end to end in round 3 (tools/fuseq_bench.py, profiles/r03_ab/fuseq_crossover.log), after the prologue learnt
# to issue its tile copies before the Q loads: fused <= separate at every length -- 0.96-0.97 at 2K keys, 0.98-1.00 at 4K,
# 0.993 (fp16 PV) / 0.994 (fp8 PV) at C3, 0.995 / 0.992 at C4; cross-attention with 16K-32K query rows on 256-4096 keys
# (tools/cross_bench.py): 1.00-1.11x faster fused -- so there is no length limit any more (round 2 stopped at 4096 rows).  Default ON since round 2: the rare wrong 32-row wave seen under perturbed
# timing in round 1 was an LDS race in the attention kernel's prologue, fixed by one barrier
# (profiles/r02_race_evidence.md).  SAGEATTN_FUSE_Q=0 selects the stand-alone Q quantizer + kernel path (bit-identical).
FUSE_Q_QUANT = os.environ.get("SAGEATTN_FUSE_Q", "1") == "1"
FUSE_Q_MAX_SEQ = 1 << 30


# quantizer pairings of core.py:295-299,621-624: granularity name -> (K granularity, Q granularity, rounding).
# "per_block" = the Triton quantizer (quant_per_block.py), "per_block_cuda" = csrc/fused (quant.py:23-104, RNE), both with
# sm_scale*log2e folded into Q.
_PAIRING = {"per_block": (L.GRAN_PER_BLOCK, L.GRAN_PER_BLOCK, L.ROUND_TRITON),
            "per_block_cuda": (L.GRAN_PER_BLOCK, L.GRAN_PER_BLOCK, L.ROUND_CUDA),
            "per_warp": (L.GRAN_PER_BLOCK, L.GRAN_PER_WARP, L.ROUND_CUDA),
            "per_thread": (L.GRAN_PER_THREAD, L.GRAN_PER_THREAD, L.ROUND_TRITON)}


def _pairing(qk_quant_gran):
    try:
        return _PAIRING[qk_quant_gran]
    except KeyError:
        raise ValueError(f"Unsupported qk_quant_gran: {qk_quant_gran}") from None


def _k_pairing(qk_quant_gran):
    kg, _, rnd = _pairing(qk_quant_gran)
    return kg, rnd


def _prep_k(k, tensor_layout, qk_quant_gran, smooth_k):
    """``km = k.mean(seq)`` (core.py:612) + the K half of the quantizer pairings of core.py:621-624 -> (k8, ks, km).
    Two passes over K in three launches on purpose: single-pass forms (workgroups of a head exchanging partial sums
    inside one launch; one workgroup per head re-reading out of L2) were built and measured SLOWER on MI355X for every
    shape from 2K keys up (pre-pass) -- K's second read is an L2 / Infinity-Cache hit anyway."""
    gran, rnd = _k_pairing(qk_quant_gran)
    if smooth_k:
        return k_smooth_quant(k, tensor_layout, gran, rnd)
    km = None
    k8, ks, _ = _quant(k, tensor_layout, gran, True, 64, 64, 1.0, rnd, mean=km, dense_heads=True)
    return k8, ks, km


def _quant_q(q, km, tensor_layout, qk_quant_gran, sm_scale, WARPQ, want_lse_corr, Hq, Hk):
    """Q half of core.py:621-624 (+ the LSE correction q.km of core.py:613-617 in the same pass) -> (q8, qs, corr)."""
    _, gran, rnd = _pairing(qk_quant_gran)
    dot_vec = km if (want_lse_corr and km is not None) else None
    if qk_quant_gran in ("per_block", "per_block_cuda"):  # triton path: sm_scale*log2e folded into Q (quant_per_block.py:84)
        return _quant(q, tensor_layout, gran, False, 128, 128, sm_scale * 1.44269504, rnd, dot_vec=dot_vec,
                      dot_group=Hq // Hk, dense_heads=True)
    return _quant(q, tensor_layout, gran, False, 128, WARPQ, 1.0, rnd, dot_vec=dot_vec, dot_group=Hq // Hk, dense_heads=True)


def _fused_attn(q, k8, ks, v, o, km, v_scale, v_mean, tensor_layout, is_causal, qk_quant_gran, warpq, sm_scale, return_lse,
                pv_fp8):
    B, Hq, M, D = L.dims(q, tensor_layout)
    _, Hk, N, _ = L.dims(k8, tensor_layout)
    if Hq % Hk != 0:
        raise ValueError(f"num_qo_heads ({Hq}) must be divisible by num_kv_heads ({Hk})")
    lse = torch.empty((B, Hq, M), dtype=torch.float32, device=q.device) if return_lse else None
    lib, st = L.lib(), L.stream_ptr(q.device)
    vm = v_mean.to(torch.float32).contiguous() if v_mean is not None else None
    if pv_fp8:
        vd = (L.SageTensor(v.data_ptr(), v.stride(0), v.stride(1), v.stride(2)) if tensor_layout == "HND"
              else L.SageTensor(v.data_ptr(), v.stride(0), v.stride(2), v.stride(1)))
        status = lib.sage_attn_fusedq_pv_f8(L.desc(q, tensor_layout), L.dtype_code(q.dtype), L.desc(k8, tensor_layout), vd,
                                            L.desc(o, tensor_layout), L.dtype_code(o.dtype), ks.data_ptr(), L.ptr(km),
                                            v_scale.data_ptr(), L.ptr(vm), L.ptr(lse), B, Hq, Hk, M, N, D, int(is_causal),
                                            _GRAN_CODE[qk_quant_gran], warpq, float(sm_scale), st)
        L.check(status, "sage_attn_fusedq_pv_f8")
    else:
        status = lib.sage_attn_fusedq_pv_f16(L.desc(q, tensor_layout), L.dtype_code(q.dtype), L.desc(k8, tensor_layout),
                                             L.desc(v, tensor_layout), L.dtype_code(v.dtype), L.desc(o, tensor_layout),
                                             L.dtype_code(o.dtype), ks.data_ptr(), L.ptr(km), L.ptr(vm), L.ptr(lse),
                                             B, Hq, Hk, M, N, D, int(is_causal), _GRAN_CODE[qk_quant_gran], warpq,
                                             float(sm_scale), st)
        L.check(status, "sage_attn_fusedq_pv_f16")
    return lse


# One ctypes crossing and two allocations (output + workspace) per call: sage_sageattn_pv_{f16,f8} (csrc/sage_op.hip)
# sequences the same entry points the multi-call path below uses, bit-identical.  SAGEATTN_ONE_CALL=0 selects that path.
ONE_CALL = os.environ.get("SAGEATTN_ONE_CALL", "1") == "1"


def _one_call(q, k, v, tensor_layout, is_causal, qk_quant_gran, warpq, sm_scale, return_lse, pv_fp8, scale_max=448.0):
    B, Hq, M, D = L.dims(q, tensor_layout)
    _, Hk, N, _ = L.dims(k, tensor_layout)
    if Hq % Hk != 0:
        raise ValueError(f"num_qo_heads ({Hq}) must be divisible by num_kv_heads ({Hk})")
    lib = L.lib()
    opts = L.OpOpts(_GRAN_CODE[qk_quant_gran], warpq, 1, 1 if (FUSE_Q_QUANT and M <= FUSE_Q_MAX_SEQ) else 0, 0)
    nbytes = lib.sage_sageattn_workspace_bytes(int(pv_fp8), B, Hq, Hk, M, N, D, int(return_lse), opts)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
    o = torch.empty(q.size(), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, Hq, M), dtype=torch.float32, device=q.device) if return_lse else None
    code, st = L.dtype_code(q.dtype), L.stream_ptr(q.device)
    qd, kd, vd, od = (L.desc(t, tensor_layout) for t in (q, k, v, o))
    if pv_fp8:
        L.check(lib.sage_sageattn_pv_f8(qd, kd, vd, code, od, L.ptr(lse), B, Hq, Hk, M, N, D, int(is_causal), float(sm_scale),
                                        float(scale_max), opts, ws.data_ptr(), nbytes, st), "sage_sageattn_pv_f8")
    else:
        L.check(lib.sage_sageattn_pv_f16(qd, kd, vd, code, od, L.ptr(lse), B, Hq, Hk, M, N, D, int(is_causal), float(sm_scale),
                                         opts, ws.data_ptr(), nbytes, st), "sage_sageattn_pv_f16")
    return o, lse


def _sage_fp16(q, k, v, tensor_layout, is_causal, qk_quant_gran, sm_scale, smooth_k, smooth_v, return_lse, WARPQ=32):
    dtype = q.dtype
    with torch.cuda.device(q.device):  # the reference's torch.cuda.set_device(v.device) workaround, core.py:583
        q, k, v, head_dim_og = _pad_head_dim(q, k, v)
        if sm_scale is None:
            sm_scale = head_dim_og ** -0.5
        _, Hq, _, _ = L.dims(q, tensor_layout)
        _, Hk, _, _ = L.dims(k, tensor_layout)
        if ONE_CALL and smooth_k and not smooth_v and qk_quant_gran in ("per_warp", "per_thread"):
            o, lse = _one_call(q, k, v, tensor_layout, is_causal, qk_quant_gran, WARPQ, sm_scale, return_lse, False)
            o = o[..., :head_dim_og]
            return (o, lse) if return_lse else o
        k8, ks, km = _prep_k(k, tensor_layout, qk_quant_gran, smooth_k)
        o = torch.empty(q.size(), dtype=dtype, device=q.device)
        vm = None
        if smooth_v:
            v, vm = sub_mean(v, tensor_layout)
        if FUSE_Q_QUANT and not qk_quant_gran.startswith("per_block") and L.dims(q, tensor_layout)[2] <= FUSE_Q_MAX_SEQ:
            lse = _fused_attn(q, k8, ks, v, o, km, None, vm, tensor_layout, is_causal, qk_quant_gran, WARPQ, sm_scale,
                              return_lse, False)
            o = o[..., :head_dim_og]
            return (o, lse) if return_lse else o
        q8, qs, corr = _quant_q(q, km, tensor_layout, qk_quant_gran, sm_scale, WARPQ, return_lse, Hq, Hk)
        lse2 = _qattn._attn_f16(q8, k8, v, o, qs, ks, vm, 0 if tensor_layout == "NHD" else 1, int(is_causal),
                                _GRAN_CODE[qk_quant_gran], sm_scale, int(return_lse),
                                logit_mult_is_one=qk_quant_gran.startswith("per_block"))
        o = o[..., :head_dim_og]
        if return_lse:
            return o, _finish_lse(lse2, corr if smooth_k else None, sm_scale)
        return o


@torch.compiler.disable
def sageattn_qk_int8_pv_fp16_cuda(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    tensor_layout: str = "HND",
    is_causal: bool = False,
    qk_quant_gran: str = "per_thread",
    sm_scale: Optional[float] = None,
    pv_accum_dtype: str = "fp32",
    smooth_k: bool = True,
    smooth_v: bool = False,
    return_lse: bool = False,
    **kwargs: Any,
) -> torch.Tensor:
    """SageAttention with INT8 Q/K and FP16 PV (reference core.py:480-653).

    On gfx950 the PV MFMA (v_mfma_f32_32x32x16_f16) accumulates in fp32 for every ``pv_accum_dtype`` the
    reference names ("fp32", "fp16", "fp16+fp32"); ``smooth_v`` is honoured only for "fp16" as in the reference
    (core.py:628-630) -- it is numerically a no-op with an fp32 accumulator, but kept for API parity.
    Unknown keyword arguments (SDPA's ``attn_mask=``, ``dropout_p=`` ...) are accepted and ignored, as in the
    reference (core.py:492)."""
    _common_checks(q, k, v)
    assert qk_quant_gran in ["per_warp", "per_thread"], "qk_quant_gran must be either 'per_warp' or 'per_thread'."
    if pv_accum_dtype not in ("fp32", "fp16", "fp16+fp32"):
        raise ValueError(f"Unsupported pv_accum_dtype: {pv_accum_dtype}")
    if pv_accum_dtype in ["fp32", "fp16+fp32"] and smooth_v:
        warnings.warn(f"pv_accum_dtype is {pv_accum_dtype}, smooth_v will be ignored.")
        smooth_v = False
    warpq = 16 if (q.size(-1) > 64 and pv_accum_dtype == "fp16+fp32") else 32  # core.py:622
    return _sage_fp16(q, k, v, tensor_layout, is_causal, qk_quant_gran, sm_scale, smooth_k, smooth_v, return_lse, warpq)


@torch.compiler.disable
def sageattn_qk_int8_pv_fp16_triton(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    tensor_layout: str = "HND",
    quantization_backend: str = "triton",
    is_causal: bool = False,
    attn_mask: Optional[torch.Tensor] = None,
    sm_scale: Optional[float] = None,
    smooth_k: bool = True,
    return_lse: bool = False,
    **kwargs: Any,
) -> torch.Tensor:
    """Reference core.py:161-360 (INT8 Q/K, FP16 PV, the reference's Triton kernels), served by the HIP kernels.

    ``quantization_backend`` selects the quantizer pairing the way the fork does (core.py:295-318):
      * ``"triton"``, non-causal (with or without ``attn_mask``): per-thread quantization (quant_per_thread.py) and the
        per-thread kernel (attn_qk_int8_per_thread.py); sm_scale*log2e applied in the kernel.
      * ``"cuda"``: per-block quantization with the CUDA quantizer's rounding (quant.py:23-104, sm_scale*log2e folded
        into Q) and the per-block kernel.
      * causal calls keep the upstream per-block pairing (quant_per_block.py + attn_qk_int8_per_block_causal.py) for
        ``"triton"``: the fork hands per-thread scales to the per-block causal kernel there, which indexes them with
        per-block strides -- not reproduced.
    ``attn_mask`` (bool, or additive in q's dtype; any shape broadcastable to [B,H,M,N]) follows the reference kernels:
    False adds -1e6, an additive mask is added to the base-2 logits (attn_qk_int8_per_thread.py:37-75).  ``sm_scale`` is
    honoured (the fork's per-thread kernel hard-codes 1/sqrt(padded head_dim), attn_qk_int8_per_thread.py:66)."""
    dtype = _common_checks(q, k, v)
    if quantization_backend not in ("triton", "cuda"):
        raise ValueError(f"Unsupported quantization backend: {quantization_backend}")
    if quantization_backend == "cuda":
        gran = "per_block_cuda"
    else:
        gran = "per_block" if is_causal else "per_thread"
    if attn_mask is None:
        return _sage_fp16(q, k, v, tensor_layout, is_causal, gran, sm_scale, smooth_k, False, return_lse)
    # ---- attn_mask (core.py:249-251, 302-318)
    assert attn_mask.dtype == torch.bool or attn_mask.dtype == q.dtype, "attn_mask must be of dtype bool or the same dtype as q."
    assert attn_mask.device == q.device, "All tensors must be on the same device."
    assert not is_causal, "Mask should be None for causal attention."
    with torch.cuda.device(q.device):
        q, k, v, head_dim_og = _pad_head_dim(q, k, v)
        if sm_scale is None:
            sm_scale = 1.0 / (head_dim_og ** 0.5)
        B, Hq, M, D = L.dims(q, tensor_layout)
        _, Hk, N, _ = L.dims(k, tensor_layout)
        try:
            attn_mask = attn_mask.expand((B, Hq, M, N))
        except Exception:
            raise AssertionError(f"attn_mask shape {attn_mask.shape} cannot be broadcast to {(B, Hq, M, N)}")
        # (the reference converts a bf16 V to fp16 here, core.py:289-290; this kernel multiplies it as bf16)
        k8, ks, km = _prep_k(k, tensor_layout, gran, smooth_k)
        q8, qs, corr = _quant_q(q, km, tensor_layout, gran, sm_scale, 32, return_lse, Hq, Hk)
        o = torch.empty(q.size(), dtype=dtype, device=q.device)
        lse2 = torch.empty((B, Hq, M), dtype=torch.float32, device=q.device) if return_lse else None
        kind = 1 if attn_mask.dtype == torch.bool else (2 if attn_mask.dtype == torch.float16 else 3)
        strides = (ctypes.c_int64 * 4)(*attn_mask.stride())
        per_block = gran != "per_thread"
        L.check(L.lib().sage_attn_qk_int8_pv_f16_masked(
            L.desc(q8, tensor_layout), L.desc(k8, tensor_layout), L.desc(v, tensor_layout), L.dtype_code(v.dtype),
            L.desc(o, tensor_layout), L.dtype_code(dtype), qs.data_ptr(), ks.data_ptr(), attn_mask.data_ptr(), kind, strides,
            L.ptr(lse2), B, Hq, Hk, M, N, D, _GRAN_CODE[gran], 128, 128 if per_block else 32, float(sm_scale),
            1 if per_block else 0, L.stream_ptr(q.device)), "sage_attn_qk_int8_pv_f16_masked")
        o = o[..., :head_dim_og]
        if return_lse:
            return o, _finish_lse(lse2, corr if smooth_k else None, sm_scale)
        return o


@torch.compiler.disable
def sageattn_qk_int8_pv_fp8_cuda(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    tensor_layout: str = "HND",
    is_causal: bool = False,
    qk_quant_gran: str = "per_thread",
    sm_scale: Optional[float] = None,
    pv_accum_dtype: str = "fp32+fp16",
    smooth_k: bool = True,
    smooth_v: bool = False,
    return_lse: bool = False,
    **kwargs: Any,
) -> torch.Tensor:
    """SageAttention with INT8 Q/K and FP8 (OCP e4m3fn) PV, fp32 accumulation (reference core.py:656-905).
    All of "fp32", "fp32+fp32", "fp32+fp16" accumulate in true fp32 on MFMA (no fp22 accumulator issue, so no
    two-level buffer); V is quantized per channel with scale_max 448 (quant.py:228)."""
    dtype = _common_checks(q, k, v)
    assert qk_quant_gran in ["per_warp", "per_thread"], "qk_quant_gran must be either 'per_warp' or 'per_thread'."
    if pv_accum_dtype not in ("fp32", "fp32+fp32", "fp32+fp16"):
        raise ValueError(f"Unsupported pv_accum_dtype: {pv_accum_dtype}")
    if pv_accum_dtype in ("fp32+fp32", "fp32+fp16") and smooth_v:
        warnings.warn(f"pv_accum_dtype is '{pv_accum_dtype}', smooth_v will be ignored.")
        smooth_v = False
    with torch.cuda.device(q.device):
        q, k, v, head_dim_og = _pad_head_dim(q, k, v)
        if sm_scale is None:
            sm_scale = head_dim_og ** -0.5
        _, Hq, _, _ = L.dims(q, tensor_layout)
        _, Hk, _, _ = L.dims(k, tensor_layout)
        if ONE_CALL and smooth_k and not smooth_v and k.shape == v.shape:
            o, lse = _one_call(q, k, v, tensor_layout, is_causal, qk_quant_gran, 32, sm_scale, return_lse, True)
            o = o[..., :head_dim_og]
            return (o, lse) if return_lse else o
        o = torch.empty(q.size(), dtype=dtype, device=q.device)
        if smooth_k and not smooth_v and k.shape == v.shape and k.dtype == v.dtype:
            # the default configuration: K and V prepared by one call (two launches at every length)
            gran, rnd = _k_pairing(qk_quant_gran)
            k8, ks, km, v8, v_scale = kv_prepare_fp8(k, v, tensor_layout, gran, rnd, scale_max=448.0)
            vm = None
        else:
            k8, ks, km = _prep_k(k, tensor_layout, qk_quant_gran, smooth_k)
            v8, v_scale, vm = per_channel_fp8(v, tensor_layout=tensor_layout, scale_max=448.0, smooth_v=smooth_v)
        if FUSE_Q_QUANT and L.dims(q, tensor_layout)[2] <= FUSE_Q_MAX_SEQ:
            lse = _fused_attn(q, k8, ks, v8, o, km, v_scale, vm, tensor_layout, is_causal, qk_quant_gran, 32, sm_scale,
                              return_lse, True)
            o = o[..., :head_dim_og]
            return (o, lse) if return_lse else o
        q8, qs, corr = _quant_q(q, km, tensor_layout, qk_quant_gran, sm_scale, 32, return_lse, Hq, Hk)
        lse2 = _qattn._attn_f8(q8, k8, v8, o, qs, ks, v_scale, vm, 0 if tensor_layout == "NHD" else 1, int(is_causal),
                               _GRAN_CODE[qk_quant_gran], sm_scale, int(return_lse))
        o = o[..., :head_dim_og]
        if return_lse:
            return o, _finish_lse(lse2, corr if smooth_k else None, sm_scale)
        return o


@torch.compiler.disable
def sageattn_qk_int8_pv_fp8_cuda_sm90(q, k, v, tensor_layout="HND", is_causal=False, qk_quant_gran="per_thread",
                                      sm_scale=None, pv_accum_dtype="fp32+fp32", smooth_k=True, return_lse=False,
                                      **kwargs):
    """Reference core.py:908-1065 (Hopper TMA/wgmma variant).  Same operator; served by the gfx950 FP8 kernel."""
    return sageattn_qk_int8_pv_fp8_cuda(q, k, v, tensor_layout=tensor_layout, is_causal=is_causal,
                                        qk_quant_gran=qk_quant_gran, sm_scale=sm_scale, pv_accum_dtype=pv_accum_dtype,
                                        smooth_k=smooth_k, smooth_v=False, return_lse=return_lse)


def dispatch_pv(q: torch.Tensor, k: torch.Tensor, tensor_layout: str = "HND", is_causal: bool = False,
                n_kv: Optional[int] = None) -> str:
    """"fp8" or "fp16": the P.V precision ``sageattn`` uses for these shapes (see its docstring).  ``n_kv`` overrides
    the key count (sequence-parallel callers pass the length of the WHOLE sequence)."""
    choice = os.environ.get("SAGEATTN_DISPATCH", "auto")
    if choice not in ("auto", "fp8", "fp16"):
        raise ValueError(f"SAGEATTN_DISPATCH must be auto, fp8 or fp16, got {choice}")
    if choice != "auto":
        return choice
    if tensor_layout not in ("HND", "NHD"):
        raise ValueError(f"Unknown tensor layout: {tensor_layout}")
    if n_kv is None:
        n_kv = k.size(2) if tensor_layout == "HND" else k.size(1)
    keys_per_row = n_kv // 2 if is_causal else n_kv
    return "fp8" if keys_per_row >= (4096 if q.size(-1) <= 64 else 2048) else "fp16"


def sageattn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, tensor_layout: str = "HND", is_causal: bool = False,
             sm_scale: Optional[float] = None, return_lse: bool = False, **kwargs: Any):
    """Drop-in for ``F.scaled_dot_product_attention`` (reference core.py:80-144: "automatically selects the optimal
    kernel").  The fork dispatches to the INT8-QK / FP8-PV path with fp32 accumulation (core.py:144), and upstream picks
    its FP8 kernels on every FP8-capable architecture (core.py:151-156): on gfx950 the MX-scaled FP8 MFMA makes that the
    fastest variant from a few thousand keys upwards.  Below that the per-channel V quantizer (two more launches and
    two passes over V) costs more than the FP8 MFMA saves, so short sequences take the FP16-PV operator, which is also
    the more accurate of the two; the crossover was measured end to end (profiles/r01d_sweep_end_to_end.md):
    keys per query row >= 4096 at head_dim <= 64, >= 2048 above (a causal row sees half the keys on average).
    ``SAGEATTN_DISPATCH=fp8|fp16`` pins the choice; the two operators are also exported by name."""
    choice = dispatch_pv(q, k, tensor_layout, is_causal)
    if choice == "fp16":
        return sageattn_qk_int8_pv_fp16_cuda(q, k, v, tensor_layout=tensor_layout, is_causal=is_causal,
                                             sm_scale=sm_scale, return_lse=return_lse, pv_accum_dtype="fp32")
    return sageattn_qk_int8_pv_fp8_cuda(q, k, v, tensor_layout=tensor_layout, is_causal=is_causal, sm_scale=sm_scale,
                                        return_lse=return_lse, pv_accum_dtype="fp32")


@torch.compiler.disable
def sageattn_varlen(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    cu_seqlens_q: torch.Tensor,
    cu_seqlens_k: torch.Tensor,
    max_seqlen_q: int,
    max_seqlen_k: int,
    is_causal: bool = False,
    sm_scale: Optional[float] = None,
    smooth_k: bool = True,
    **kwargs: Any,
) -> torch.Tensor:
    """Packed variable-length SageAttention (reference core.py:363-477): q ``[cu_seqlens_q[-1], Hq, D]``, k/v
    ``[cu_seqlens_k[-1], Hk, D]``; per-block INT8 Q/K (blocks restart at each sequence), FP16 PV; the K smoothing
    mean is taken over ALL packed tokens (core.py:461).  The cumulative lengths stay on the device: no host sync."""
    dtype = _common_checks(q, k, v)
    assert cu_seqlens_q.is_contiguous() and cu_seqlens_k.is_contiguous(), "cu_seqlens_q and cu_seqlens_k must be contiguous."
    assert q.dim() == 3 and k.dim() == 3 and v.dim() == 3, "q, k, v must be [total_tokens, heads, head_dim]"
    with torch.cuda.device(q.device):
        q, k, v, head_dim_og = _pad_head_dim(q, k, v)
        if sm_scale is None:
            sm_scale = 1.0 / (head_dim_og ** 0.5)
        Tq, Hq, D = q.shape
        Tk, Hk, _ = k.shape
        nseq = cu_seqlens_q.numel() - 1
        cu_q = cu_seqlens_q.to(device=q.device, dtype=torch.int32)
        cu_k = cu_seqlens_k.to(device=q.device, dtype=torch.int32)
        lib, st = L.lib(), L.stream_ptr(q.device)
        code = L.dtype_code(dtype)
        # packed [T,H,D] == NHD with batch 1
        km = k_mean(k.unsqueeze(0), "NHD") if smooth_k else None  # [1,Hk,D]

        def desc3(t):
            return L.SageTensor(t.data_ptr(), 0, t.stride(1), t.stride(0))
        q8 = torch.empty(q.shape, dtype=torch.int8, device=q.device)
        k8 = torch.empty(k.shape, dtype=torch.int8, device=q.device)
        qs = torch.empty((nseq, Hq, (max_seqlen_q + 127) // 128), dtype=torch.float32, device=q.device)
        ks = torch.empty((nseq, Hk, (max_seqlen_k + 63) // 64), dtype=torch.float32, device=q.device)
        L.check(lib.sage_quant_qk_int8_varlen(desc3(q), code, cu_q.data_ptr(), nseq, Hq, int(max_seqlen_q), D, None,
                                              desc3(q8), qs.data_ptr(), L.GRAN_PER_BLOCK, 0, 128, 128,
                                              float(sm_scale * 1.44269504), L.ROUND_TRITON, st), "sage_quant_qk_int8_varlen")
        L.check(lib.sage_quant_qk_int8_varlen(desc3(k), code, cu_k.data_ptr(), nseq, Hk, int(max_seqlen_k), D, L.ptr(km),
                                              desc3(k8), ks.data_ptr(), L.GRAN_PER_BLOCK, 1, 64, 64, 1.0, L.ROUND_TRITON, st),
                "sage_quant_qk_int8_varlen")
        o = torch.empty(q.shape, dtype=dtype, device=q.device)
        L.check(lib.sage_attn_qk_int8_pv_f16_varlen(desc3(q8), desc3(k8), desc3(v), code, desc3(o), code, qs.data_ptr(),
                                                    ks.data_ptr(), cu_q.data_ptr(), cu_k.data_ptr(), nseq, Hq, Hk,
                                                    int(max_seqlen_q), int(max_seqlen_k), D, int(is_causal), L.GRAN_PER_BLOCK,
                                                    128, 128, float(sm_scale), 1, st), "sage_attn_qk_int8_pv_f16_varlen")
        return o[..., :head_dim_og]
