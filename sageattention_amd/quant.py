"""Host-side mirror of ``sageattention/quant.py`` (reference) on top of the gfx950 C-ABI library.

Same function names, argument meaning and returned shapes as the reference
(``per_block_int8`` quant.py:23, ``per_warp_int8`` :106, ``sub_mean`` :183, ``per_channel_fp8`` :225) plus
``per_thread_int8`` (sageattention/triton/quant_per_thread.py:158) and ``k_mean`` (core.py:612)."""
from typing import Optional

import torch

from . import _lib as L


def _km3(km: Optional[torch.Tensor], tensor_layout: str) -> Optional[torch.Tensor]:
    """km arrives keepdim (core.py:612) or squeezed (quant.py:99); the C ABI wants [B,H,D] contiguous."""
    if km is None:
        return None
    if km.dim() == 4:
        km = km.squeeze(1) if tensor_layout == "NHD" else km.squeeze(2)
    return km.contiguous()


def _quant(x, tensor_layout, gran, is_key, blk, warp, mult, rounding, mean=None, dot_vec=None, dot_group=1,
           dense_heads=False, out=None, scale=None):
    """dense_heads: for an NHD input, give the int8 result head-major storage ([B,H,N,D] contiguous, returned as its
    [B,N,H,D] view): it is an internal operand of the attention kernel, which then streams K/Q rows of one head from
    consecutive lines instead of one line per H*D bytes (the C ABI takes strides per tensor).
    out / scale: caller-allocated results (the reference's pybind convention, quant.py:73-90; the ring quantizes
    straight into its send buffer)."""
    B, H, N, D = L.dims(x, tensor_layout)
    nblk = (N + blk - 1) // blk
    if gran == L.GRAN_PER_BLOCK:
        G = nblk
    elif gran == L.GRAN_PER_WARP:
        G = nblk * (blk // warp)
    else:
        G = nblk * (blk // warp) * (4 if is_key else 8)
    if out is None:
        if dense_heads and tensor_layout == "NHD":
            out = torch.empty((B, H, N, D), dtype=torch.int8, device=x.device).transpose(1, 2)
        else:
            out = torch.empty(x.shape, dtype=torch.int8, device=x.device)
    elif out.dtype != torch.int8 or out.shape != x.shape or out.device != x.device or out.stride(-1) != 1:
        raise ValueError("out must be an int8 tensor of the input's shape on its device with a contiguous last dim")
    if scale is None:
        scale = torch.empty((B, H, G), dtype=torch.float32, device=x.device)
    elif scale.dtype != torch.float32 or scale.numel() != B * H * G or not scale.is_contiguous():
        raise ValueError(f"scale must be a contiguous float32 tensor with {B * H * G} elements")
    dot = torch.empty((B, H, N), dtype=torch.float32, device=x.device) if dot_vec is not None else None
    xd, od = L.desc(x, tensor_layout), L.desc(out, tensor_layout)
    st = L.lib().sage_quant_qk_int8(xd, L.dtype_code(x.dtype), B, H, N, D, L.ptr(mean), od, scale.data_ptr(),
                                    gran, int(is_key), blk, warp, float(mult), rounding,
                                    L.ptr(dot_vec), dot_group, L.ptr(dot), L.stream_ptr(x.device))
    L.check(st, "sage_quant_qk_int8")
    return out, scale, dot


def k_mean(k: torch.Tensor, tensor_layout: str = "HND") -> torch.Tensor:
    """``k.mean(dim=seq)`` of core.py:612 as one deterministic HIP reduction; returns [B,Hk,D] in k's dtype."""
    B, H, N, D = L.dims(k, tensor_layout)
    lib = L.lib()
    ws = torch.empty(max(1, lib.sage_k_mean_workspace_bytes(B, H, N, D) // 4), dtype=torch.float32, device=k.device)
    km = torch.empty((B, H, D), dtype=k.dtype, device=k.device)
    L.check(lib.sage_k_mean(L.desc(k, tensor_layout), L.dtype_code(k.dtype), B, H, N, D, km.data_ptr(), ws.data_ptr(),
                            L.stream_ptr(k.device)), "sage_k_mean")
    return km


def k_smooth_quant(k: torch.Tensor, tensor_layout: str, gran: int, rounding: int, dense_heads: bool = True):
    """``km = k.mean(seq)`` (core.py:612) and the INT8 quantization of ``k - km`` (K half of core.py:621-624) as one call
    of the library (sage_k_smooth_quant): bit-identical to ``k_mean`` + ``_quant(..., mean=km)``, two launches at every
    length (the quantizer finishes the mean from at most 16 chunk sums).  Returns (k_int8, k_scale, km [B,H,D])."""
    B, H, N, D = L.dims(k, tensor_layout)
    if dense_heads and tensor_layout == "NHD":
        out = torch.empty((B, H, N, D), dtype=torch.int8, device=k.device).transpose(1, 2)
    else:
        out = torch.empty(k.shape, dtype=torch.int8, device=k.device)
    G = (N + 63) // 64 * (4 if gran == L.GRAN_PER_THREAD else 1)
    scale = torch.empty((B, H, G), dtype=torch.float32, device=k.device)
    km = torch.empty((B, H, D), dtype=k.dtype, device=k.device)
    lib = L.lib()
    ws = torch.empty(max(1, lib.sage_k_mean_workspace_bytes(B, H, N, D) // 4), dtype=torch.float32, device=k.device)
    L.check(lib.sage_k_smooth_quant(L.desc(k, tensor_layout), L.dtype_code(k.dtype), B, H, N, D, L.desc(out, tensor_layout),
                                    scale.data_ptr(), km.data_ptr(), gran, rounding, ws.data_ptr(), L.stream_ptr(k.device)),
            "sage_k_smooth_quant")
    return out, scale, km


def kv_prepare_fp8(k: torch.Tensor, v: torch.Tensor, tensor_layout: str, gran: int, rounding: int, scale_max: float = 448.0):
    """The K/V side of the FP8-PV operator's pre-pass as one call of the library (sage_kv_prepare_fp8): ``k_smooth_quant(k)``
    and ``per_channel_fp8(v, smooth_v=False)``, bit-identical to them, two launches at every length (five separate ones before).
    Returns (k_int8, k_scale, km, v_fp8, v_scale)."""
    B, H, N, D = L.dims(k, tensor_layout)
    assert L.dims(v, tensor_layout) == (B, H, N, D), "k and v must have the same shape"
    npad = (N + 63) // 64 * 64
    if tensor_layout == "HND":
        k8 = torch.empty(k.shape, dtype=torch.int8, device=k.device)
        v8 = torch.empty((B, H, D, npad), dtype=torch.float8_e4m3fn, device=v.device)
        vd = L.SageTensor(v8.data_ptr(), v8.stride(0), v8.stride(1), v8.stride(2))
    else:
        k8 = torch.empty((B, H, N, D), dtype=torch.int8, device=k.device).transpose(1, 2)
        v8 = torch.empty((B, D, H, npad), dtype=torch.float8_e4m3fn, device=v.device)
        vd = L.SageTensor(v8.data_ptr(), v8.stride(0), v8.stride(2), v8.stride(1))
    G = (N + 63) // 64 * (4 if gran == L.GRAN_PER_THREAD else 1)
    ks = torch.empty((B, H, G), dtype=torch.float32, device=k.device)
    km = torch.empty((B, H, D), dtype=k.dtype, device=k.device)
    v_scale = torch.empty((B, H, D), dtype=torch.float32, device=v.device)
    lib = L.lib()
    ws = torch.empty(max(1, lib.sage_kv_prepare_fp8_workspace_bytes(B, H, N, D) // 4), dtype=torch.float32, device=k.device)
    L.check(lib.sage_kv_prepare_fp8(L.desc(k, tensor_layout), L.desc(v, tensor_layout), L.dtype_code(k.dtype), B, H, N, D,
                                    L.desc(k8, tensor_layout), ks.data_ptr(), km.data_ptr(), gran, rounding, vd,
                                    v_scale.data_ptr(), float(scale_max), ws.data_ptr(), L.stream_ptr(k.device)),
            "sage_kv_prepare_fp8")
    return k8, ks, km, v8, v_scale


def per_block_int8(q, k, km=None, BLKQ=128, BLKK=64, sm_scale=None, tensor_layout="HND", rounding="cuda"):
    """quant.py:23-104.  ``rounding="triton"`` gives the numerics of triton/quant_per_block.py:48-101."""
    D = q.size(-1)
    if sm_scale is None:
        sm_scale = D ** -0.5
    rnd = L.ROUND_CUDA if rounding == "cuda" else L.ROUND_TRITON
    q8, qs, _ = _quant(q, tensor_layout, L.GRAN_PER_BLOCK, False, BLKQ, BLKQ, sm_scale * 1.44269504, rnd)
    k8, ks, _ = _quant(k, tensor_layout, L.GRAN_PER_BLOCK, True, BLKK, BLKK, 1.0, rnd, mean=_km3(km, tensor_layout))
    return q8, qs, k8, ks


def per_warp_int8(q, k, km=None, BLKQ=128, WARPQ=32, BLKK=64, tensor_layout="HND"):
    """quant.py:106-181 (CUDA numerics: csrc/fused/fused.cu:685-768 for Q, :594-682 for K)."""
    q8, qs, _ = _quant(q, tensor_layout, L.GRAN_PER_WARP, False, BLKQ, WARPQ, 1.0, L.ROUND_CUDA)
    k8, ks, _ = _quant(k, tensor_layout, L.GRAN_PER_BLOCK, True, BLKK, BLKK, 1.0, L.ROUND_CUDA, mean=_km3(km, tensor_layout))
    return q8, qs, k8, ks


def per_thread_int8(q, k, km=None, BLKQ=128, WARPQ=32, BLKK=64, WARPK=64, sm_scale=None, tensor_layout="HND"):
    """sageattention/triton/quant_per_thread.py:158-207 (sm_scale unused there as well, :187-188)."""
    q8, qs, _ = _quant(q, tensor_layout, L.GRAN_PER_THREAD, False, BLKQ, WARPQ, 1.0, L.ROUND_TRITON)
    k8, ks, _ = _quant(k, tensor_layout, L.GRAN_PER_THREAD, True, BLKK, WARPK, 1.0, L.ROUND_TRITON,
                       mean=_km3(km, tensor_layout))
    return q8, qs, k8, ks


def sub_mean(v: torch.Tensor, tensor_layout: str = "HND"):
    """quant.py:183-223: returns (fp16(v - mean_seq(v)), vm [B,H,D] in v's dtype)."""
    B, H, N, D = L.dims(v, tensor_layout)
    vm = k_mean(v, tensor_layout)
    out = torch.empty(v.shape, dtype=torch.float16, device=v.device)
    L.check(L.lib().sage_sub_mean_f16(L.desc(v, tensor_layout), L.dtype_code(v.dtype), B, H, N, D, vm.data_ptr(),
                                      L.desc(out, tensor_layout), L.stream_ptr(v.device)), "sage_sub_mean_f16")
    return out, vm


def fp8_token_order() -> torch.Tensor:
    """perm[pos] = token held at position ``pos`` of every 64-token block of the fp8 V^T tensor ("MFMA order",
    csrc/sage_fp8.hip).  It plays the role of the reference's NVIDIA-fragment permutation (quant.py:234) for the
    gfx950 PV MFMA; consumers other than this library's attention kernel must undo it."""
    pos = torch.arange(64)
    h, j = pos >> 5, pos & 31
    return 32 * (j >> 4) + (j & 3) + 8 * ((j & 15) >> 2) + 4 * h


def per_channel_fp8(v: torch.Tensor, tensor_layout: str = "HND", scale_max: float = 448.0, smooth_v: bool = True):
    """quant.py:225-322.  Returns (v_fp8 [B,H,D,ceil64(N)] (HND) / [B,D,H,ceil64(N)] (NHD) float8_e4m3fn (OCP, the
    gfx950 MFMA format; the fork emits fnuz for gfx942) with the tokens of each 64-block in ``fp8_token_order()``,
    v_scale fp32 [B,H,D], vm fp32 [B,H,D] or None)."""
    B, H, N, D = L.dims(v, tensor_layout)
    npad = (N + 63) // 64 * 64
    if tensor_layout == "HND":
        v8 = torch.empty((B, H, D, npad), dtype=torch.float8_e4m3fn, device=v.device)
        od = L.SageTensor(v8.data_ptr(), v8.stride(0), v8.stride(1), v8.stride(2))
    else:
        v8 = torch.empty((B, D, H, npad), dtype=torch.float8_e4m3fn, device=v.device)
        od = L.SageTensor(v8.data_ptr(), v8.stride(0), v8.stride(2), v8.stride(1))
    v_scale = torch.empty((B, H, D), dtype=torch.float32, device=v.device)
    vm = torch.empty((B, H, D), dtype=torch.float32, device=v.device) if smooth_v else None
    lib = L.lib()
    ws = torch.empty(max(1, lib.sage_quant_v_fp8_workspace_bytes(B, H, N, D) // 4), dtype=torch.float32, device=v.device)
    L.check(lib.sage_quant_v_fp8(L.desc(v, tensor_layout), L.dtype_code(v.dtype), B, H, N, D, od, v_scale.data_ptr(),
                                 L.ptr(vm), float(scale_max), ws.data_ptr(), L.stream_ptr(v.device)), "sage_quant_v_fp8")
    return v8, v_scale, vm
