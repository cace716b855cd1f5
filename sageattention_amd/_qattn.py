"""Tensor-level shims with the names and positional signatures of the reference's pybind modules
``sageattention._qattn_sm80`` (csrc/qattn/attn_cuda_sm80.h:19-65, pybind_sm80.cpp) and
``sageattention._qattn_sm89`` / ``_qattn_rocm`` (attn_cuda_sm89.h, attn_rocm_gfx942.h:20-32), forwarding to the
C ABI (include/sageattn_hip.h).  ``lse`` is allocated here and returned, like the reference
(qk_int_sv_f16_cuda_sm80.cu:776-780): base-2 LSE of the scaled, smoothed logits, or an empty tensor."""
import torch

from . import _lib as L


def _blkq_warpq(query_scale, query, tensor_layout, qk_quant_gran):
    """Recover (BLKQ, WARPQ) from the scale tensor shape (…sm80.cu:796-805: BLKQ=128, WARPQ in {32,16})."""
    M = L.dims(query, "NHD" if tensor_layout == 0 else "HND")[2]
    nblk = (M + 127) // 128
    g = query_scale.size(-1)
    if qk_quant_gran == L.GRAN_PER_BLOCK:
        return 128, 128
    per = g // nblk
    if qk_quant_gran == L.GRAN_PER_THREAD:
        per //= 8
    if per not in (1, 2, 4, 8) or (g % nblk):
        raise ValueError(f"query_scale has {g} scales per head, inconsistent with qo_len={M}")
    return 128, 128 // per


def _attn_f16(query, key, value, output, query_scale, key_scale, value_mean, tensor_layout, is_causal, qk_quant_gran,
              sm_scale, return_lse, logit_mult_is_one=False):
    layout = "NHD" if tensor_layout == 0 else "HND" if tensor_layout == 1 else None
    if layout is None:
        raise ValueError("tensor_layout must be 0 or 1")
    for name, t in (("query", query), ("key", key), ("value", value), ("output", output),
                    ("query_scale", query_scale), ("key_scale", key_scale)):
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA/HIP tensor")
    if query.dtype != torch.int8 or key.dtype != torch.int8:
        raise RuntimeError("query and key must be int8")
    if query_scale.dtype != torch.float32 or key_scale.dtype != torch.float32:
        raise RuntimeError("scales must be float32")
    B, Hq, M, D = L.dims(query, layout)
    _, Hk, N, _ = L.dims(key, layout)
    if Hq % Hk != 0:
        raise ValueError(f"num_qo_heads ({Hq}) must be divisible by num_kv_heads ({Hk})")
    blkq, warpq = _blkq_warpq(query_scale, query, tensor_layout, qk_quant_gran)
    lse = torch.empty((B, Hq, M), dtype=torch.float32, device=query.device) if return_lse else \
        torch.empty((0,), dtype=torch.float32, device=query.device)
    vm = value_mean.to(torch.float32).contiguous() if value_mean is not None else None
    st = L.lib().sage_attn_qk_int8_pv_f16(
        L.desc(query, layout), L.desc(key, layout), L.desc(value, layout), L.dtype_code(value.dtype),
        L.desc(output, layout), L.dtype_code(output.dtype), query_scale.contiguous().data_ptr(),
        key_scale.contiguous().data_ptr(), L.ptr(vm), lse.data_ptr() if return_lse else None,
        B, Hq, Hk, M, N, D, int(is_causal), int(qk_quant_gran), blkq, warpq, float(sm_scale),
        int(logit_mult_is_one), L.stream_ptr(query.device))
    L.check(st, "sage_attn_qk_int8_pv_f16")
    return lse


def qk_int8_sv_f16_accum_f32_attn(query, key, value, output, query_scale, key_scale, tensor_layout, is_causal,
                                  qk_quant_gran, sm_scale, return_lse):
    """attn_cuda_sm80.h:19-29."""
    return _attn_f16(query, key, value, output, query_scale, key_scale, None, tensor_layout, is_causal,
                     qk_quant_gran, sm_scale, return_lse)


# On gfx950 the PV MFMA always accumulates in fp32: the fp16-accumulate entry points of the reference
# (attn_cuda_sm80.h:31-53) map onto the same kernel.
qk_int8_sv_f16_accum_f16_attn = qk_int8_sv_f16_accum_f32_attn
qk_int8_sv_f16_accum_f16_attn_inst_buf = qk_int8_sv_f16_accum_f32_attn


def qk_int8_sv_f16_accum_f16_fuse_v_mean_attn(query, key, value, output, query_scale, key_scale, value_mean,
                                              tensor_layout, is_causal, qk_quant_gran, sm_scale, return_lse):
    """attn_cuda_sm80.h:55-65."""
    return _attn_f16(query, key, value, output, query_scale, key_scale, value_mean, tensor_layout, is_causal,
                     qk_quant_gran, sm_scale, return_lse)


def _attn_f8(query, key, value, output, query_scale, key_scale, value_scale, value_mean, tensor_layout, is_causal,
             qk_quant_gran, sm_scale, return_lse):
    layout = "NHD" if tensor_layout == 0 else "HND" if tensor_layout == 1 else None
    if layout is None:
        raise ValueError("tensor_layout must be 0 or 1")
    if query.dtype != torch.int8 or key.dtype != torch.int8:
        raise RuntimeError("query and key must be int8")
    if value.dtype != torch.float8_e4m3fn:
        raise RuntimeError("value must be float8_e4m3fn (OCP) on gfx950")
    B, Hq, M, D = L.dims(query, layout)
    _, Hk, N, _ = L.dims(key, layout)
    if Hq % Hk != 0:
        raise ValueError(f"num_qo_heads ({Hq}) must be divisible by num_kv_heads ({Hk})")
    blkq, warpq = _blkq_warpq(query_scale, query, tensor_layout, qk_quant_gran)
    lse = torch.empty((B, Hq, M), dtype=torch.float32, device=query.device) if return_lse else \
        torch.empty((0,), dtype=torch.float32, device=query.device)
    # value: [B,Hk,D,Npad] (HND) or [B,D,Hk,Npad] (NHD) -> descriptor with stride_n := stride of the d index
    if layout == "HND":
        vd = L.SageTensor(value.data_ptr(), value.stride(0), value.stride(1), value.stride(2))
    else:
        vd = L.SageTensor(value.data_ptr(), value.stride(0), value.stride(2), value.stride(1))
    vm = value_mean.to(torch.float32).contiguous() if value_mean is not None else None
    st = L.lib().sage_attn_qk_int8_pv_f8(
        L.desc(query, layout), L.desc(key, layout), vd, L.desc(output, layout), L.dtype_code(output.dtype),
        query_scale.contiguous().data_ptr(), key_scale.contiguous().data_ptr(),
        value_scale.contiguous().data_ptr(), L.ptr(vm), lse.data_ptr() if return_lse else None,
        B, Hq, Hk, M, N, D, int(is_causal), int(qk_quant_gran), blkq, warpq, float(sm_scale), 0,
        L.stream_ptr(query.device))
    L.check(st, "sage_attn_qk_int8_pv_f8")
    return lse


def qk_int8_sv_f8_accum_f32_fuse_v_scale_attn(query, key, value, output, query_scale, key_scale, value_scale,
                                              tensor_layout, is_causal, qk_quant_gran, sm_scale, return_lse):
    """attn_cuda_sm89.h (fuse_v_scale) and the fork's attn_rocm_gfx942.h:20-32."""
    return _attn_f8(query, key, value, output, query_scale, key_scale, value_scale, None, tensor_layout, is_causal,
                    qk_quant_gran, sm_scale, return_lse)


qk_int8_sv_f8_accum_f32_attn = qk_int8_sv_f8_accum_f32_fuse_v_scale_attn  # fork name (core.py:893)
qk_int8_sv_f8_accum_f32_fuse_v_scale_attn_inst_buf = qk_int8_sv_f8_accum_f32_fuse_v_scale_attn
qk_int8_sv_f8_accum_f16_fuse_v_scale_attn_inst_buf = qk_int8_sv_f8_accum_f32_fuse_v_scale_attn


def qk_int8_sv_f8_accum_f32_fuse_v_scale_fuse_v_mean_attn(query, key, value, output, query_scale, key_scale,
                                                          value_scale, value_mean, tensor_layout, is_causal,
                                                          qk_quant_gran, sm_scale, return_lse):
    return _attn_f8(query, key, value, output, query_scale, key_scale, value_scale, value_mean, tensor_layout,
                    is_causal, qk_quant_gran, sm_scale, return_lse)
