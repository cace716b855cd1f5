"""Tensor-level shim with the names and positional signatures of the reference's pybind module
``sageattention._fused`` (csrc/fused/pybind.cpp:23-32, prototypes in csrc/fused/fused.h), forwarding to the C ABI
(include/sageattn_hip.h).  As there, the caller allocates ``output`` / ``scale`` and the functions return nothing;
``tensor_layout`` is the integer of core.py:585 (0 = NHD, 1 = HND); numerics are the CUDA kernels' (fused.cu:147-184:
reciprocal multiply, round half to even, eps 1e-7).

Not mirrored: ``transpose_pad_permute_cuda`` + ``scale_fuse_quant_cuda`` / ``mean_scale_fuse_quant_cuda``
(fused.cu:850-1083).  That pair communicates through a transposed fp16 copy of V that exists only because the
reference quantizes V in two kernels; here ``quant.per_channel_fp8`` (quant.py:225-322, the function that calls the
pair) is one fused HIP pass straight to the gfx950 FP8 layout, so the intermediate tensor has no counterpart."""
import torch

from . import _lib as L
from .quant import _quant

__all__ = ["quant_per_block_int8_cuda", "quant_per_block_int8_fuse_sub_mean_cuda", "quant_per_warp_int8_cuda",
           "sub_mean_cuda"]


def _layout(tensor_layout: int) -> str:
    if tensor_layout not in (0, 1):
        raise ValueError("tensor_layout must be 0 (NHD) or 1 (HND)")
    return "NHD" if tensor_layout == 0 else "HND"


def _check(input, output, scale):
    if not (input.is_cuda and output.is_cuda and scale.is_cuda):
        raise RuntimeError("input, output and scale must be CUDA/HIP tensors")  # CHECK_CUDA, fused.cu:438-440
    if input.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError("input must be float16 or bfloat16")
    if output.dtype != torch.int8 or scale.dtype != torch.float32:
        raise RuntimeError("output must be int8 and scale float32")          # CHECK_DTYPE, fused.cu:442-443


def quant_per_block_int8_cuda(input, output, scale, *args):
    """Both overloads of fused.cu:429-592: ``(input, output, scale, sm_scale, block_size, tensor_layout)`` scales the
    input by ``sm_scale`` first (Q); ``(input, output, scale, block_size, tensor_layout)`` does not (K)."""
    if len(args) == 3:
        sm_scale, block_size, tensor_layout = args
    elif len(args) == 2:
        (block_size, tensor_layout), sm_scale = args, 1.0
    else:
        raise TypeError("quant_per_block_int8_cuda(input, output, scale, [sm_scale,] block_size, tensor_layout)")
    _check(input, output, scale)
    _quant(input, _layout(tensor_layout), L.GRAN_PER_BLOCK, False, int(block_size), int(block_size), float(sm_scale),
           L.ROUND_CUDA, out=output, scale=scale)


def quant_per_block_int8_fuse_sub_mean_cuda(input, mean, output, scale, block_size, tensor_layout):
    """fused.cu:594-682: ``input - mean`` (fp32 subtraction) quantized per block; ``mean`` is [B,H,D] in input's dtype."""
    _check(input, output, scale)
    if mean.dtype != input.dtype:
        raise RuntimeError("mean must have the dtype of input")
    _quant(input, _layout(tensor_layout), L.GRAN_PER_BLOCK, True, int(block_size), int(block_size), 1.0, L.ROUND_CUDA,
           mean=mean.contiguous(), out=output, scale=scale)


def quant_per_warp_int8_cuda(input, output, scale, block_size, warp_block_size, tensor_layout):
    """fused.cu:685-768: one scale per ``warp_block_size`` rows inside blocks of ``block_size`` rows."""
    _check(input, output, scale)
    _quant(input, _layout(tensor_layout), L.GRAN_PER_WARP, False, int(block_size), int(warp_block_size), 1.0,
           L.ROUND_CUDA, out=output, scale=scale)


def sub_mean_cuda(input, mean, output, tensor_layout):
    """fused.cu:770-848: ``output = fp16(input - mean)``; ``mean`` [B,H,D] in input's dtype, ``output`` float16."""
    layout = _layout(tensor_layout)
    if output.dtype != torch.float16 or output.shape != input.shape:
        raise RuntimeError("output must be a float16 tensor of input's shape")
    B, H, N, D = L.dims(input, layout)
    L.check(L.lib().sage_sub_mean_f16(L.desc(input, layout), L.dtype_code(input.dtype), B, H, N, D,
                                      mean.contiguous().data_ptr(), L.desc(output, layout),
                                      L.stream_ptr(input.device)), "sage_sub_mean_f16")
