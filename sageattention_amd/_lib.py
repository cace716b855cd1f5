"""ctypes binding of the C-ABI library ``libsageattn_hip.so`` (include/sageattn_hip.h).

PyTorch is plumbing only: it owns device memory and the current HIP stream; every compute step of the
hot path goes through the C ABI.  There is NO CPU or eager fallback: if the library is missing or the
call fails, this raises."""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsageattn_hip.so")

SAGE_F16, SAGE_BF16 = 0, 1
GRAN_PER_BLOCK, GRAN_PER_WARP, GRAN_PER_THREAD = 1, 2, 3
ROUND_TRITON, ROUND_CUDA = 0, 1


class SageTensor(ctypes.Structure):
    _fields_ = [("data", c_void_p), ("stride_b", c_int64), ("stride_h", c_int64), ("stride_n", c_int64)]


class KvLayout(ctypes.Structure):
    """sage_kv_layout (include/sageattn_hip.h): tile strides of k8 / v and the strides of the k scales."""
    _fields_ = [("k_tile_stride", c_int64), ("v_tile_stride", c_int64), ("ks_stride_b", c_int64),
                ("ks_stride_h", c_int64), ("ks_stride_tile", c_int64)]


class OpOpts(ctypes.Structure):
    """sage_op_opts (include/sageattn_hip.h)"""
    _fields_ = [("qk_gran", c_int), ("warpq", c_int), ("smooth_k", c_int), ("fuse_q", c_int), ("nwaves", c_int),
                ("reserved", c_int * 3)]


_P = ctypes.POINTER(SageTensor)
_PO = ctypes.POINTER(OpOpts)
_PL = ctypes.POINTER(KvLayout)
_lib = None

# name -> (restype, argtypes); must list every symbol include/sageattn_hip.h declares
SIGNATURES = {
    "sage_abi_version": (c_int, []),
    "sage_status_string": (c_char_p, [c_int]),
    "sage_target_arch": (c_char_p, []),
    "sage_k_mean_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "sage_k_mean": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sage_quant_qk_int8": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_void_p, _P, c_void_p, c_int, c_int, c_int,
                                   c_int, c_float, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "sage_sub_mean_f16": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_void_p, _P, c_void_p]),
    "sage_quant_v_fp8_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "sage_quant_v_fp8": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "sage_attn_qk_int8_pv_f16": (c_int, [_P, _P, _P, c_int, _P, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                         c_float, c_int, c_void_p]),
    "sage_attn_qk_int8_pv_f8": (c_int, [_P, _P, _P, _P, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_float, c_int, c_void_p]),
    "sage_quant_qk_int8_varlen": (c_int, [_P, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, _P, c_void_p, c_int,
                                          c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "sage_attn_qk_int8_pv_f16_varlen": (c_int, [_P, _P, _P, c_int, _P, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                                c_float, c_int, c_void_p]),
    "sage_attn_fusedq_pv_f16": (c_int, [_P, c_int, _P, _P, c_int, _P, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "sage_attn_fusedq_pv_f8": (c_int, [_P, c_int, _P, _P, _P, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "sage_attn_qk_int8_pv_f16_masked": (c_int, [_P, _P, _P, c_int, _P, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                                ctypes.POINTER(c_int64), c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                                c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "sage_merge_attn_states": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p]),
    "sage_merge_attn_states_multi": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "sage_finish_lse": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_int64, c_void_p]),
    "sage_set_tuning": (c_int, [c_int, c_int]),
    "sage_get_tuning": (c_int, [c_int]),
    "sage_sageattn_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _PO]),
    "sage_sageattn_pv_f16": (c_int, [_P, _P, _P, c_int, _P, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                     c_float, _PO, c_void_p, c_size_t, c_void_p]),
    "sage_sageattn_pv_f8": (c_int, [_P, _P, _P, c_int, _P, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                    c_float, c_float, _PO, c_void_p, c_size_t, c_void_p]),
    "sage_k_smooth_quant": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                    c_void_p]),
    "sage_kv_prepare_fp8_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "sage_kv_prepare_fp8": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_void_p, c_void_p, c_int, c_int, _P,
                                    c_void_p, c_float, c_void_p, c_void_p]),
    "sage_attn_qk_int8_pv_f16_kvtiles": (c_int, [_P, _P, _P, c_int, _P, c_int, c_void_p, c_void_p, _PL, c_void_p,
                                                 c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                                 c_float, c_void_p]),
    "sage_attn_qk_int8_pv_f8_kvtiles": (c_int, [_P, _P, _P, _P, c_int, c_void_p, c_void_p, c_void_p, _PL, c_void_p,
                                                c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                                c_float, c_void_p]),
    "sage_seq_stats_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "sage_seq_stats": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sage_kv_stats_reduce": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int64, c_int, c_float, c_void_p,
                                     c_void_p, c_void_p, c_void_p]),
    "sage_quant_k_int8_kvtiles": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_void_p, _P, c_int64, c_void_p,
                                          ctypes.POINTER(c_int64), c_int, c_int, c_void_p]),
    "sage_quant_v_fp8_apply": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, c_int64, c_void_p, c_void_p]),
    "sage_merge_attn_states_multi_ex": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int64, c_int,
                                                c_float, c_void_p, c_float, c_void_p]),
}


def lib():
    """Load the HIP library (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the gfx950 HIP library has not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `python sageattention_amd/_build.py`). "
                "There is no CPU fallback for the SageAttention hot path.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
        if os.environ.get("SAGE_NWAVES"):  # tuning knob (speed only): waves per attention workgroup, 4 or 8
            l.sage_set_tuning(0, int(os.environ["SAGE_NWAVES"]))
    return _lib


def check(status: int, what: str):
    """Map sage_status to the reference's exception types (TORCH_CHECK -> RuntimeError,
    std::invalid_argument -> ValueError)."""
    if status == 0:
        return
    msg = f"{what}: {lib().sage_status_string(status).decode()} (status {status})"
    if status in (-1, -2):
        raise ValueError(msg)
    raise RuntimeError(msg)


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float16:
        return SAGE_F16
    if dt == torch.bfloat16:
        return SAGE_BF16
    raise ValueError(f"unsupported dtype {dt}")


def desc(t: torch.Tensor, tensor_layout: str) -> SageTensor:
    """[B,H,N,D] descriptor of a 4-D tensor in either reference layout (core.py:585)."""
    assert t.dim() == 4 and t.stride(-1) == 1, "Last dim must be contiguous."
    if tensor_layout == "HND":
        return SageTensor(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2))
    if tensor_layout == "NHD":
        return SageTensor(t.data_ptr(), t.stride(0), t.stride(2), t.stride(1))
    raise ValueError(f"Unknown tensor layout: {tensor_layout}")


def dims(t: torch.Tensor, tensor_layout: str):
    """-> (B, H, N, D)"""
    if tensor_layout == "HND":
        return t.size(0), t.size(1), t.size(2), t.size(3)
    if tensor_layout == "NHD":
        return t.size(0), t.size(2), t.size(1), t.size(3)
    raise ValueError(f"Unknown tensor layout: {tensor_layout}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device) -> int:
    """hipStream_t of torch's current stream on `device` (the raw getter is ~10x cheaper than building a Stream object:
    4 us of the ~40 us a call costs on the host)."""
    if _raw_stream is not None and device.index is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()
