"""``torch.library`` registration of the operator, so that ``torch.compile`` traces THROUGH a model that calls it
instead of breaking the graph.  The reference marks every entry point ``@torch.compiler.disable``
(core.py:160,362,479,655,907), which forces a graph break around each attention call.

    torch.ops.sageattention_amd.attn(q, k, v, tensor_layout, is_causal, sm_scale, pv, qk_quant_gran) -> o
    torch.ops.sageattention_amd.attn_lse(...) -> (o, lse)

The bodies call the same host code as ``sageattn_qk_int8_pv_{fp16,fp8}_cuda`` (core.py) and therefore the same HIP
kernels; the fake (meta) implementations only describe shapes, dtypes and strides.  ``sageattn_compilable`` is the
keyword-friendly wrapper with the reference's signature."""
from typing import Any, Optional, Tuple

import torch

from . import core

__all__ = ["sageattn_compilable"]


def _entry(pv: str):
    if pv == "fp16":
        return core.sageattn_qk_int8_pv_fp16_cuda
    if pv == "fp8":
        return core.sageattn_qk_int8_pv_fp8_cuda
    raise ValueError(f"Unknown pv: {pv}")


@torch.library.custom_op("sageattention_amd::attn", mutates_args=())
def attn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, tensor_layout: str, is_causal: bool, sm_scale: float,
         pv: str, qk_quant_gran: str) -> torch.Tensor:
    # contiguous result whatever the inputs' strides / head-dim padding: the fake implementation below must describe
    # exactly the layout the real op returns, or Inductor indexes the output with the wrong strides
    return _entry(pv)(q, k, v, tensor_layout=tensor_layout, is_causal=is_causal, sm_scale=sm_scale,
                      qk_quant_gran=qk_quant_gran).contiguous()


@attn.register_fake
def _(q, k, v, tensor_layout, is_causal, sm_scale, pv, qk_quant_gran):
    return q.new_empty(q.shape)


@torch.library.custom_op("sageattention_amd::attn_lse", mutates_args=())
def attn_lse(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, tensor_layout: str, is_causal: bool, sm_scale: float,
             pv: str, qk_quant_gran: str) -> Tuple[torch.Tensor, torch.Tensor]:
    o, lse = _entry(pv)(q, k, v, tensor_layout=tensor_layout, is_causal=is_causal, sm_scale=sm_scale,
                        qk_quant_gran=qk_quant_gran, return_lse=True)
    return o.contiguous(), lse


@attn_lse.register_fake
def _(q, k, v, tensor_layout, is_causal, sm_scale, pv, qk_quant_gran):
    if tensor_layout == "HND":
        B, H, M = q.shape[0], q.shape[1], q.shape[2]
    else:
        B, M, H = q.shape[0], q.shape[1], q.shape[2]
    return q.new_empty(q.shape), q.new_empty((B, H, M), dtype=torch.float32)


def sageattn_compilable(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, tensor_layout: str = "HND",
                        is_causal: bool = False, sm_scale: Optional[float] = None, return_lse: bool = False,
                        pv: str = "auto", qk_quant_gran: str = "per_thread", **kwargs: Any):
    """``sageattn`` (core.py:80-144) as a traceable custom op.  ``pv="auto"`` follows the dispatcher of the package
    (``core.dispatch_pv``: FP8 PV from a few thousand keys per row upwards, FP16 PV below; per_thread scales); unknown
    keyword arguments are accepted and ignored like there."""
    if tensor_layout not in ("HND", "NHD"):
        raise ValueError(f"Unknown tensor layout: {tensor_layout}")
    if pv == "auto":
        pv = core.dispatch_pv(q, k, tensor_layout, bool(is_causal))
    if sm_scale is None:
        sm_scale = q.size(-1) ** -0.5
    if return_lse:
        return torch.ops.sageattention_amd.attn_lse(q, k, v, tensor_layout, bool(is_causal), float(sm_scale), pv,
                                                    qk_quant_gran)
    return torch.ops.sageattention_amd.attn(q, k, v, tensor_layout, bool(is_causal), float(sm_scale), pv, qk_quant_gran)
