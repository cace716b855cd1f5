"""sageattention_amd -- MI355X (gfx950) native drop-in for the SageAttention quantized attention operator.

Exports the names the reference package exports (sageattention/__init__.py:25-95): ``sageattn`` plus the
``sageattn_qk_int8_*`` entry points that diffusers imports by name."""
from .core import (sageattn, sageattn_qk_int8_pv_fp16_cuda, sageattn_qk_int8_pv_fp16_triton,
                   sageattn_qk_int8_pv_fp8_cuda, sageattn_qk_int8_pv_fp8_cuda_sm90, sageattn_varlen)
from . import quant, _qattn, _fused  # noqa: F401
from .ring import ring_sageattn  # noqa: F401  (sequence parallel, RCCL send/recv)
from .ulysses import ulysses_sageattn  # noqa: F401  (head parallel, RCCL all-to-all)

__all__ = ["sageattn", "sageattn_qk_int8_pv_fp16_cuda", "sageattn_qk_int8_pv_fp16_triton",
           "sageattn_qk_int8_pv_fp8_cuda", "sageattn_qk_int8_pv_fp8_cuda_sm90", "sageattn_varlen", "ring_sageattn",
           "ulysses_sageattn"]
__version__ = "0.2.0"
