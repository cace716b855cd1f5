"""Ulysses (head-parallel) SageAttention over ``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm).

New component: the reference delegates Ulysses to xDiT (example/parallel_sageattn_cogvideo.py:44-58,
``ulysses_degree``) and only supplies the attention callable.  Here the exchange is part of the package.

Scheme.  Rank r holds the rows [r*n, (r+1)*n) of q, k, v for ALL heads.  One all-to-all turns that into the WHOLE
sequence for the heads [r*H/P, (r+1)*H/P); the unmodified single-device operator runs on them -- K smoothing mean,
quantizer groups and tile loop see exactly the tensors the unsharded call would see for those heads, so the result is
bit-identical to the single-GPU operator -- and a second all-to-all returns the output rows (and LSE) to their owner.
Per rank and call the fabric carries (P-1)/P of q, k, v, o (fp16/bf16); an all-to-all puts 1/P of that on each of the
P-1 point-to-point xGMI links of the node at the same time, which is the pattern the fully connected MI355X node is
built for.  No causal load imbalance (every rank sees the whole sequence).  Needs Hq % P == 0 and Hk % P == 0;
sequences too long for one GPU's activations, or with fewer KV heads than ranks, use ``ring_sageattn`` instead.
"""
from typing import Any, Callable, Optional

import torch
import torch.distributed as dist

__all__ = ["ulysses_sageattn"]


def _all_to_all(x: torch.Tensor, group) -> torch.Tensor:
    out = torch.empty_like(x)
    dist.all_to_all_single(out, x, group=group)
    return out


def _seq_to_head(x: torch.Tensor, world: int, group) -> torch.Tensor:
    """[B,H,n,...] (my rows, all heads) -> [B,H/P,P*n,...] (all rows, my heads)."""
    B, H, n = x.shape[:3]
    tail = tuple(x.shape[3:])
    xs = x.reshape(B, world, H // world, n, *tail).movedim(1, 0).contiguous()      # [P,B,H/P,n,...] chunk p -> rank p
    y = _all_to_all(xs, group)                                                     # chunk p <- rank p (its rows)
    return y.movedim(0, 2).reshape(B, H // world, world * n, *tail)


def _head_to_seq(y: torch.Tensor, world: int, group) -> torch.Tensor:
    """[B,H/P,P*n,...] (all rows, my heads) -> [B,H,n,...] (my rows, all heads)."""
    B, Hl, N = y.shape[:3]
    tail = tuple(y.shape[3:])
    ys = y.reshape(B, Hl, world, N // world, *tail).movedim(2, 0).contiguous()     # [P,B,H/P,n,...] chunk p -> rank p
    x = _all_to_all(ys, group)                                                     # chunk p <- rank p (its heads)
    return x.movedim(0, 1).reshape(B, world * Hl, N // world, *tail)


@torch.compiler.disable
def ulysses_sageattn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, tensor_layout: str = "HND",
                     is_causal: bool = False, sm_scale: Optional[float] = None,
                     group: Optional[dist.ProcessGroup] = None, pv: str = "auto", qk_quant_gran: str = "per_thread",
                     return_lse: bool = False, attn_fn: Optional[Callable] = None, **kwargs: Any):
    """SageAttention over a sequence sharded across the ranks of ``group`` (rank r holds rows [r*n, (r+1)*n); equal
    shard lengths), parallel over heads.  Same tensor conventions as ``sageattn``; returns this rank's output rows
    (and their LSE).  ``attn_fn`` replaces the local operator (tests run the CPU oracle under gloo)."""
    if tensor_layout == "NHD":
        q, k, v = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    elif tensor_layout != "HND":
        raise ValueError(f"Unknown tensor layout: {tensor_layout}")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if pv == "auto":  # the dispatcher's rule on the whole sequence
        from .core import dispatch_pv
        pv = dispatch_pv(q, k, "HND", is_causal, n_kv=k.size(2) * world)
    if pv not in ("fp16", "fp8"):
        raise ValueError(f"Unknown pv: {pv}")
    Hq, Hk = q.size(1), k.size(1)
    if Hq % world or Hk % world:
        raise ValueError(f"ulysses_sageattn needs head counts divisible by the group size: Hq={Hq}, Hk={Hk}, P={world}")
    if attn_fn is None:
        from . import core
        attn_fn = core.sageattn_qk_int8_pv_fp16_cuda if pv == "fp16" else core.sageattn_qk_int8_pv_fp8_cuda
        kwargs = dict(kwargs, qk_quant_gran=qk_quant_gran)
    if world > 1:
        q, k, v = (_seq_to_head(t, world, group) for t in (q, k, v))
    res = attn_fn(q, k, v, tensor_layout="HND", is_causal=is_causal, sm_scale=sm_scale, return_lse=return_lse, **kwargs)
    o, lse = res if return_lse else (res, None)
    if world > 1:
        o = _head_to_seq(o.contiguous(), world, group)
        if lse is not None:
            lse = _head_to_seq(lse.contiguous(), world, group)
    if tensor_layout == "NHD":
        o = o.transpose(1, 2)
    return (o, lse) if return_lse else o
