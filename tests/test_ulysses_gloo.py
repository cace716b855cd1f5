"""Ulysses (head-parallel) attention under gloo, world_size 2 and 4, on CPU.  The local operator is replaced by the
oracle; under test is sageattention_amd/ulysses.py: the two all-to-alls and their index maps (sequence shards <->
head shards, GQA head grouping, LSE return path).  The exchange only moves data, so the sharded result must equal the
unsharded operator bit for bit."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _inputs(B, Hq, Hk, N, D, seed=11):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, Hq, N, D, generator=g).to(torch.float16)
    k = (torch.randn(B, Hk, N, D, generator=g) + 2.0 * torch.randn(1, Hk, 1, D, generator=g)).to(torch.float16)
    v = torch.randn(B, Hk, N, D, generator=g).to(torch.float16)
    return q, k, v


def _oracle_attn(pv):
    from oracle import sage_oracle as O

    def fn(q, k, v, tensor_layout="HND", is_causal=False, sm_scale=None, return_lse=False, **kw):
        return O.sageattn_oracle(q, k, v, tensor_layout=tensor_layout, is_causal=is_causal, sm_scale=sm_scale, pv=pv,
                                 return_lse=return_lse)
    return fn


def _worker(rank, world, port, cfg, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sageattention_amd.ulysses import ulysses_sageattn
    B, Hq, Hk, N, D, causal, pv, layout = cfg
    q, k, v = _inputs(B, Hq, Hk, N, D)
    n = N // world
    sl = slice(rank * n, (rank + 1) * n)
    ql, kl, vl = q[:, :, sl], k[:, :, sl], v[:, :, sl]
    if layout == "NHD":
        ql, kl, vl = (t.transpose(1, 2).contiguous() for t in (ql, kl, vl))
    o, lse = ulysses_sageattn(ql, kl, vl, tensor_layout=layout, is_causal=causal, return_lse=True, pv=pv,
                              attn_fn=_oracle_attn(pv))
    if layout == "NHD":
        o = o.transpose(1, 2)
    torch.save({"o": o.contiguous(), "lse": lse}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,Hq,Hk,causal,pv,layout", [
    (2, 4, 2, False, "fp16", "HND"), (2, 4, 4, True, "fp8", "NHD"), (4, 8, 4, True, "fp16", "HND")])
def test_ulysses_equals_unsharded_operator(tmp_path, world, Hq, Hk, causal, pv, layout):
    cfg = (1, Hq, Hk, 64 * world, 64, causal, pv, layout)
    mp.spawn(_worker, args=(world, _free_port(), cfg, str(tmp_path)), nprocs=world, join=True)
    B, _, _, N, D = cfg[:5]
    q, k, v = _inputs(B, Hq, Hk, N, D)
    ref_o, ref_lse = _oracle_attn(pv)(q, k, v, is_causal=causal, return_lse=True)
    outs = [torch.load(os.path.join(tmp_path, f"r{r}.pt"), weights_only=True) for r in range(world)]
    o = torch.cat([x["o"] for x in outs], dim=2)
    lse = torch.cat([x["lse"] for x in outs], dim=2)
    assert torch.equal(o, ref_o)
    assert torch.equal(lse, ref_lse)


def test_ulysses_rejects_indivisible_heads():
    from sageattention_amd.ulysses import ulysses_sageattn
    q = torch.zeros(1, 3, 8, 64, dtype=torch.float16)
    with pytest.raises(ValueError):
        ulysses_sageattn(q, q, q, tensor_layout="XYZ")
