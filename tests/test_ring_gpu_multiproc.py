"""The product's multi-process sequence-parallel paths end to end on ONE GPU: 2 and 3 ranks (gloo transport carrying
device tensors; RCCL needs one GPU per rank) run ring_sageattn / ulysses_sageattn with the HIP backend -- real
quantizers, real attention kernels, real multi-way merge, real exchange -- and the gathered result is compared with
exact fp32 attention over the whole sequence and, for Ulysses, bit for bit with the single-process operator."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import calc_diff


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _inputs(N, D, Hq=8, Hk=4, seed=5):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(1, Hq, N, D, generator=g).to(torch.float16)
    k = (torch.randn(1, Hk, N, D, generator=g) + 2.0 * torch.randn(1, Hk, 1, D, generator=g)).to(torch.float16)
    v = torch.randn(1, Hk, N, D, generator=g).to(torch.float16)
    return q, k, v


def _worker(rank, world, port, cfg, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sageattention_amd as sa
    from sageattention_amd.ring import zigzag_split
    mode, N, D, causal, pv, schedule, layout = cfg[:7]
    q, k, v = (t.cuda() for t in _inputs(N, D, *cfg[7:]))
    n = N // world
    if layout == "zigzag":
        ql, kl, vl = (zigzag_split(t, world, rank).contiguous() for t in (q, k, v))
    else:
        ql, kl, vl = (t[:, :, rank * n:(rank + 1) * n].contiguous() for t in (q, k, v))
    if mode == "ring":
        o, lse = sa.ring_sageattn(ql, kl, vl, is_causal=causal, pv=pv, schedule=schedule, causal_layout=layout, return_lse=True)
    else:
        o, lse = sa.ulysses_sageattn(ql, kl, vl, is_causal=causal, pv=pv, return_lse=True)
    torch.save({"o": o.cpu(), "lse": lse.cpu()}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,cfg", [
    (2, ("ring", 1024, 128, False, "fp16", "direct", "contiguous")),
    (3, ("ring", 1536, 64, True, "fp8", "ring", "contiguous")),
    (2, ("ring", 1024, 128, True, "fp16", "direct", "zigzag")),
    (2, ("ulysses", 768, 128, True, "fp8", "", "contiguous")),
    (3, ("ring", 1536, 128, False, "fp8", "gather", "contiguous")),
    (2, ("ring", 1024, 64, True, "fp16", "gather", "contiguous")),
    (2, ("ring", 16384, 128, False, "fp8", "gather", "contiguous", 4, 2)),   # 8192 rows per rank, as C5
    (2, ("ring", 1024, 128, True, "fp16", "gather", "zigzag")),
    (3, ("ring", 1536, 64, True, "fp8", "gather", "zigzag")),
])
def test_multi_process_sequence_parallel_on_one_gpu(tmp_path, world, cfg):
    from oracle import sage_oracle as O
    from sageattention_amd.ring import zigzag_merge
    mp.spawn(_worker, args=(world, _free_port(), cfg, str(tmp_path)), nprocs=world, join=True)
    mode, N, D, causal, pv, schedule, layout = cfg[:7]
    q, k, v = _inputs(N, D, *cfg[7:])
    outs = [torch.load(os.path.join(tmp_path, f"r{r}.pt"), weights_only=True) for r in range(world)]
    if layout == "zigzag":
        o, lse = zigzag_merge([x["o"] for x in outs]).float(), zigzag_merge([x["lse"] for x in outs])
    else:
        o, lse = torch.cat([x["o"] for x in outs], dim=2).float(), torch.cat([x["lse"] for x in outs], dim=2)
    ref, ref_lse = O.sdpa_fp32(q, k, v, is_causal=causal, return_lse=True)
    assert (o - ref).abs().max() < (0.08 if pv == "fp16" else 0.2)
    assert calc_diff(o, ref) < (2e-3 if pv == "fp16" else 5e-3)
    assert (lse - ref_lse).abs().max() < 0.06
    if schedule == "gather":  # whole-sequence smoothing / V scale: the unsharded operator's operands (tests/test_gather_gpu.py)
        import sageattention_amd as sa
        fn = sa.sageattn_qk_int8_pv_fp16_cuda if pv == "fp16" else sa.sageattn_qk_int8_pv_fp8_cuda
        o1, l1 = fn(q.cuda(), k.cuda(), v.cuda(), is_causal=causal, return_lse=True)
        assert (o1.cpu().float() - o).abs().max() < (2e-3 if pv == "fp16" else (6e-2 if causal else 3e-2))
        assert (l1.cpu() - lse).abs().max() < 1e-3
    if mode == "ulysses":  # the exchange only moves data: identical to the single-process operator
        import sageattention_amd as sa
        fn = sa.sageattn_qk_int8_pv_fp16_cuda if pv == "fp16" else sa.sageattn_qk_int8_pv_fp8_cuda
        o1, l1 = fn(q.cuda(), k.cuda(), v.cuda(), is_causal=causal, return_lse=True)
        assert torch.equal(o1.cpu().float(), o) and torch.equal(l1.cpu(), lse)
