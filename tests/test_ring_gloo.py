"""Ring sequence-parallel attention under gloo, world_size 2 (and 3), on CPU: the N>1 path of bench.py.
The device steps are replaced by the oracle (tests/ring_cpu_backend.py); what is under test is the product's ring
driver sageattention_amd/ring.py: buffer packing, rotation order, causal shard skipping and the LSE merge."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import calc_diff


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _inputs(B, Hq, Hk, N, D, seed=0):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, Hq, N, D, generator=g).to(torch.float16)
    k = (torch.randn(B, Hk, N, D, generator=g) + 2.0 * torch.randn(1, Hk, 1, D, generator=g)).to(torch.float16)
    v = torch.randn(B, Hk, N, D, generator=g).to(torch.float16)
    return q, k, v


def _worker(rank, world, port, cfg, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from ring_cpu_backend import OracleRingBackend
    from sageattention_amd.ring import ring_sageattn
    B, Hq, Hk, N, D, causal, pv, layout, schedule = cfg[:9]
    causal_layout = cfg[9] if len(cfg) > 9 else "contiguous"
    q, k, v = _inputs(B, Hq, Hk, N, D)
    n = N // world
    sl = slice(rank * n, (rank + 1) * n)
    if causal_layout == "zigzag":
        from sageattention_amd.ring import zigzag_split
        ql, kl, vl = (zigzag_split(t, world, rank) for t in (q, k, v))
    else:
        ql, kl, vl = q[:, :, sl], k[:, :, sl], v[:, :, sl]
    if layout == "NHD":
        ql, kl, vl = (t.transpose(1, 2).contiguous() for t in (ql, kl, vl))
    o, lse = ring_sageattn(ql, kl, vl, tensor_layout=layout, is_causal=causal, return_lse=True,
                           backend=OracleRingBackend(pv=pv), schedule=schedule, causal_layout=causal_layout)
    if layout == "NHD":
        o = o.transpose(1, 2)
    torch.save({"o": o.contiguous(), "lse": lse}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,causal,pv,layout,schedule", [
    (2, False, "fp16", "HND", "ring"), (2, True, "fp16", "NHD", "ring"), (3, True, "fp8", "HND", "ring"),
    (2, False, "fp8", "HND", "direct"), (3, True, "fp16", "NHD", "direct"), (4, False, "fp16", "HND", "direct")])
def test_ring_matches_full_attention(tmp_path, world, causal, pv, layout, schedule):
    from oracle import sage_oracle as O
    cfg = (1, 4, 2, 128 * world, 64, causal, pv, layout, schedule)
    mp.spawn(_worker, args=(world, _free_port(), cfg, str(tmp_path)), nprocs=world, join=True)
    B, Hq, Hk, N, D = cfg[:5]
    q, k, v = _inputs(B, Hq, Hk, N, D)
    ref, ref_lse = O.sdpa_fp32(q, k, v, is_causal=causal, return_lse=True)
    outs = [torch.load(os.path.join(tmp_path, f"r{r}.pt"), weights_only=True) for r in range(world)]
    o = torch.cat([x["o"] for x in outs], dim=2).float()
    lse = torch.cat([x["lse"] for x in outs], dim=2)
    # per-shard smoothing/quantisation + LSE merge against exact attention over the whole sequence:
    # the operator's stated tolerance (fp16 PV: 0.08 / 2e-3, fp8 PV: 0.2 / 5e-3), LSE 0.06
    assert (o - ref).abs().max() < (0.08 if pv == "fp16" else 0.2)
    assert calc_diff(o, ref) < (2e-3 if pv == "fp16" else 5e-3)
    assert (lse - ref_lse).abs().max() < 0.06

    # protocol check, exact: the same steps run serially in one process must give bit-identical results
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from ring_cpu_backend import OracleRingBackend
    be = OracleRingBackend(pv=pv)
    n = N // world
    shards = [be.prepare_kv(k[:, :, r * n:(r + 1) * n], v[:, :, r * n:(r + 1) * n]) for r in range(world)]
    for r in range(world):
        qs = be.prepare_q(q[:, :, r * n:(r + 1) * n], D ** -0.5)
        blks = []
        for step in range(world):
            src = (r - step) % world
            if causal and src > r:
                continue
            blks.append(be.block_attn(qs, shards[src], causal and src == r))
        st = be.merge_all(blks)
        assert torch.equal(st[0].to(torch.float16), outs[r]["o"]), f"rank {r} output differs from the serial ring"
        assert torch.equal(st[1], outs[r]["lse"])


@pytest.mark.parametrize("world,pv,layout,schedule", [(2, "fp16", "HND", "ring"), (3, "fp8", "NHD", "direct"),
                                                      (2, "fp16", "HND", "direct")])
def test_zigzag_causal_ring(tmp_path, world, pv, layout, schedule):
    """Causal attention with the load-balanced zigzag layout (rank r owns chunks r and 2P-1-r): equals exact causal
    attention over the whole sequence, and equals the same half-block products replayed serially, bit for bit."""
    from oracle import sage_oracle as O
    from sageattention_amd.ring import zigzag_merge, zigzag_split
    cfg = (1, 4, 2, 256 * world, 64, True, pv, layout, schedule, "zigzag")
    mp.spawn(_worker, args=(world, _free_port(), cfg, str(tmp_path)), nprocs=world, join=True)
    B, Hq, Hk, N, D = cfg[:5]
    q, k, v = _inputs(B, Hq, Hk, N, D)
    ref, ref_lse = O.sdpa_fp32(q, k, v, is_causal=True, return_lse=True)
    outs = [torch.load(os.path.join(tmp_path, f"r{r}.pt"), weights_only=True) for r in range(world)]
    o = zigzag_merge([x["o"] for x in outs]).float()
    lse = zigzag_merge([x["lse"] for x in outs])
    assert (o - ref).abs().max() < (0.08 if pv == "fp16" else 0.2)
    assert calc_diff(o, ref) < (2e-3 if pv == "fp16" else 5e-3)
    assert (lse - ref_lse).abs().max() < 0.06

    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from ring_cpu_backend import OracleRingBackend
    be = OracleRingBackend(pv=pv)
    n, h = N // world, N // world // 2
    shards = [be.prepare_kv(zigzag_split(k, world, r), zigzag_split(v, world, r)) for r in range(world)]
    work = []
    for r in range(world):
        ql = zigzag_split(q, world, r)
        qs = be.prepare_q(ql, D ** -0.5)
        qp = {"lo": be.slice_q(qs, 0, h), "hi": be.slice_q(qs, h, n)}
        st = {"lo": [], "hi": []}
        rng = {"lo": (0, h), "hi": (h, n)}
        blocks = 0.0
        for step in range(world):
            s = (r - step) % world
            pairs = ((("lo", "lo", False), ("hi", "lo", False)) if s < r else (("hi", "all", False),) if s > r
                     else (("lo", "lo", True), ("hi", "lo", False), ("hi", "hi", True)))
            for qa, kb, diag in pairs:
                kv = shards[s] if kb == "all" else be.slice_kv(shards[s], *rng[kb])
                st[qa].append(be.block_attn(qp[qa], kv, diag))
                blocks += (2.0 if kb == "all" else 1.0) * (0.5 if diag else 1.0)
        work.append(blocks)
        mlo, mhi = be.merge_all(st["lo"]), be.merge_all(st["hi"])
        o_ser = torch.cat([mlo[0], mhi[0]], dim=2).to(torch.float16)
        assert torch.equal(o_ser, outs[r]["o"]), f"rank {r} output differs from the serial replay"
        assert torch.equal(torch.cat([mlo[1], mhi[1]], dim=2), outs[r]["lse"])
    assert len(set(work)) == 1, f"zigzag must balance the half-block products across ranks, got {work}"


def _gather_worker(rank, world, port, cfg, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from ring_cpu_backend import OracleGatherBackend
    from sageattention_amd.ring import ring_sageattn
    B, Hq, Hk, N, D, causal, pv, layout, gran = cfg[:9]
    zig = len(cfg) > 9 and cfg[9] == "zigzag"
    q, k, v = _inputs(B, Hq, Hk, N, D)
    n = N // world
    if zig:
        from sageattention_amd.ring import zigzag_split
        ql, kl, vl = (zigzag_split(t, world, rank) for t in (q, k, v))
    else:
        ql, kl, vl = (t[:, :, rank * n:(rank + 1) * n] for t in (q, k, v))
    if layout == "NHD":
        ql, kl, vl = (t.transpose(1, 2).contiguous() for t in (ql, kl, vl))
    extra = cfg[10] if len(cfg) > 10 else {}
    kw = {}
    if "batches" in extra:
        kw["gather_batches"] = extra["batches"]
    if extra.get("late_rank") == rank:   # this rank posts its sends late: every peer's first batch lands late
        import time
        kw["_gather_delay"] = lambda: time.sleep(1.5)
    o, lse = ring_sageattn(ql, kl, vl, tensor_layout=layout, is_causal=causal, return_lse=True, pv=pv, qk_quant_gran=gran,
                           backend=OracleGatherBackend(pv=pv, qk_quant_gran=gran), schedule="gather",
                           causal_layout="zigzag" if zig else "contiguous", **kw)
    if layout == "NHD":
        o = o.transpose(1, 2)
    torch.save({"o": o.contiguous(), "lse": lse}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,causal,pv,layout,gran", [
    (2, False, "fp8", "HND", "per_thread"), (2, True, "fp16", "NHD", "per_thread"), (3, True, "fp8", "HND", "per_warp"),
    (4, False, "fp16", "HND", "per_thread"), (3, False, "fp8", "NHD", "per_thread")])
def test_gather_schedule_matches_the_unsharded_operator(tmp_path, world, causal, pv, layout, gran):
    """schedule="gather": the ranks exchange per-channel statistics, quantize their shards with the WHOLE sequence's
    smoothing mean and V scale, exchange the quantized slots and attend the gathered slots as one sequence.  Checks:
    (1) vs exact fp32 attention over the whole sequence (operator tolerances); (2) vs the UNSHARDED oracle operator on
    the gathered tensors -- same quantized operands by construction, so only the key order (own shard first), the
    two-way merge and the per-rank Q blocks differ: |do| <= 2e-3 (fp16 PV) / 3e-2 (fp8 PV: one e4m3 step of a weight),
    LSE <= 1e-3; (3) protocol, exact: a serial replay of every rank's steps in one process is bit-identical."""
    from oracle import sage_oracle as O
    cfg = (1, 4, 2, 128 * world, 64, causal, pv, layout, gran)
    mp.spawn(_gather_worker, args=(world, _free_port(), cfg, str(tmp_path)), nprocs=world, join=True)
    B, Hq, Hk, N, D = cfg[:5]
    q, k, v = _inputs(B, Hq, Hk, N, D)
    outs = [torch.load(os.path.join(tmp_path, f"r{r}.pt"), weights_only=True) for r in range(world)]
    o = torch.cat([x["o"] for x in outs], dim=2).float()
    lse = torch.cat([x["lse"] for x in outs], dim=2)
    ref, ref_lse = O.sdpa_fp32(q, k, v, is_causal=causal, return_lse=True)
    assert (o - ref).abs().max() < (0.08 if pv == "fp16" else 0.2)
    assert calc_diff(o, ref) < (2e-3 if pv == "fp16" else 5e-3)
    assert (lse - ref_lse).abs().max() < 0.06
    oo, ol = O.sageattn_oracle(q, k, v, is_causal=causal, qk_quant_gran=gran, pv=pv, return_lse=True)
    assert (o - oo.float()).abs().max() < (2e-3 if pv == "fp16" else 3e-2)
    assert (lse - ol).abs().max() < 1e-3

    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from ring_cpu_backend import OracleGatherBackend
    from sageattention_amd.ring import GATHER_BATCHES, _exchange_batches
    n = N // world
    bes = [OracleGatherBackend(pv=pv, qk_quant_gran=gran) for _ in range(world)]
    shards = [(k[:, :, r * n:(r + 1) * n], v[:, :, r * n:(r + 1) * n]) for r in range(world)]
    all_stats = torch.stack([bes[r].stats(*shards[r]) for r in range(world)])
    Gs = [bes[r].setup(all_stats, world, *shards[r]) for r in range(world)]
    for r in range(world):
        for p in range(1, world):
            Gs[r].buf[p] = Gs[(r - p) % world].buf[0]
        qs = bes[r].prepare_q(q[:, :, r * n:(r + 1) * n], D ** -0.5, True)
        parts = [bes[r].attend(qs, Gs[r], 0, 1, causal)]
        nrem = (r if causal else world - 1)
        for lo, hi in _exchange_batches(world, GATHER_BATCHES):   # one launch per exchange batch, in slot order
            hi = min(hi, nrem + 1)
            if hi > lo:
                parts.append(bes[r].attend(qs, Gs[r], lo, hi - lo, False))
        so, sl = bes[r].merge(parts, qs, True)
        assert torch.equal(so, outs[r]["o"]), f"rank {r} output differs from the serial replay"
        assert torch.equal(sl, outs[r]["lse"])


@pytest.mark.parametrize("world,pv,layout,gran", [(2, "fp8", "HND", "per_thread"), (3, "fp16", "NHD", "per_thread"),
                                                  (4, "fp8", "HND", "per_warp")])
def test_gather_schedule_zigzag_causal(tmp_path, world, pv, layout, gran):
    """Causal attention, zigzag layout, on the gather schedule (half-shard slots, five launches per rank): equals exact
    causal attention and the UNSHARDED oracle operator on the whole sequence (same quantized operands), and every rank
    does the same number of half-block products."""
    from oracle import sage_oracle as O
    from sageattention_amd.ring import zigzag_merge
    cfg = (1, 4, 2, 256 * world, 64, True, pv, layout, gran, "zigzag")
    mp.spawn(_gather_worker, args=(world, _free_port(), cfg, str(tmp_path)), nprocs=world, join=True)
    B, Hq, Hk, N, D = cfg[:5]
    q, k, v = _inputs(B, Hq, Hk, N, D)
    outs = [torch.load(os.path.join(tmp_path, f"r{r}.pt"), weights_only=True) for r in range(world)]
    o = zigzag_merge([x["o"] for x in outs]).float()
    lse = zigzag_merge([x["lse"] for x in outs])
    ref, ref_lse = O.sdpa_fp32(q, k, v, is_causal=True, return_lse=True)
    assert (o - ref).abs().max() < (0.08 if pv == "fp16" else 0.2)
    assert calc_diff(o, ref) < (2e-3 if pv == "fp16" else 5e-3)
    assert (lse - ref_lse).abs().max() < 0.06
    oo, ol = O.sageattn_oracle(q, k, v, is_causal=True, qk_quant_gran=gran, pv=pv, return_lse=True)
    assert (o - oo.float()).abs().max() < (2e-3 if pv == "fp16" else 6e-2)   # fp8: rows with a handful of keys, one e4m3 step
    assert (lse - ol).abs().max() < 1e-3


def test_exchange_batches_partition_the_slots():
    from sageattention_amd.ring import _exchange_batches
    for world in range(2, 10):
        for nb in (1, 2, 3, 4, 9):
            b = _exchange_batches(world, nb)
            assert 1 <= len(b) <= min(nb, world - 1)
            assert b[0][0] == 1 and b[-1][1] == world and all(x[1] == y[0] for x, y in zip(b, b[1:]))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1


@pytest.mark.parametrize("causal", [False, True])
def test_gather_schedule_is_independent_of_arrival_order(tmp_path, causal):
    """The exchange of the gather schedule is posted in batches and a rank attends each batch as it lands.  Which launches
    run and the order of the merge are fixed by (world, batches), so one deliberately LATE rank (it posts its sends 1.5 s
    after the others) must not change a single bit of any rank's result; a different number of batches changes only the
    grouping of the launches and of the multi-way merge (<= 2 output ulps, LSE <= 1e-4: fp32 sums in another association)."""
    world = 4
    res = {}
    for tag, extra in (("on_time", {"batches": 3}), ("late", {"batches": 3, "late_rank": 2}), ("one_batch", {"batches": 1})):
        d = tmp_path / tag
        d.mkdir()
        cfg = (1, 4, 2, 128 * world, 64, causal, "fp16", "HND", "per_thread", "contiguous", extra)
        mp.spawn(_gather_worker, args=(world, _free_port(), cfg, str(d)), nprocs=world, join=True)
        res[tag] = [torch.load(os.path.join(d, f"r{r}.pt"), weights_only=True) for r in range(world)]
    for r in range(world):
        assert torch.equal(res["late"][r]["o"], res["on_time"][r]["o"]) and torch.equal(res["late"][r]["lse"], res["on_time"][r]["lse"])
        a, b = res["one_batch"][r]["o"].float(), res["on_time"][r]["o"].float()
        assert ((a - b).abs() <= 2 * 2.0 ** -10 * b.abs().clamp(min=0.25)).all()
        assert (res["one_batch"][r]["lse"] - res["on_time"][r]["lse"]).abs().max() < 1e-4
