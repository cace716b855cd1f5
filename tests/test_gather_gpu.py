"""Sequence-parallel "gather" schedule (sageattention_amd/ring.py: HipGatherBackend) on ONE GPU: every rank's device
steps replayed serially -- statistics, whole-sequence smoothing mean / V scale, quantization straight into the tile-major
exchange slots, the two attention launches over the gathered records, the merge -- with the exchange itself replaced by
device copies.  Checked against the single-launch operator on the gathered sequence (same quantized operands by
construction), against the oracle and against exact fp32 attention.  The communication half runs under gloo on CPU
(tests/test_ring_gloo.py) and with real processes in tests/test_ring_gpu_multiproc.py."""
import pytest
import torch

from conftest import calc_diff

pytestmark = pytest.mark.gpu


def _replay(sa, q, k, v, P, causal, pv, gran, ranks=None, return_lse=True):
    """-> {rank: (o, lse)} for the requested ranks (all by default)."""
    from sageattention_amd.ring import HipGatherBackend
    B, Hq, N, D = q.shape
    n = N // P
    bes = [HipGatherBackend(pv, gran) for _ in range(P)]
    shards = [(k[:, :, r * n:(r + 1) * n], v[:, :, r * n:(r + 1) * n]) for r in range(P)]
    all_stats = torch.stack([bes[r].stats(*shards[r]) for r in range(P)])
    own = [bes[r].setup(all_stats, P, *shards[r]).buf[0].clone() for r in range(P)]   # every rank's slot 0
    out = {}
    for r in (range(P) if ranks is None else ranks):
        S = bes[r].setup(all_stats, P, *shards[r])
        for p in range(1, P):
            S.buf[p].copy_(own[(r - p) % P])                                        # the exchange
        qs = bes[r].prepare_q(q[:, :, r * n:(r + 1) * n], D ** -0.5, return_lse)
        parts = [bes[r].attend(qs, S, 0, 1, causal)]
        nrem = r if causal else P - 1
        if nrem:
            parts.append(bes[r].attend(qs, S, 1, nrem, False))
        out[r] = bes[r].merge(parts, qs, return_lse)
        del S
    return out


def _replay_zigzag(sa, q, k, v, P, pv, gran, ranks=None):
    """The zigzag / gather steps of every requested rank, serially (sageattention_amd.ring._gather_zigzag with the
    exchange replaced by copies).  -> {rank: (o [lo; hi], lse)}"""
    from sageattention_amd.ring import HipGatherBackend, zigzag_split
    B, Hq, N, D = q.shape
    n, h = N // P, N // P // 2
    bes = [HipGatherBackend(pv, gran) for _ in range(P)]
    loc = [tuple(zigzag_split(t, P, r).contiguous() for t in (q, k, v)) for r in range(P)]
    all_stats = torch.stack([bes[r].stats(loc[r][1], loc[r][2]) for r in range(P)])
    lo, hi = [], []
    for r in range(P):
        bes[r].reduce(all_stats, P, N, loc[r][1], loc[r][2])
        a = bes[r].new_slots(1, B, k.shape[1], h, D, k.device); bes[r].quantize(a, loc[r][1][:, :, :h], loc[r][2][:, :, :h])
        b = bes[r].new_slots(1, B, k.shape[1], h, D, k.device); bes[r].quantize(b, loc[r][1][:, :, h:], loc[r][2][:, :, h:])
        lo.append(a.buf[0]); hi.append(b.buf[0])
    out = {}
    for r in (range(P) if ranks is None else ranks):
        be = bes[r]
        LO = be.new_slots(P, B, k.shape[1], h, D, k.device)
        HI = be.new_slots(max(1, P - r), B, k.shape[1], h, D, k.device)
        for p in range(P):
            LO.buf[p].copy_(lo[(r - p) % P])
        for p in range(P - r):
            HI.buf[p].copy_(hi[r + p])
        qs = be.prepare_q(loc[r][0], D ** -0.5, True)
        q_lo, q_hi = be.slice_q(qs, 0, h), be.slice_q(qs, h, n)
        lo_parts = [be.attend(q_lo, LO, 0, 1, True)]
        hi_parts = [be.attend(q_hi, HI, 0, 1, True)]
        if r > 0:
            lo_parts.append(be.attend(q_lo, LO, 1, r, False))
        hi_parts.append(be.attend(q_hi, LO, 0, P, False))
        if r < P - 1:
            hi_parts.append(be.attend(q_hi, HI, 1, P - 1 - r, False))
        (o_lo, l_lo), (o_hi, l_hi) = be.merge(lo_parts, q_lo, True), be.merge(hi_parts, q_hi, True)
        out[r] = (torch.cat([o_lo, o_hi], dim=2), torch.cat([l_lo, l_hi], dim=2))
    return out


@pytest.mark.parametrize("pv,gran,D", [("fp8", "per_thread", 128), ("fp16", "per_thread", 64), ("fp8", "per_warp", 64),
                                       ("fp16", "per_warp", 128)])
@pytest.mark.parametrize("causal", [False, True])
def test_gather_steps_on_one_gpu(pv, gran, D, causal):
    import sageattention_amd as sa
    from oracle import sage_oracle as O
    torch.manual_seed(13)
    B, Hq, Hk, P, n = 2, 8, 4, 4, 256
    N = P * n
    q = torch.randn(B, Hq, N, D, dtype=torch.float16, device="cuda")
    k = (torch.randn(B, Hk, N, D, device="cuda") + 2 * torch.randn(1, Hk, 1, D, device="cuda")).half()
    v = torch.randn(B, Hk, N, D, dtype=torch.float16, device="cuda")
    res = _replay(sa, q, k, v, P, causal, pv, gran)
    o = torch.cat([res[r][0] for r in range(P)], dim=2).float().cpu()
    lse = torch.cat([res[r][1] for r in range(P)], dim=2).cpu()
    # exact attention over the whole sequence: the operator's tolerances
    ref, ref_lse = O.sdpa_fp32(q.cpu(), k.cpu(), v.cpu(), is_causal=causal, return_lse=True)
    assert (o - ref).abs().max() < (0.08 if pv == "fp16" else 0.2)
    assert calc_diff(o, ref) < (2e-3 if pv == "fp16" else 5e-3)
    assert (lse - ref_lse).abs().max() < 0.06
    # the unsharded operator on the same tensors: same quantized K/V by construction (one smoothing mean, one V scale);
    # the key order (own shard first), the merge and the rows' lazy-rescale history differ
    fn = sa.sageattn_qk_int8_pv_fp16_cuda if pv == "fp16" else sa.sageattn_qk_int8_pv_fp8_cuda
    o1, l1 = fn(q, k, v, is_causal=causal, qk_quant_gran=gran, return_lse=True)
    assert (o - o1.float().cpu()).abs().max() < (2e-3 if pv == "fp16" else 3e-2)
    assert (lse - l1.cpu()).abs().max() < 1e-3
    # world size 1 through the public entry point = the unsharded operator, one launch
    o2, l2 = sa.ring_sageattn(q, k, v, is_causal=causal, pv=pv, qk_quant_gran=gran, return_lse=True)
    assert (o2.float().cpu() - o1.float().cpu()).abs().max() < (2e-3 if pv == "fp16" else 3e-2)
    assert (l2.cpu() - l1.cpu()).abs().max() < 1e-3


def test_tile_major_quantizers_are_bit_identical_to_the_dense_ones():
    """The exchange slot written by sage_quant_k_int8_kvtiles / sage_quant_v_fp8_apply holds exactly the bytes of the
    dense quantizers (given the same mean / coefficients), re-arranged tile-major."""
    import sageattention_amd as sa
    from sageattention_amd import _lib as L
    from sageattention_amd.ring import HipGatherBackend
    torch.manual_seed(2)
    B, Hk, n, D = 2, 3, 320, 128
    k = (torch.randn(B, Hk, n, D, device="cuda") + torch.randn(1, Hk, 1, D, device="cuda")).half()
    v = (torch.randn(B, Hk, n, D, device="cuda") * 3).half()
    be = HipGatherBackend("fp8", "per_thread")
    st = be.stats(k, v)
    G = be.setup(st.unsqueeze(0), 1, k, v).buf
    BH, kb, vb, R = be._layout(B, Hk, D)
    T = n // 64
    rec = G[0].view(T, R)
    # statistics and the reduced operands
    assert torch.equal(st[0, :, 2].view(B, Hk, D), k.float().sum(2)) or (st[0, :, 2].view(B, Hk, D) - k.float().sum(2)).abs().max() < 1e-2
    assert torch.equal(st[1, :, 0].view(B, Hk, D), v.float().amax(2)) and torch.equal(st[1, :, 1].view(B, Hk, D), v.float().amin(2))
    km = sa.quant.k_mean(k)
    assert (be.km.float() - km.float()).abs().max() <= 2.0 ** -10 * km.float().abs().max()
    # K: dense quantizer with the SAME mean
    k8, ks, _ = sa.quant._quant(k, "HND", L.GRAN_PER_THREAD, True, 64, 64, 1.0, L.ROUND_TRITON, mean=be.km)
    got_k = rec[:, :kb].view(torch.int8).view(T, B, Hk, 64, D).permute(1, 2, 0, 3, 4).reshape(B, Hk, n, D)
    assert torch.equal(got_k, k8)
    got_s = rec[:, kb + vb:kb + vb + BH * 16].view(torch.float32).view(T, B, Hk, 4).permute(1, 2, 0, 3).reshape(B, Hk, T * 4)
    assert torch.equal(got_s, ks)
    # V: dense FP8 quantizer (its scale is the same max|v|/448 here: one shard)
    v8, vs, _ = sa.quant.per_channel_fp8(v, smooth_v=False)
    assert torch.equal(be.v_scale, vs)
    got_v = rec[:, kb:kb + vb].view(T, B, Hk, D, 64).permute(1, 2, 3, 0, 4).reshape(B, Hk, D, n)
    assert torch.equal(got_v, v8.view(torch.uint8))


@pytest.mark.parametrize("causal", [False, True])
def test_c5_shape_rehearsal_on_one_gpu(causal):
    """BASELINE configs[4] at its true per-rank shape -- (B1, H32, 8192 rows per rank, D128) x 8 ranks = 65536 keys,
    FP8 PV -- with the ranks' steps replayed serially on one GPU.  Ranks 0, 3 and 7 (first, middle, last: no / some /
    all remote shards in the causal case) against the single-launch operator on the gathered sequence; one head of rank
    7 against the oracle."""
    import sageattention_amd as sa
    from oracle import sage_oracle as O
    torch.manual_seed(17)
    B, H, P, n, D = 1, 32, 8, 8192, 128
    N = P * n
    q = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
    k = (torch.randn(B, H, N, D, device="cuda") + 2 * torch.randn(1, H, 1, D, device="cuda")).half()
    v = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
    ranks = (0, 3, 7)
    res = _replay(sa, q, k, v, P, causal, "fp8", "per_thread", ranks=ranks)
    # the unsharded operator on the whole sequence: same quantized operands by construction (one smoothing mean, one V
    # scale); the key order (own shard first), the merge and the lazy-rescale history of a row differ
    o_full, l_full = sa.sageattn_qk_int8_pv_fp8_cuda(q, k, v, is_causal=causal, return_lse=True)
    for r in ranks:
        rows = slice(r * n, (r + 1) * n)
        o, lse = res[r]
        assert torch.isfinite(o).all() and torch.isfinite(lse).all()
        assert (o.float() - o_full[:, :, rows].float()).abs().max() < 3e-2, r
        assert (lse - l_full[:, :, rows]).abs().max() < 1e-3, r
        assert calc_diff(o.float().cpu(), o_full[:, :, rows].float().cpu()) < 2e-3   # outputs are ~1e-2 (64K random keys): fp8 P noise
    # oracle, one head of the last rank (exact fp32 attention of that head on the CPU)
    r, h = 7, 5
    rows = slice(r * n, (r + 1) * n)
    qh, kh, vh = q[:, h:h + 1, rows].cpu(), k[:, h:h + 1].cpu(), v[:, h:h + 1].cpu()
    if causal:
        lim = torch.arange(r * n, (r + 1) * n).view(-1, 1) >= torch.arange(N).view(1, -1)
        s = (qh.float() @ kh.float().transpose(2, 3)) * D ** -0.5
        s = s.masked_fill(~lim, float("-inf"))
        ref = torch.softmax(s, -1) @ vh.float()
        ref_lse = torch.logsumexp(s, -1)
    else:
        ref, ref_lse = O.sdpa_fp32(qh, kh, vh, return_lse=True)
    assert (res[r][0][:, h:h + 1].float().cpu() - ref).abs().max() < 0.1
    assert calc_diff(res[r][0][:, h:h + 1].float().cpu(), ref) < 5e-3
    assert (res[r][1][:, h:h + 1].cpu() - ref_lse).abs().max() < 0.06


@pytest.mark.parametrize("pv,gran,D", [("fp8", "per_thread", 128), ("fp16", "per_warp", 64)])
def test_gather_zigzag_steps_on_one_gpu(pv, gran, D):
    """Causal / zigzag on the gather schedule (half-shard slots, five launches per rank), replayed serially for 3 ranks:
    vs exact causal attention and vs the unsharded operator on the whole sequence (same quantized operands)."""
    import sageattention_amd as sa
    from oracle import sage_oracle as O
    from sageattention_amd.ring import zigzag_merge
    torch.manual_seed(21)
    B, Hq, Hk, P = 1, 4, 2, 3
    N = 512 * P
    q = torch.randn(B, Hq, N, D, dtype=torch.float16, device="cuda")
    k = (torch.randn(B, Hk, N, D, device="cuda") + 2 * torch.randn(1, Hk, 1, D, device="cuda")).half()
    v = torch.randn(B, Hk, N, D, dtype=torch.float16, device="cuda")
    res = _replay_zigzag(sa, q, k, v, P, pv, gran)
    o = zigzag_merge([res[r][0] for r in range(P)]).float().cpu()
    lse = zigzag_merge([res[r][1] for r in range(P)]).cpu()
    ref, ref_lse = O.sdpa_fp32(q.cpu(), k.cpu(), v.cpu(), is_causal=True, return_lse=True)
    assert (o - ref).abs().max() < (0.08 if pv == "fp16" else 0.2)
    assert calc_diff(o, ref) < (2e-3 if pv == "fp16" else 5e-3)
    assert (lse - ref_lse).abs().max() < 0.06
    fn = sa.sageattn_qk_int8_pv_fp16_cuda if pv == "fp16" else sa.sageattn_qk_int8_pv_fp8_cuda
    o1, l1 = fn(q, k, v, is_causal=True, qk_quant_gran=gran, return_lse=True)
    assert (o - o1.float().cpu()).abs().max() < (2e-3 if pv == "fp16" else 6e-2)
    assert (lse - l1.cpu()).abs().max() < 1e-3
    # world size 1 through the public entry point
    o2, l2 = sa.ring_sageattn(q, k, v, is_causal=True, pv=pv, qk_quant_gran=gran, return_lse=True, causal_layout="zigzag")
    assert (o2.float().cpu() - o1.float().cpu()).abs().max() < (2e-3 if pv == "fp16" else 6e-2)
    assert (l2.cpu() - l1.cpu()).abs().max() < 1e-3


def test_c5_shape_zigzag_rehearsal_on_one_gpu():
    """BASELINE configs[4] at its true per-rank shape, causal with the zigzag layout (4096-row half-shards, 8 ranks):
    ranks 0, 3, 7 replayed on one GPU against the unsharded causal operator on the whole 64K sequence."""
    import sageattention_amd as sa
    from sageattention_amd.ring import zigzag_split
    torch.manual_seed(19)
    B, H, P, n, D = 1, 32, 8, 8192, 128
    N = P * n
    q = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
    k = (torch.randn(B, H, N, D, device="cuda") + 2 * torch.randn(1, H, 1, D, device="cuda")).half()
    v = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
    ranks = (0, 3, 7)
    res = _replay_zigzag(sa, q, k, v, P, "fp8", "per_thread", ranks=ranks)
    o_full, l_full = sa.sageattn_qk_int8_pv_fp8_cuda(q, k, v, is_causal=True, return_lse=True)
    for r in ranks:
        o, lse = res[r]
        of, lf = zigzag_split(o_full, P, r), zigzag_split(l_full, P, r)
        assert torch.isfinite(o).all() and torch.isfinite(lse).all()
        assert (o.float() - of.float()).abs().max() < 6e-2, r     # first rows of chunk 0: a handful of keys, one e4m3 step
        assert (lse - lf).abs().max() < 1e-3, r
        assert calc_diff(o.float().cpu(), of.float().cpu()) < 2e-3


def test_gather_bf16_inputs_fp16_pv():
    """bf16 q/k/v through the gather schedule with the FP16-PV operator: the V records stay bf16 in the exchange slots and
    are converted on the fly by the kernel's register-staged path (tile-major strides there as well)."""
    import sageattention_amd as sa
    from oracle import sage_oracle as O
    torch.manual_seed(29)
    B, Hq, Hk, P, n, D = 1, 4, 2, 3, 256, 128
    N = P * n
    q = torch.randn(B, Hq, N, D, device="cuda").bfloat16()
    k = (torch.randn(B, Hk, N, D, device="cuda") + 2 * torch.randn(1, Hk, 1, D, device="cuda")).bfloat16()
    v = torch.randn(B, Hk, N, D, device="cuda").bfloat16()
    for causal in (False, True):
        res = _replay(sa, q, k, v, P, causal, "fp16", "per_thread")
        o = torch.cat([res[r][0] for r in range(P)], dim=2)
        lse = torch.cat([res[r][1] for r in range(P)], dim=2)
        assert o.dtype == torch.bfloat16
        o1, l1 = sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v, is_causal=causal, return_lse=True)
        assert (o.float() - o1.float()).abs().max() < 1.6e-2      # bf16 outputs: 2 ulp at |o| ~ 1
        assert (lse - l1).abs().max() < 1e-3
        ref = O.sdpa_fp32(q.cpu(), k.cpu(), v.cpu(), is_causal=causal)
        assert (o.float().cpu() - ref).abs().max() < 0.08


def test_c5_shape_per_shard_schedules_on_one_gpu():
    """The round-1 per-shard schedules ("direct" / "ring": per-shard smoothing, per-shard outputs, multi-way merge) at the
    true C5 per-rank shape, last rank (8 blocks of 8192 x 8192, FP8 PV), non-causal and causal-zigzag half-blocks, against
    the unsharded operator (different smoothing statistics per shard: operator-level tolerance) and one head of exact
    attention."""
    import sageattention_amd as sa
    from oracle import sage_oracle as O
    from sageattention_amd.ring import HipRingBackend
    torch.manual_seed(23)
    B, H, P, n, D = 1, 32, 8, 8192, 128
    N = P * n
    q = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
    k = (torch.randn(B, H, N, D, device="cuda") + 2 * torch.randn(1, H, 1, D, device="cuda")).half()
    v = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
    be = HipRingBackend("fp8", "per_thread")
    r = P - 1
    qs = be.prepare_q(q[:, :, r * n:(r + 1) * n].contiguous(), D ** -0.5)
    shards = [be.prepare_kv(k[:, :, s * n:(s + 1) * n].contiguous(), v[:, :, s * n:(s + 1) * n].contiguous()) for s in range(P)]
    corrs = be.lse_corrections(qs, shards)
    for causal in (False, True):
        blks = [be.block_attn(qs, shards[s], causal and s == r, corr=corrs[s]) for s in range(P)]
        o, lse = be.merge_all(blks)
        o_full, l_full = sa.sageattn_qk_int8_pv_fp8_cuda(q, k, v, is_causal=causal, return_lse=True)
        rows = slice(r * n, (r + 1) * n)
        assert torch.isfinite(o).all() and torch.isfinite(lse).all()
        assert (o.float() - o_full[:, :, rows].float()).abs().max() < 0.1
        assert (lse - l_full[:, :, rows]).abs().max() < 0.05
        h = 7
        qh, kh, vh = q[:, h:h + 1, rows].cpu(), k[:, h:h + 1].cpu(), v[:, h:h + 1].cpu()
        s_ = (qh.float() @ kh.float().transpose(2, 3)) * D ** -0.5
        if causal:
            s_ = s_.masked_fill(~(torch.arange(r * n, (r + 1) * n).view(-1, 1) >= torch.arange(N).view(1, -1)), float("-inf"))
        ref, ref_lse = torch.softmax(s_, -1) @ vh.float(), torch.logsumexp(s_, -1)
        assert (o[:, h:h + 1].float().cpu() - ref).abs().max() < 0.1
        assert calc_diff(o[:, h:h + 1].float().cpu(), ref) < 5e-3
        assert (lse[:, h:h + 1].cpu() - ref_lse).abs().max() < 0.06
