"""sageattn_varlen (reference core.py:363-477).  Golden vectors: tests/golden/varlen/varlen_d64.npz, produced by the
reference's Triton varlen quantizer + attention kernels (oracle/gen_golden.py: gen_varlen)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, calc_diff


def _load():
    z = np.load(os.path.join(GOLDEN_DIR, "varlen", "varlen_d64.npz"), allow_pickle=False)
    f16 = lambda a: torch.from_numpy(a.view(np.int16).copy()).view(torch.float16)
    return dict(q=f16(z["q"]), k=f16(z["k"]), v=f16(z["v"]), cu=torch.from_numpy(z["cu"].copy()),
                q8=torch.from_numpy(z["q8"].copy()), k8=torch.from_numpy(z["k8"].copy()),
                o=f16(z["o"]), o_causal=f16(z["o_causal"]))


def test_varlen_oracle_vs_reference():
    """CPU: the oracle's restatement against the reference outputs: int8 tensors bit-exact (blocks restart per
    sequence, global K mean), outputs within 2 fp16 ulps when the reference's per-tile fp16 PV rounding is restated."""
    from oracle import sage_oracle as O
    g = _load()
    cu = g["cu"].tolist()
    km = g["k"].float().mean(dim=0, keepdim=True).to(torch.float16)
    for s in range(len(cu) - 1):
        qs_, ks_ = g["q"][cu[s]:cu[s + 1]].unsqueeze(0), g["k"][cu[s]:cu[s + 1]].unsqueeze(0)
        q8, _, k8, _ = O.per_block_int8(qs_, ks_, km.unsqueeze(0), sm_scale=64 ** -0.5, tensor_layout="NHD")
        assert torch.equal(q8[0], g["q8"][cu[s]:cu[s + 1]]) and torch.equal(k8[0], g["k8"][cu[s]:cu[s + 1]])
    for causal, key in ((False, "o"), (True, "o_causal")):
        o = O.sageattn_varlen_oracle(g["q"], g["k"], g["v"], g["cu"], g["cu"], is_causal=causal, flavor="triton")
        ref = g[key].float()
        assert ((o.float() - ref).abs() <= 2 * 2.0 ** -10 * ref.abs().clamp(min=0.25)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("causal", [False, True])
def test_varlen_hip_vs_reference_and_oracle(causal):
    """GPU: the native varlen path (device-side cu_seqlens, no host sync) vs the reference output (|do| <= 4e-3: fp32 PV
    accumulation vs the reference's fp16 tiles) and vs the oracle (<= 4 ulp)."""
    import sageattention_amd as sa
    from oracle import sage_oracle as O
    g = _load()
    mx = int((g["cu"][1:] - g["cu"][:-1]).max())
    o = sa.sageattn_varlen(g["q"].cuda(), g["k"].cuda(), g["v"].cuda(), g["cu"].cuda(), g["cu"].cuda(), mx, mx, is_causal=causal)
    torch.cuda.synchronize()
    ref = g["o_causal" if causal else "o"].float()
    assert o.shape == g["q"].shape
    assert (o.cpu().float() - ref).abs().max() < 4e-3
    assert calc_diff(o.cpu().float(), ref) < 1e-5
    oo = O.sageattn_varlen_oracle(g["q"], g["k"], g["v"], g["cu"], g["cu"], is_causal=causal).float()
    # kernel and oracle round P to fp16 against different (lazy vs exact) row maxima; rows with one or two keys
    # (causal) expose both roundings: 4 ulp
    assert ((o.cpu().float() - oo).abs() <= 4 * 2.0 ** -10 * oo.abs().clamp(min=0.25)).all()


@pytest.mark.gpu
def test_varlen_cross_lengths_bf16_int64():
    """q and k/v with different per-sequence lengths, bf16, int64 cu_seqlens, GQA, head_dim 96 (padded to 128), an
    empty sequence in the middle: vs the oracle and exact per-sequence attention."""
    import sageattention_amd as sa
    from oracle import sage_oracle as O
    torch.manual_seed(7)
    lq, lk = [130, 0, 5, 257], [64, 10, 300, 129]
    cq = torch.tensor([0] + list(np.cumsum(lq)), dtype=torch.int64)
    ck = torch.tensor([0] + list(np.cumsum(lk)), dtype=torch.int64)
    Hq, Hk, D = 6, 2, 96
    q = torch.randn(sum(lq), Hq, D).bfloat16()
    k = (torch.randn(sum(lk), Hk, D) + torch.randn(1, Hk, D)).bfloat16()
    v = torch.randn(sum(lk), Hk, D).bfloat16()
    o = sa.sageattn_varlen(q.cuda(), k.cuda(), v.cuda(), cq.cuda(), ck.cuda(), max(lq), max(lk))
    torch.cuda.synchronize()
    oo = O.sageattn_varlen_oracle(q, k, v, cq, ck)
    assert o.shape == q.shape and o.dtype == torch.bfloat16
    assert (o.cpu().float() - oo.float()).abs().max() < 1.6e-2
    for s in range(4):
        if lq[s] == 0:
            continue
        r = O.sdpa_fp32(q[cq[s]:cq[s + 1]].unsqueeze(0), k[ck[s]:ck[s + 1]].unsqueeze(0), v[ck[s]:ck[s + 1]].unsqueeze(0),
                        tensor_layout="NHD", sm_scale=D ** -0.5)[0]
        assert (o[cq[s]:cq[s + 1]].cpu().float() - r).abs().max() < 0.08


@pytest.mark.gpu
def test_varlen_sequence_without_keys_gives_zero_rows():
    """A sequence that has queries but NO keys: the reference kernel stores zeros for its rows (the tile loop never runs,
    acc = 0, l_i = 1; attn_qk_int8_block_varlen.py:109-121).  The output buffer is pre-filled with NaN bit patterns here
    (through the caching allocator) so that unwritten rows would show."""
    import sageattention_amd as sa
    from oracle import sage_oracle as O
    torch.manual_seed(9)
    lq, lk = [70, 200, 33], [128, 0, 40]
    cq = torch.tensor([0] + list(np.cumsum(lq)), dtype=torch.int32)
    ck = torch.tensor([0] + list(np.cumsum(lk)), dtype=torch.int32)
    H, D = 4, 64
    q = torch.randn(sum(lq), H, D, dtype=torch.float16)
    k = torch.randn(sum(lk), H, D, dtype=torch.float16)
    v = torch.randn(sum(lk), H, D, dtype=torch.float16)
    poison = torch.full((sum(lq), H, D), float("nan"), dtype=torch.float16, device="cuda")
    del poison  # its block is the next allocation of that size: the operator's output
    o = sa.sageattn_varlen(q.cuda(), k.cuda(), v.cuda(), cq.cuda(), ck.cuda(), max(lq), max(lk))
    torch.cuda.synchronize()
    assert torch.isfinite(o).all()
    assert (o[cq[1]:cq[2]] == 0).all()
    oo = O.sageattn_varlen_oracle(q, k, v, cq, ck)
    assert ((o.cpu().float() - oo.float()).abs() <= 4 * 2.0 ** -10 * oo.float().abs().clamp(min=0.25)).all()
