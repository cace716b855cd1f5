"""GPU parity tests proper: the HIP path (through the C ABI) against the golden vectors of the reference and
against the CPU oracle on the same seeded inputs.  Integer outputs bit-exact; floating point within the
tolerance written at each assertion."""
import pytest
import torch

from conftest import LSE2_TOL_FP32_P, LSE2_TOL_ROUNDED_P, LSE2_TOL_TWO_ROUNDED_P, softmax_weights, calc_diff

pytestmark = pytest.mark.gpu

LOG2E = 1.4426950408889634


@pytest.fixture(scope="module")
def sa():
    assert torch.cuda.is_available()
    import sageattention_amd
    return sageattention_amd


def _km3(g):
    km = g.km.squeeze(2) if g.meta["layout"] == "HND" else g.km.squeeze(1)
    return km.contiguous().cuda()


def test_quant_per_block_bit_exact_vs_reference(sa, golden):
    g, m = golden, golden.meta
    q8, qs, k8, ks = sa.quant.per_block_int8(g.q.cuda(), g.k.cuda(), km=_km3(g), sm_scale=m["sm_scale"],
                                             tensor_layout=m["layout"], rounding="triton")
    assert torch.equal(qs.cpu(), g.pb_qs) and torch.equal(ks.cpu(), g.pb_ks)
    assert torch.equal(q8.cpu(), g.pb_q8), (q8.cpu() != g.pb_q8).sum()
    assert torch.equal(k8.cpu(), g.pb_k8), (k8.cpu() != g.pb_k8).sum()


def test_quant_per_thread_bit_exact_vs_reference(sa, golden):
    g, m = golden, golden.meta
    q8, qs, k8, ks = sa.quant.per_thread_int8(g.q.cuda(), g.k.cuda(), km=_km3(g), tensor_layout=m["layout"])
    assert torch.equal(qs.cpu(), g.pt_qs) and torch.equal(ks.cpu(), g.pt_ks)
    assert torch.equal(q8.cpu(), g.pt_q8) and torch.equal(k8.cpu(), g.pt_k8)


def test_quant_per_warp_bit_exact_vs_oracle(sa, golden):
    """CUDA-only quantizer of the reference (fused.cu:685-768): parity unpinned by the reference, bit-exact
    against the oracle's restatement."""
    from oracle import sage_oracle as O
    g, m = golden, golden.meta
    for warpq in (32, 16):
        q8, qs, k8, ks = sa.quant.per_warp_int8(g.q.cuda(), g.k.cuda(), km=_km3(g), WARPQ=warpq, tensor_layout=m["layout"])
        rq8, rqs, rk8, rks = O.per_warp_int8(g.q, g.k, km=g.km, WARPQ=warpq, tensor_layout=m["layout"])
        assert torch.equal(qs.cpu(), rqs) and torch.equal(ks.cpu(), rks)
        assert torch.equal(q8.cpu(), rq8) and torch.equal(k8.cpu(), rk8)


def test_k_mean_vs_reference(sa, golden):
    g, m = golden, golden.meta
    km = sa.quant.k_mean(g.k.cuda(), m["layout"]).cpu()
    ref = (g.km.squeeze(2) if m["layout"] == "HND" else g.km.squeeze(1)).float()
    ulp = 2.0 ** -10 if g.dtype == torch.float16 else 2.0 ** -7  # reduction order: <= 1 ulp of the dtype
    assert ((km.float() - ref).abs() <= ulp * ref.abs().clamp(min=2.0 ** -14)).all()


def _run_attn(sa, g, gran, return_lse=True):
    m = g.meta
    if gran == "per_block":
        q8, qs, k8, ks, code, one = g.pb_q8, g.pb_qs, g.pb_k8, g.pb_ks, 1, True
    else:
        q8, qs, k8, ks, code, one = g.pt_q8, g.pt_qs, g.pt_k8, g.pt_ks, 3, False
    o = torch.empty(g.q.shape, dtype=g.dtype, device="cuda")
    lse = sa._qattn._attn_f16(q8.cuda(), k8.cuda(), g.v.cuda(), o, qs.cuda(), ks.cuda(), None,
                              0 if m["layout"] == "NHD" else 1, m["causal"], code, m["sm_scale"], int(return_lse),
                              logit_mult_is_one=one)
    torch.cuda.synchronize()
    return o.cpu(), lse.cpu()


@pytest.mark.parametrize("gran", ["per_block", "per_thread"])
def test_attention_kernel_vs_reference_and_oracle(sa, golden, gran):
    """Same int8 tensors and scales the reference kernel consumed (fixtures).
    vs the reference's output: |do| <= 4e-3 (fp16) / 2e-2 (bf16) and calc_diff <= 1e-5 -- the reference rounds every
    tile's PV to fp16, this kernel accumulates in fp32 (see tests/test_oracle_golden.py).
    vs the oracle restating THIS kernel's arithmetic: <= 2 ulps of the output dtype; LSE <= 5e-4 (base 2).  At head_dim 64
    the row sums are taken from the fp16-ROUNDED P (row-sum MFMA, like the reference's CUDA kernel, attn_utils.cuh:528-548);
    the LSE bounds there are DERIVED from the rounding (conftest.LSE2_TOL_*): log2(1 + 2^-11) = 7.0e-4 per rounding instance
    -- one against the reference's Triton kernel (fp32 sums), two against the oracle's "hip" flavor (it rounds p, the
    kernel p * 2^d under its lazy rescale) -- plus 1-2e-4 for the folded dequantisation constant."""
    from oracle import sage_oracle as O
    g, m = golden, golden.meta
    if gran == "per_thread" and m["causal"]:
        ref_o = None  # the reference has no valid per-thread causal pairing; oracle only
    else:
        ref_o = g.hnd(g.pb_o if gran == "per_block" else g.pt_o).float()
        ref_lse = g.pb_lse2 if gran == "per_block" else g.pt_lse2
    o, lse2 = _run_attn(sa, g, gran)
    o = g.hnd(o).float()
    if ref_o is not None:
        assert (o - ref_o).abs().max() < (4e-3 if g.dtype == torch.float16 else 2e-2)
        assert calc_diff(o, ref_o) < 1e-5
        # head_dim 128: fp32 sum of unrounded p in both; the kernel's folded dequantisation constant is rounded to
        # <= 0.75 LSB of the integer score (~1e-4 in the base-2 exponent).  head_dim 64: l from the rounded P (docstring)
        lse_tol = LSE2_TOL_ROUNDED_P if g.q.shape[-1] <= 64 else LSE2_TOL_FP32_P
        assert (lse2 - ref_lse).abs().max() < lse_tol
    if gran == "per_block":
        q8, qs, k8, ks, mult = g.pb_q8, g.pb_qs, g.pb_k8, g.pb_ks, 1.0
    else:
        q8, qs, k8, ks, mult = g.pt_q8, g.pt_qs, g.pt_k8, g.pt_ks, m["sm_scale"] * LOG2E
    # (a bf16 V goes in as it is: the kernel multiplies it as bf16 with P rounded to bf16, and so does the "hip" flavor)
    oo, ol = O.attn_tile_loop(g.hnd(q8), g.hnd(k8), g.hnd(g.v), O.expand_q_scale(qs, m["M"], gran),
                              O.expand_k_scale(ks, m["N"], gran), logit_mult=mult, is_causal=bool(m["causal"]),
                              out_dtype=g.dtype, flavor="hip")
    ulp = 2.0 ** -10 if g.dtype == torch.float16 else 2.0 ** -7
    assert ((o - oo.float()).abs() <= 2 * ulp * oo.float().abs().clamp(min=0.25)).all(), (o - oo.float()).abs().max()
    # LSE (base 2): the folded dequantisation constant adds <= 0.75 LSB of the integer score (+ docstring at head_dim 64)
    assert (lse2 - ol).abs().max() < (LSE2_TOL_TWO_ROUNDED_P if g.q.shape[-1] <= 64 else LSE2_TOL_FP32_P)


@pytest.mark.parametrize("gran", ["per_warp", "per_thread"])
@pytest.mark.parametrize("nwaves", [8, 4])
def test_end_to_end_vs_oracle(sa, golden, gran, nwaves):
    """sageattn_qk_int8_pv_fp16_cuda end to end (k mean + quantizers + kernel + LSE fix) vs the oracle's
    restatement of core.py:480-653: <= 2e-3 (fp16) / 1.6e-2 (bf16) absolute (a 1-ulp difference in km can flip
    single int8 values), and vs fp32 attention within the operator's stated tolerance (0.08 / 2e-3 calc_diff)."""
    from oracle import sage_oracle as O
    from sageattention_amd import _lib as L
    g, m = golden, golden.meta
    assert L.lib().sage_set_tuning(0, nwaves) == 0
    try:
        o, lse = sa.sageattn_qk_int8_pv_fp16_cuda(g.q.cuda(), g.k.cuda(), g.v.cuda(), tensor_layout=m["layout"],
                                                  is_causal=bool(m["causal"]), qk_quant_gran=gran, return_lse=True)
        torch.cuda.synchronize()
    finally:
        L.lib().sage_set_tuning(0, 0)
    oo, ol = O.sageattn_oracle(g.q, g.k, g.v, tensor_layout=m["layout"], is_causal=bool(m["causal"]),
                               qk_quant_gran=gran, return_lse=True)
    assert o.shape == g.q.shape and o.dtype == g.dtype and lse.shape == ol.shape
    assert (o.cpu().float() - oo.float()).abs().max() < (2e-3 if g.dtype == torch.float16 else 1.6e-2)
    assert (lse.cpu() - ol).abs().max() < 2e-3
    ref, ref_lse = O.sdpa_fp32(g.q, g.k, g.v, tensor_layout=m["layout"], is_causal=bool(m["causal"]), return_lse=True)
    assert (o.cpu().float() - ref).abs().max() < 0.08
    assert calc_diff(o.cpu().float(), ref) < 2e-3
    assert (lse.cpu() - ref_lse).abs().max() < 0.06


@pytest.mark.parametrize("smooth_v", [False, True])
def test_v_fp8_quantizer_vs_oracle(sa, golden, smooth_v):
    """FP8 V quantizer (CUDA/HIP-only in the reference: parity unpinned by it).  Against the oracle's restatement of
    fused.cu:316-427 with OCP e4m3.  Without smoothing everything is bit-exact: scales (max/min are order independent) and
    every e4m3 byte, exact rounding ties included (bf16 inputs produce them by the hundred; round 1 saw ~0.1 % of the
    bytes differ there and blamed v_cvt_pk_fp8_f32 -- the cause was the ORACLE's reciprocal, python's `scalar / tensor` being
    reciprocal * scalar in torch, 1 ulp off the single division the C source means; oracle._ieee_div).  With smoothing the
    channel mean is a differently ordered fp32 sum, so <= 0.5 % of the bytes may differ, each by exactly one code."""
    from oracle import sage_oracle as O
    g, m = golden, golden.meta
    v8, vs, vm = sa.quant.per_channel_fp8(g.v.cuda(), tensor_layout=m["layout"], smooth_v=smooth_v)
    r8, rs, rm = O.per_channel_fp8(g.v, tensor_layout=m["layout"], smooth_v=smooth_v)
    assert v8.shape == r8.shape and v8.dtype == torch.float8_e4m3fn
    perm = sa.quant.fp8_token_order()
    nblk = v8.shape[-1] // 64
    idx = (torch.arange(nblk).view(-1, 1) * 64 + perm.view(1, -1)).reshape(-1)  # position -> token
    got = v8.cpu().view(torch.uint8)
    want = r8.view(torch.uint8)[..., idx]
    if smooth_v:
        assert torch.allclose(vm.cpu(), rm, rtol=1e-5, atol=1e-6)
        assert torch.allclose(vs.cpu(), rs, rtol=1e-5, atol=0)
    else:
        assert vm is None and torch.equal(vs.cpu(), rs)
    # columns of tokens >= N are never read un-masked: this library writes exact zeros there, the reference writes
    # quantize(0 - mean) (fused.cu:399-424); compare the valid tokens only
    valid = (idx < m["N"])
    assert (got[..., ~valid] == 0).all()
    got, want = got[..., valid], want[..., valid]
    if not smooth_v:
        assert torch.equal(got, want), int((got != want).sum())
    assert (got != want).float().mean() < 5e-3
    assert (got.int() - want.int()).abs().max() <= 1


@pytest.mark.parametrize("layout", ["HND", "NHD"])
def test_v_fp8_quantizer_bit_exact_on_rounding_ties(sa, layout):
    """bf16 V (8-bit mantissas) times 448/amax lands on EXACT e4m3 rounding ties hundreds of times per tensor: every byte must
    equal the oracle's (round-to-nearest-even of one correctly rounded product with one correctly rounded quotient)."""
    from oracle import sage_oracle as O
    g = torch.Generator().manual_seed(7110)
    B, H, N, D = 2, 3, 415, 128
    shape = (B, H, N, D) if layout == "HND" else (B, N, H, D)
    v = torch.randn(shape, generator=g).to(torch.bfloat16)
    v8, vs, _ = sa.quant.per_channel_fp8(v.cuda(), tensor_layout=layout, smooth_v=False)
    r8, rs, _ = O.per_channel_fp8(v, tensor_layout=layout, smooth_v=False)
    perm = sa.quant.fp8_token_order()
    nblk = v8.shape[-1] // 64
    idx = (torch.arange(nblk).view(-1, 1) * 64 + perm.view(1, -1)).reshape(-1)
    valid = idx < N
    got = v8.cpu().view(torch.uint8)[..., valid]
    want = r8.view(torch.uint8)[..., idx][..., valid]
    assert torch.equal(vs.cpu(), rs)
    assert torch.equal(got, want), int((got != want).sum())
    # the data really exercises ties: products exactly half way between two e4m3 codes (spacing 2^(floor(log2|x|) - 3))
    vh = v.float() if layout == "HND" else v.float().transpose(1, 2)
    x = (vh * O._ieee_div(448.0, vh.abs().amax(2, keepdim=True))).abs()
    x = x[x >= 2.0 ** -6]
    u = x / torch.exp2(torch.floor(torch.log2(x)) - 3)
    assert int(((u - torch.floor(u)) == 0.5).sum()) > 100


@pytest.mark.parametrize("gran", ["per_warp", "per_thread"])
@pytest.mark.parametrize("smooth_v", [False, True])
def test_fp8_end_to_end_vs_oracle(sa, golden, gran, smooth_v):
    """sageattn_qk_int8_pv_fp8_cuda vs the oracle's restatement of core.py:656-905 (parity unpinned by the reference:
    its fp8 path is CUDA/ROCm-only).  e4m3 P has 3 mantissa bits and the kernel rounds it against a lazily updated
    max (different mantissa alignment than the oracle, two e4m3 roundings apart): |do| <= 0.06, calc_diff <= 1e-3 vs the oracle; vs fp32
    attention the operator tolerance of the fp8 path, 0.2 / 5e-3 (tests/test_oracle_golden.py)."""
    from oracle import sage_oracle as O
    g, m = golden, golden.meta
    kw = dict(tensor_layout=m["layout"], is_causal=bool(m["causal"]), qk_quant_gran=gran, smooth_v=smooth_v)
    o, lse = sa.sageattn_qk_int8_pv_fp8_cuda(g.q.cuda(), g.k.cuda(), g.v.cuda(), pv_accum_dtype="fp32", return_lse=True, **kw)
    torch.cuda.synchronize()
    oo, ol = O.sageattn_oracle(g.q, g.k, g.v, pv="fp8", return_lse=True, **kw)
    assert o.shape == g.q.shape and o.dtype == g.dtype
    assert (o.cpu().float() - oo.float()).abs().max() < 0.06
    assert calc_diff(o.cpu().float(), oo.float()) < 1e-3
    assert (lse.cpu() - ol).abs().max() < 2e-3  # l is the fp32 sum of unrounded p in both
    ref, ref_lse = O.sdpa_fp32(g.q, g.k, g.v, tensor_layout=m["layout"], is_causal=bool(m["causal"]), return_lse=True)
    assert (o.cpu().float() - ref).abs().max() < 0.2
    assert calc_diff(o.cpu().float(), ref) < 5e-3
    assert (lse.cpu() - ref_lse).abs().max() < 0.06


def test_api_surface_and_errors(sa):
    q = torch.randn(1, 2, 64, 64, dtype=torch.float16, device="cuda")
    # SDPA-style kwargs are accepted and ignored (modify_wan.py:63-72)
    o = sa.sageattn(q, q, q, attn_mask=None, dropout_p=0.0, is_causal=False)
    assert o.shape == q.shape
    with pytest.raises(AssertionError):
        sa.sageattn(q.float(), q.float(), q.float())
    with pytest.raises(ValueError):
        big = torch.randn(1, 1, 16, 192, dtype=torch.float16, device="cuda")
        sa.sageattn(big, big, big)
    with pytest.raises(ValueError):
        sa.sageattn_qk_int8_pv_fp16_cuda(q, q, q, pv_accum_dtype="int4")
    # head_dim padding (core.py:592-601): D=80 -> 128, output sliced back
    x = torch.randn(1, 2, 96, 80, dtype=torch.float16, device="cuda")
    o = sa.sageattn(x, x, x)
    ref = torch.nn.functional.scaled_dot_product_attention(x.float(), x.float(), x.float())
    # sageattn dispatches to the FP8-PV path (as the fork, core.py:144): self-attention is sharply peaked, a single
    # e4m3-rounded weight carries the row -> the fp8 operator tolerance 0.2 applies
    assert o.shape == x.shape and (o.float() - ref).abs().max() < 0.2
    o = sa.sageattn_qk_int8_pv_fp16_cuda(x, x, x)
    assert o.shape == x.shape and (o.float() - ref).abs().max() < 0.05
    # smooth_v path of pv_accum_dtype="fp16" (core.py:636-638)
    v = x + 3.0
    o = sa.sageattn_qk_int8_pv_fp16_cuda(x, x, v, pv_accum_dtype="fp16", smooth_v=True)
    ref = torch.nn.functional.scaled_dot_product_attention(x.float(), x.float(), v.float())
    assert (o.float() - ref).abs().max() < 0.05


def test_non_positive_sm_scale_is_rejected(sa):
    q = torch.randn(1, 2, 128, 64, dtype=torch.float16, device="cuda")
    for bad in (0.0, -0.125, float("inf"), float("nan")):
        with pytest.raises(ValueError):
            sa.sageattn_qk_int8_pv_fp16_cuda(q, q, q, sm_scale=bad)
        with pytest.raises(ValueError):
            sa.sageattn_qk_int8_pv_fp8_cuda(q, q, q, sm_scale=bad)


def test_full_size_properties(sa):
    """BASELINE full sizes through size-independent properties: (1) V = 1 => O = 1 exactly up to fp16 rounding
    (softmax rows sum to one, normaliser consistent with P); (2) a slice of heads equals the oracle; both at
    C2 (4,32,2048,64) and, for (1), C3 (4,32,8192,128)."""
    from oracle import sage_oracle as O
    torch.manual_seed(1)
    for (B, H, N, D) in [(4, 32, 2048, 64), (4, 32, 8192, 128)]:
        q = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
        k = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
        ones = torch.ones(B, H, N, D, dtype=torch.float16, device="cuda")
        o = sa.sageattn_qk_int8_pv_fp16_cuda(q, k, ones)
        assert (o.float() - 1).abs().max() < 2e-3
        if N == 2048:
            v = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
            o, lse = sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v, return_lse=True)
            sl = (slice(3, 4), slice(30, 32))
            oo, ol = O.sageattn_oracle(q[sl].cpu(), k[sl].cpu(), v[sl].cpu(), qk_quant_gran="per_thread", return_lse=True)
            assert (o[sl].cpu().float() - oo.float()).abs().max() < 2e-3
            assert (lse[sl].cpu() - ol).abs().max() < 2e-3
            # causal at full size vs oracle slice
            o = sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v, is_causal=True)
            oo = O.sageattn_oracle(q[sl].cpu(), k[sl].cpu(), v[sl].cpu(), qk_quant_gran="per_thread", is_causal=True)
            assert (o[sl].cpu().float() - oo.float()).abs().max() < 4e-3
            # fp8 PV at full size, causal, vs the oracle slice
            o = sa.sageattn_qk_int8_pv_fp8_cuda(q, k, v, is_causal=True)
            oo = O.sageattn_oracle(q[sl].cpu(), k[sl].cpu(), v[sl].cpu(), qk_quant_gran="per_thread", is_causal=True, pv="fp8")
            assert (o[sl].cpu().float() - oo.float()).abs().max() < 0.04
        o8 = sa.sageattn_qk_int8_pv_fp8_cuda(q, k, ones)
        assert (o8.float() - 1).abs().max() < 0.08  # e4m3 P: the fp32 normaliser is not the sum of the rounded P


@pytest.mark.parametrize("cfg", [
    # (B, H, N, D, causal, pv)
    (4, 32, 2048, 64, True, "fp16"),   # three waves per SIMD: the configuration in which the round-1 prologue
    (4, 32, 2048, 64, True, "fp8"),    # race (K buffer 0 re-filled too early, see sage_attn.hip) showed
    (4, 32, 2048, 64, False, "fp8"),
    (2, 16, 4096, 128, True, "fp16"),
    (2, 16, 4096, 128, True, "fp8"),
])
def test_run_to_run_determinism(sa, cfg):
    """The operator is a pure function of its inputs: 40 launches on the same tensors, with the whole chip busy (so that
    waves queue on the matrix pipe), must give bit-identical outputs and LSE."""
    B, H, N, D, causal, pv = cfg
    torch.manual_seed(23)
    q = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
    fn = sa.sageattn_qk_int8_pv_fp16_cuda if pv == "fp16" else sa.sageattn_qk_int8_pv_fp8_cuda
    o0, l0 = fn(q, k, v, is_causal=causal, return_lse=True)
    for _ in range(40):
        o, l = fn(q, k, v, is_causal=causal, return_lse=True)
        assert torch.equal(o, o0) and torch.equal(l, l0)


@pytest.mark.parametrize("cfg", [(4, 32, 2048, 64, True, "fp16"), (4, 32, 2048, 64, True, "fp8"), (2, 32, 4096, 128, True, "fp16"),
                                 (2, 32, 4096, 128, False, "fp8"), (4, 32, 2048, 64, False, "fp16")])
def test_workgroup_geometry_does_not_change_results(sa, cfg):
    """A wave computes its 32 query rows from the same tiles in the same order whatever the workgroup size, so the
    4-wave and the 8-wave builds of a kernel must agree bit for bit over the WHOLE output -- a cross-check between two
    differently scheduled instantiations (registers, occupancy, barriers) that needs no reference."""
    from sageattention_amd import _lib as L
    B, H, N, D, causal, pv = cfg
    torch.manual_seed(29)
    q = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
    fn = sa.sageattn_qk_int8_pv_fp16_cuda if pv == "fp16" else sa.sageattn_qk_int8_pv_fp8_cuda
    lib = L.lib()
    outs = []
    try:
        for nw in (4, 8):
            assert lib.sage_set_tuning(0, nw) == 0
            outs.append(fn(q, k, v, is_causal=causal, return_lse=True))
    finally:
        lib.sage_set_tuning(0, 0)
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])


def test_sageattn_dispatch_rule(sa, monkeypatch):
    """``sageattn`` picks FP8 PV from a few thousand keys per query row upwards and FP16 PV below (measured crossover,
    profiles/r01d_sweep_end_to_end.md); SAGEATTN_DISPATCH pins it.  Checked through bit-equality with the named
    operators."""
    torch.manual_seed(3)
    for (N, D, causal, want) in [(1024, 128, False, "fp16"), (2048, 128, False, "fp8"), (2048, 128, True, "fp16"),
                                 (2048, 64, False, "fp16"), (4096, 64, False, "fp8")]:
        q = torch.randn(1, 2, N, D, dtype=torch.float16, device="cuda")
        k = torch.randn(1, 2, N, D, dtype=torch.float16, device="cuda")
        v = torch.randn(1, 2, N, D, dtype=torch.float16, device="cuda")
        assert sa.core.dispatch_pv(q, k, "HND", causal) == want
        named = sa.sageattn_qk_int8_pv_fp8_cuda if want == "fp8" else sa.sageattn_qk_int8_pv_fp16_cuda
        assert torch.equal(sa.sageattn(q, k, v, is_causal=causal), named(q, k, v, is_causal=causal))
    monkeypatch.setenv("SAGEATTN_DISPATCH", "fp8")
    assert torch.equal(sa.sageattn(q[:, :, :256], k[:, :, :256], v[:, :, :256]),
                       sa.sageattn_qk_int8_pv_fp8_cuda(q[:, :, :256], k[:, :, :256], v[:, :, :256]))
    monkeypatch.setenv("SAGEATTN_DISPATCH", "bogus")
    with pytest.raises(ValueError):
        sa.sageattn(q, k, v)


@pytest.mark.parametrize("cfg", [(2, 8, 2048, 64, False, 0), (2, 8, 1000, 64, True, 0), (1, 4, 2115, 64, False, 8), (2, 4, 777, 128, True, 0),
                                 (1, 8, 1024, 128, False, 4), (1, 2, 63, 64, False, 0), (2, 16, 2048, 64, True, 0)])
def test_bf16_v_is_multiplied_as_bf16(sa, cfg):
    """core.py:633 converts a bf16 V to fp16 before the kernel; this library keeps it ("bf16-native V"): P is rounded
    to bf16 and P.V runs on the bf16 MFMA with fp32 accumulation.  Same INT8 operands; against the oracle restating exactly that
    (<= 2 ulps of the bf16 output + the slack explained below, LSE as for fp16 V -- l is the fp32 sum of the unrounded p), bit-stable over 10 launches, and
    within bf16 rounding of the same call with ``v.to(float16)`` (the reference's form)."""
    from sageattention_amd import _lib as L, core
    from oracle import sage_oracle as O
    B, H, N, D, causal, nw = cfg
    torch.manual_seed(41)
    q, k = (torch.randn(B, H, N, D, dtype=torch.bfloat16, device="cuda") for _ in range(2))
    v = (torch.randn(B, H, N, D, device="cuda") * 3).to(torch.bfloat16)
    km = sa.quant.k_mean(k)
    q8, qs, k8, ks, _ = core._quant_qk(q, k, km, "HND", "per_thread", D ** -0.5, 32, False, H, H)
    v16 = v.to(torch.float16)
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream

    def run(vt, vdt):
        o = torch.empty(B, H, N, D, dtype=torch.bfloat16, device="cuda")
        lse = torch.empty(B, H, N, dtype=torch.float32, device="cuda")
        r = lib.sage_attn_qk_int8_pv_f16(L.desc(q8, "HND"), L.desc(k8, "HND"), L.desc(vt, "HND"), vdt, L.desc(o, "HND"), L.SAGE_BF16,
                                         qs.data_ptr(), ks.data_ptr(), None, lse.data_ptr(), B, H, H, N, N, D, int(causal), 3, 128,
                                         32, D ** -0.5, 0, st)
        assert r == 0, r
        return o, lse
    lib.sage_set_tuning(0, nw)
    try:
        o16, l16 = run(v16, L.SAGE_F16)
        o, l = run(v, L.SAGE_BF16)
        for _ in range(10):
            o2, l2 = run(v, L.SAGE_BF16)
            assert torch.equal(o, o2) and torch.equal(l, l2), cfg
    finally:
        lib.sage_set_tuning(0, 0)
    hs = slice(0, min(H, 2))  # two heads on the CPU
    oo, ol = O.attn_tile_loop(q8[:, hs].cpu(), k8[:, hs].cpu(), v[:, hs].cpu(), O.expand_q_scale(qs[:, hs].cpu(), N, "per_thread"),
                              O.expand_k_scale(ks[:, hs].cpu(), N, "per_thread"), logit_mult=D ** -0.5 * LOG2E, is_causal=causal,
                              out_dtype=torch.bfloat16, flavor="hip")
    of, oof = o[:, hs].cpu().float(), oo.float()
    # DERIVED bound per output element.  P is rounded to bf16 (RNE, 8 significant bits: within 2^-9 relative of p) at the
    # kernel's lazily rescaled magnitude p * 2^d and at the oracle's exact one -- two rounding instances, so the two P differ by
    # at most 2 * 2^-9 * p and the numerators sum(P v) by at most 2^-8 * sum(p |v|); both divide by the same fp32 l.  With the
    # normalised weights W of the quantized operands that is 2^-8 * (W @ |v|) per element, on top of the 2 output ulps.
    W = softmax_weights(q8[:, hs].cpu(), k8[:, hs].cpu(), O.expand_q_scale(qs[:, hs].cpu(), N, "per_thread"),
                        O.expand_k_scale(ks[:, hs].cpu(), N, "per_thread"), D ** -0.5 * LOG2E, causal)
    wv = W @ v[:, hs].cpu().float().abs()
    assert ((of - oof).abs() <= 2 * 2.0 ** -7 * oof.abs().clamp(min=0.25) + 2.0 ** -8 * wv).all(), (of - oof).abs().max()
    assert (l[:, hs].cpu() - ol).abs().max() < 5e-4
    # the reference's form (V converted to fp16 -- exact for these magnitudes -- and P rounded to fp16, within 2^-12 of p): the two
    # numerators differ by at most (2^-9 + 2^-12) * sum(p |v|), plus the bf16 rounding of the two outputs
    d16 = (o.float() - o16.float()).abs()[:, hs].cpu()
    o16h = o16.float()[:, hs].cpu()
    assert (d16 <= 2.0 ** -6 * o16h.abs().clamp(min=0.5) + (2.0 ** -9 + 2.0 ** -12) * wv).all(), d16.max()
    assert calc_diff(o.float().cpu(), o16.float().cpu()) < 2e-5
    # the row sums are taken from the unrounded p in both: at head_dim 128 the LSE is the same number, at head_dim 64 the fp16
    # path sums the fp16-rounded P (one rounding instance: conftest.LSE2_TOL_ROUNDED_P)
    assert (l - l16).abs().max() < (LSE2_TOL_ROUNDED_P if D == 64 else 1e-6)


@pytest.mark.parametrize("cfg", [(2, 4, 333, 64, False, "fp16"), (1, 4, 520, 128, True, "fp16"), (2, 2, 257, 128, False, "fp8"),
                                 (1, 4, 192, 64, True, "fp8")])
def test_output_rows_that_are_only_8_byte_aligned(sa, cfg):
    """The epilogue stores 16 bytes per lane when every output row starts on a 16-byte boundary (the lane halves exchange
    4-channel runs first) and keeps 8-byte stores otherwise: an output view whose row stride is a multiple of 4 but not of 8
    elements must receive exactly the bytes of the contiguous call, and nothing outside its rows."""
    from sageattention_amd import _lib as L, core
    B, H, N, D, causal, pv = cfg
    torch.manual_seed(5)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.float16, device="cuda") for _ in range(3))
    km = sa.quant.k_mean(k)
    q8, qs, k8, ks, _ = core._quant_qk(q, k, km, "HND", "per_thread", D ** -0.5, 32, False, H, H)
    lib, st = L.lib(), torch.cuda.current_stream().cuda_stream
    if pv == "fp8":
        v8, vsc, _ = sa.quant.per_channel_fp8(v, smooth_v=False)
        vd = L.SageTensor(v8.data_ptr(), v8.stride(0), v8.stride(1), v8.stride(2))

    def run(o):
        od = L.SageTensor(o.data_ptr(), o.stride(0), o.stride(1), o.stride(2))
        if pv == "fp8":
            r = lib.sage_attn_qk_int8_pv_f8(L.desc(q8, "HND"), L.desc(k8, "HND"), vd, od, L.SAGE_F16, qs.data_ptr(), ks.data_ptr(),
                                            vsc.data_ptr(), None, None, B, H, H, N, N, D, int(causal), 3, 128, 32, D ** -0.5, 0, st)
        else:
            r = lib.sage_attn_qk_int8_pv_f16(L.desc(q8, "HND"), L.desc(k8, "HND"), L.desc(v, "HND"), L.SAGE_F16, od, L.SAGE_F16,
                                             qs.data_ptr(), ks.data_ptr(), None, None, B, H, H, N, N, D, int(causal), 3, 128, 32,
                                             D ** -0.5, 0, st)
        assert r == 0, r
    o_ref = torch.empty(B, H, N, D, dtype=torch.float16, device="cuda")
    run(o_ref)
    wide = torch.full((B, H, N, D + 4), 7.0, dtype=torch.float16, device="cuda")  # row stride D + 4: 8-byte aligned rows
    run(wide[..., :D])
    torch.cuda.synchronize()
    assert torch.equal(wide[..., :D], o_ref)
    assert (wide[..., D:] == 7.0).all()


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("cfg", [(4, 32, 2048, 64, True, "fp16"), (4, 32, 2048, 64, True, "fp8"), (4, 32, 2048, 64, False, "fp8")])
def test_determinism_under_perturbed_timing(sa, cfg, fused):
    """As test_run_to_run_determinism, for the three-waves-per-SIMD instantiations, with a DIFFERENT heavy kernel
    between the checked launches (clock, cache and co-residency state change from launch to launch), through the
    fused-Q entry points (the default) and through quantizer + kernel.  This is the regime in which the round-1
    kernel returned a wrong 32-row wave about once in 150-300 launches: K buffer 0 was re-filled with K(2) before
    every wave had read K(0) (sage_attn.hip, barrier after the prologue S(0); profiles/r02_race_evidence.md;
    tools/stress_determinism.py is the long form: 0 in 3000 after the fix, 8 in 1200 without it)."""
    B, H, N, D, causal, pv = cfg
    torch.manual_seed(31)
    big = [torch.randn(4, 32, 4096, 128, dtype=torch.float16, device="cuda") for _ in range(3)]
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.float16, device="cuda") for _ in range(3))
    fn = sa.sageattn_qk_int8_pv_fp16_cuda if pv == "fp16" else sa.sageattn_qk_int8_pv_fp8_cuda
    keep = sa.core.FUSE_Q_QUANT
    try:
        sa.core.FUSE_Q_QUANT = fused
        o0, l0 = fn(q, k, v, is_causal=causal, return_lse=True)
        for it in range(150):
            if it % 3 == 0:
                sa.sageattn_qk_int8_pv_fp8_cuda(*big, is_causal=(it % 2 == 0))
            elif it % 3 == 1:
                torch.mm(big[0].view(-1, 128)[:4096].float(), big[1].view(-1, 128)[:4096].float().t())
            o, l = fn(q, k, v, is_causal=causal, return_lse=True)
            assert torch.equal(o, o0) and torch.equal(l, l0), f"launch {it} differs"
    finally:
        sa.core.FUSE_Q_QUANT = keep


def test_full_size_c4_fp8_causal_properties(sa):
    """BASELINE configs[3] = (4,32,16384,128), INT8 QK^T + FP8 PV, causal, at full size, through properties that do
    not need an O(N^2) reference for the whole tensor: (1) V = 1 => O = 1; (2) causality: with K smoothing off, the
    first half of the output rows does not change by one bit when the second half of K is replaced (keys in the
    future of every one of those rows; V is left alone because its per-channel FP8 scale spans all tokens);
    (3) one head equals the oracle."""
    from oracle import sage_oracle as O
    torch.manual_seed(4)
    B, H, N, D = 4, 32, 16384, 128
    q = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
    ones = torch.ones_like(v)
    o1 = sa.sageattn(q, k, ones, is_causal=True)                     # the dispatcher: FP8 PV
    assert (o1.float() - 1).abs().max() < 0.08
    del ones, o1
    oa = sa.sageattn_qk_int8_pv_fp8_cuda(q, k, v, is_causal=True, smooth_k=False)
    k2 = k.clone()
    k2[:, :, N // 2:] = torch.randn(B, H, N // 2, D, dtype=torch.float16, device="cuda") * 3
    ob = sa.sageattn_qk_int8_pv_fp8_cuda(q, k2, v, is_causal=True, smooth_k=False)
    assert torch.equal(oa[:, :, :N // 2], ob[:, :, :N // 2])
    assert not torch.equal(oa[:, :, N // 2:], ob[:, :, N // 2:])
    del k2, ob
    o, lse = sa.sageattn_qk_int8_pv_fp8_cuda(q, k, v, is_causal=True, return_lse=True)
    sl = (slice(2, 3), slice(17, 18))
    oo, ol = O.sageattn_oracle(q[sl].cpu(), k[sl].cpu(), v[sl].cpu(), qk_quant_gran="per_thread", is_causal=True, pv="fp8",
                               return_lse=True)
    assert (o[sl].cpu().float() - oo.float()).abs().max() < 0.06
    assert calc_diff(o[sl].cpu().float(), oo.float()) < 1e-3
    assert (lse[sl].cpu() - ol).abs().max() < 2e-3


@pytest.mark.parametrize("count,dt,D", [(1, torch.float16, 64), (3, torch.float16, 128), (8, torch.bfloat16, 128), (16, torch.float16, 64)])
def test_multiway_merge_vs_torch(sa, count, dt, D):
    """sage_merge_attn_states_multi: o = sum_i o_i exp(lse_i - lse), lse = logsumexp_i lse_i, in one pass; blocks with
    lse = -inf (nothing attended) weigh zero whatever their o holds; rows where every block is empty give (0, -inf)."""
    import ctypes
    from sageattention_amd import _lib as L
    torch.manual_seed(count)
    rows = 1000
    os_ = [torch.randn(rows, D, device="cuda").to(dt) for _ in range(count)]
    ls = [torch.randn(rows, device="cuda") * 3 for _ in range(count)]
    if count > 1:
        ls[1][::7] = float("-inf")
        os_[1][::7] = float("nan")          # must not leak: weight 0
        for l in ls:
            l[5] = float("-inf")            # a row nobody attended
    o = torch.empty(rows, D, dtype=dt, device="cuda")
    lse = torch.empty(rows, dtype=torch.float32, device="cuda")
    op = (ctypes.c_void_p * count)(*[t.data_ptr() for t in os_])
    lp = (ctypes.c_void_p * count)(*[t.data_ptr() for t in ls])
    L.check(L.lib().sage_merge_attn_states_multi(op, lp, count, L.dtype_code(dt), o.data_ptr(), lse.data_ptr(), rows, D,
                                                 L.stream_ptr(o.device)), "merge")
    lst = torch.stack(ls)
    ref_l = torch.logsumexp(lst, dim=0)
    w = torch.exp(lst - ref_l).nan_to_num(0.0)
    ref_o = sum(torch.where(w[i].unsqueeze(-1) > 0, os_[i].float() * w[i].unsqueeze(-1), torch.zeros(1, device="cuda"))
                for i in range(count))
    assert torch.equal(torch.isinf(lse), torch.isinf(ref_l))
    fin = ~torch.isinf(ref_l)
    assert (lse[fin] - ref_l[fin]).abs().max() < 1e-5
    assert (o.float() - ref_o).abs().max() < (2e-3 if dt == torch.float16 else 2e-2)
    assert L.lib().sage_merge_attn_states_multi(op, lp, 17, 0, o.data_ptr(), lse.data_ptr(), rows, D, None) == -1


@pytest.mark.parametrize("pv", ["fp16", "fp8"])
@pytest.mark.parametrize("causal", [False, True])
def test_ring_steps_on_one_gpu(sa, pv, causal):
    """The device half of ring attention (sageattention_amd/ring.py HipRingBackend: per-shard quantisation with the
    local K mean, block attention with LSE, sage_merge_attn_states) run serially for 4 sequence shards on one GPU,
    against exact fp32 attention over the whole sequence and against the single-device operator.  The communication
    half is covered on CPU by tests/test_ring_gloo.py."""
    from oracle import sage_oracle as O
    from sageattention_amd.ring import HipRingBackend, ring_sageattn
    torch.manual_seed(3)
    B, Hq, Hk, N, D, P = 1, 8, 4, 1024, 128, 4
    q = torch.randn(B, Hq, N, D, dtype=torch.float16, device="cuda")
    k = (torch.randn(B, Hk, N, D, device="cuda") + 2 * torch.randn(1, Hk, 1, D, device="cuda")).half()
    v = torch.randn(B, Hk, N, D, dtype=torch.float16, device="cuda")
    be = HipRingBackend(pv=pv)
    n = N // P
    shards = [be.prepare_kv(k[:, :, r * n:(r + 1) * n], v[:, :, r * n:(r + 1) * n]) for r in range(P)]
    outs, lses = [], []
    for r in range(P):
        qr = q[:, :, r * n:(r + 1) * n]
        qs = be.prepare_q(qr, D ** -0.5)
        blks = []
        for step in range(P):
            src = (r - step) % P
            if causal and src > r:
                continue
            blks.append(be.block_attn(qs, shards[src], causal and src == r))
        st = be.merge_all(blks)
        outs.append(st[0].float()); lses.append(st[1])
    o = torch.cat(outs, dim=2).cpu()
    lse = torch.cat(lses, dim=2).cpu()
    ref, ref_lse = O.sdpa_fp32(q.cpu(), k.cpu(), v.cpu(), is_causal=causal, return_lse=True)
    assert (o - ref).abs().max() < (0.08 if pv == "fp16" else 0.2)
    assert calc_diff(o, ref) < (2e-3 if pv == "fp16" else 5e-3)
    assert (lse - ref_lse).abs().max() < 0.06
    # world_size 1 through the public entry point == one block
    o1, l1 = ring_sageattn(q, k, v, is_causal=causal, pv=pv, return_lse=True)
    assert (o1.cpu().float() - ref).abs().max() < (0.08 if pv == "fp16" else 0.2)
    assert (l1.cpu() - ref_lse).abs().max() < 0.06


@pytest.mark.parametrize("pv,gran", [("fp16", "per_thread"), ("fp8", "per_warp")])
def test_zigzag_half_blocks_on_one_gpu(sa, pv, gran):
    """Causal ring with the zigzag layout (rank r owns chunks r and 2P-1-r), device half: whole-shard quantisation,
    half-block products on ROW SLICES of the quantized tensors and scale vectors (strided q8/k8/v views through the
    C ABI), LSE merge -- replayed serially for 3 ranks on one GPU against exact causal attention, and against the
    public entry point at world size 1."""
    from oracle import sage_oracle as O
    from sageattention_amd.ring import HipRingBackend, ring_sageattn, zigzag_merge, zigzag_split
    torch.manual_seed(5)
    B, Hq, Hk, D, P = 1, 4, 2, 128, 3
    N = 256 * P
    q = torch.randn(B, Hq, N, D, dtype=torch.float16, device="cuda")
    k = (torch.randn(B, Hk, N, D, device="cuda") + 2 * torch.randn(1, Hk, 1, D, device="cuda")).half()
    v = torch.randn(B, Hk, N, D, dtype=torch.float16, device="cuda")
    be = HipRingBackend(pv=pv, qk_quant_gran=gran)
    n, h = N // P, N // P // 2
    shards = [be.prepare_kv(zigzag_split(k, P, r), zigzag_split(v, P, r)) for r in range(P)]
    rng = {"lo": (0, h), "hi": (h, n)}
    outs, lses = [], []
    for r in range(P):
        ql = zigzag_split(q, P, r)
        qs = be.prepare_q(ql, D ** -0.5)
        qp = {"lo": be.slice_q(qs, 0, h), "hi": be.slice_q(qs, h, n)}
        st = {"lo": [], "hi": []}
        for step in range(P):
            s = (r - step) % P
            pairs = ((("lo", "lo", False), ("hi", "lo", False)) if s < r else (("hi", "all", False),) if s > r
                     else (("lo", "lo", True), ("hi", "lo", False), ("hi", "hi", True)))
            for qa, kb, diag in pairs:
                kv = shards[s] if kb == "all" else be.slice_kv(shards[s], *rng[kb])
                st[qa].append(be.block_attn(qp[qa], kv, diag))
        mlo, mhi = be.merge_all(st["lo"]), be.merge_all(st["hi"])
        outs.append(torch.cat([mlo[0], mhi[0]], dim=2).float())
        lses.append(torch.cat([mlo[1], mhi[1]], dim=2))
    o = zigzag_merge(outs).cpu()
    lse = zigzag_merge(lses).cpu()
    ref, ref_lse = O.sdpa_fp32(q.cpu(), k.cpu(), v.cpu(), is_causal=True, return_lse=True)
    assert (o - ref).abs().max() < (0.08 if pv == "fp16" else 0.2)
    assert calc_diff(o, ref) < (2e-3 if pv == "fp16" else 5e-3)
    assert (lse - ref_lse).abs().max() < 0.06
    # world size 1: zigzag == the two halves of one shard
    o1, l1 = ring_sageattn(q, k, v, is_causal=True, pv=pv, qk_quant_gran=gran, return_lse=True, causal_layout="zigzag")
    assert (o1.cpu().float() - ref).abs().max() < (0.08 if pv == "fp16" else 0.2)
    assert (l1.cpu() - ref_lse).abs().max() < 0.06


@pytest.mark.parametrize("shape", [
    # (B, Hq, Hk, M, N, D, causal, layout)
    (1, 2, 2, 1, 1, 64, False, "HND"),       # single token
    (1, 2, 1, 7, 63, 128, False, "NHD"),     # shorter than one tile, GQA
    (2, 4, 2, 65, 65, 64, True, "HND"),      # one key into the second tile, causal
    (1, 2, 2, 300, 130, 128, True, "HND"),   # M > N causal (top-left aligned mask, reference semantics)
    (1, 3, 3, 100, 333, 64, True, "NHD"),    # M < N causal, odd head count
    (1, 2, 2, 513, 1027, 128, False, "HND"), # ragged, several query blocks
])
@pytest.mark.parametrize("pv", ["fp16", "fp8"])
@pytest.mark.parametrize("nwaves", [8, 4])
def test_edge_shapes_vs_oracle(sa, shape, pv, nwaves):
    """Empty-ish and ragged inputs (the edge cases the reference's kernels guard: predicated loads,
    qk_int_sv_f16_cuda_sm80.cu:224-258; out-of-bound and causal masks, attn_utils.cuh:296-352; GQA head mapping):
    end-to-end operator vs the oracle, both PV variants, both workgroup sizes.  Tolerances as in the golden tests
    (fp16 PV 2e-3 / LSE 2e-3; fp8 PV 0.06 / 2e-3)."""
    from oracle import sage_oracle as O
    from sageattention_amd import _lib as L
    B, Hq, Hk, M, N, D, causal, layout = shape
    g = torch.Generator().manual_seed(M * 1000 + N)
    mk = (lambda h, n: (B, h, n, D)) if layout == "HND" else (lambda h, n: (B, n, h, D))
    q = torch.randn(mk(Hq, M), generator=g).half()
    k = torch.randn(mk(Hk, N), generator=g).half()
    v = torch.randn(mk(Hk, N), generator=g).half()
    fn = sa.sageattn_qk_int8_pv_fp16_cuda if pv == "fp16" else sa.sageattn_qk_int8_pv_fp8_cuda
    assert L.lib().sage_set_tuning(0, nwaves) == 0
    try:
        o, lse = fn(q.cuda(), k.cuda(), v.cuda(), tensor_layout=layout, is_causal=causal, return_lse=True,
                    pv_accum_dtype="fp32")
        torch.cuda.synchronize()
    finally:
        L.lib().sage_set_tuning(0, 0)
    oo, ol = O.sageattn_oracle(q, k, v, tensor_layout=layout, is_causal=causal, pv=pv, return_lse=True)
    assert torch.isfinite(o).all() and torch.isfinite(lse).all()
    assert (o.cpu().float() - oo.float()).abs().max() < (2e-3 if pv == "fp16" else 0.06)
    assert (lse.cpu() - ol).abs().max() < 2e-3


@pytest.mark.parametrize("gran", ["per_warp", "per_thread"])
@pytest.mark.parametrize("pv", ["fp16", "fp8"])
def test_fused_q_quantizer_is_bit_identical(sa, golden, gran, pv):
    """The Q quantizer folded into the attention kernel (sage_attn_fusedq_*) uses the arithmetic of the stand-alone
    quantizer K1: outputs must be BIT-identical to the two-kernel path; the fused LSE (final, natural log) matches the
    unfused one to 1e-5 (the q.km dot is summed in a different order)."""
    from sageattention_amd import core
    g, m = golden, golden.meta
    fn = sa.sageattn_qk_int8_pv_fp16_cuda if pv == "fp16" else sa.sageattn_qk_int8_pv_fp8_cuda
    kw = dict(tensor_layout=m["layout"], is_causal=bool(m["causal"]), qk_quant_gran=gran, return_lse=True, pv_accum_dtype="fp32")
    assert m["M"] <= core.FUSE_Q_MAX_SEQ
    keep = core.FUSE_Q_QUANT
    try:
        core.FUSE_Q_QUANT = True    # the default path (core.py)
        o1, l1 = fn(g.q.cuda(), g.k.cuda(), g.v.cuda(), **kw)
        core.FUSE_Q_QUANT = False
        o0, l0 = fn(g.q.cuda(), g.k.cuda(), g.v.cuda(), **kw)
    finally:
        core.FUSE_Q_QUANT = keep
    torch.cuda.synchronize()
    assert torch.equal(o1, o0), (o1.float() - o0.float()).abs().max()
    assert (l1 - l0).abs().max() < 1e-5 * max(1.0, float(l0.abs().max()))


def test_randomized_sweep_vs_oracle(sa):
    """40 seeded random configurations (layout, dtype, GQA ratio, M != N, head_dim incl. padded ones, granularity,
    causal, PV precision, smoothing flags, workgroup size) against the oracle; tolerances of the golden tests."""
    import random
    from oracle import sage_oracle as O
    from sageattention_amd import _lib as L
    rng = random.Random(20250101)
    for it in range(40):
        layout = rng.choice(["HND", "NHD"])
        dt = rng.choice([torch.float16, torch.bfloat16])
        Hk = rng.choice([1, 2, 3])
        Hq = Hk * rng.choice([1, 2, 4])
        D = rng.choice([64, 128, 64, 128, 40, 96])
        causal = rng.random() < 0.4
        M = rng.randint(1, 400)
        N = M if (causal and rng.random() < 0.7) else rng.randint(1, 600)
        B = rng.choice([1, 2])
        pv = rng.choice(["fp16", "fp8"])
        gran = rng.choice(["per_warp", "per_thread"])
        smooth_k = rng.random() < 0.8
        nw = rng.choice([0, 4, 8])
        g = torch.Generator().manual_seed(it)
        mk = (lambda h, n: (B, h, n, D)) if layout == "HND" else (lambda h, n: (B, n, h, D))
        q = torch.randn(mk(Hq, M), generator=g).to(dt)
        k = (torch.randn(mk(Hk, N), generator=g) + rng.choice([0.0, 2.0]) * torch.randn(mk(Hk, 1), generator=g)).to(dt)
        v = torch.randn(mk(Hk, N), generator=g).to(dt)
        fn = sa.sageattn_qk_int8_pv_fp16_cuda if pv == "fp16" else sa.sageattn_qk_int8_pv_fp8_cuda
        L.lib().sage_set_tuning(0, nw)
        try:
            o, lse = fn(q.cuda(), k.cuda(), v.cuda(), tensor_layout=layout, is_causal=causal, qk_quant_gran=gran,
                        smooth_k=smooth_k, return_lse=True, pv_accum_dtype="fp32")
            torch.cuda.synchronize()
        finally:
            L.lib().sage_set_tuning(0, 0)
        oo, ol = O.sageattn_oracle(q, k, v, tensor_layout=layout, is_causal=causal, qk_quant_gran=gran, pv=pv,
                                   smooth_k=smooth_k, return_lse=True)
        cfg = (it, layout, dt, Hq, Hk, D, causal, M, N, B, pv, gran, smooth_k, nw)
        assert o.shape == q.shape and torch.isfinite(o).all(), cfg
        tol = {("fp16", torch.float16): 2e-3, ("fp16", torch.bfloat16): 1.6e-2, ("fp8", torch.float16): 0.06,
               ("fp8", torch.bfloat16): 0.07}[(pv, dt)]
        assert (o.cpu().float() - oo.float()).abs().max() < tol, cfg
        assert (lse.cpu() - ol).abs().max() < 3e-3, cfg


@pytest.mark.parametrize("layout", ["HND", "NHD"])
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_sub_mean_bit_exact_vs_oracle(sa, layout, dt):
    """``sub_mean`` (reference quant.py:183-223, fused.cu:200-260) against ``oracle.sub_mean``: the mean within one ulp
    of the storage dtype (summation order differs from torch's), the smoothed tensor BIT-exact given that mean
    (subtraction in the input dtype, bf16 results converted to fp16).  CUDA-only in the reference: parity unpinned."""
    from oracle import sage_oracle as O
    torch.manual_seed(11)
    B, H, N, D = 2, 3, 333, 128
    shape = (B, H, N, D) if layout == "HND" else (B, N, H, D)
    v = (torch.randn(shape) * 2 + torch.randn(1, 1, 1, D) * 3).to(dt)
    vs, vm = sa.quant.sub_mean(v.cuda(), layout)
    assert vs.dtype == torch.float16 and vs.shape == v.shape and vm.shape == (B, H, D) and vm.dtype == dt
    _, vm_ref = O.sub_mean(v, layout)
    ulp = 2.0 ** -10 if dt == torch.float16 else 2.0 ** -7
    assert ((vm.cpu().float() - vm_ref.float()).abs() <= ulp * vm_ref.float().abs().clamp(min=2.0 ** -14)).all()
    vs_ref, _ = O.sub_mean(v, layout, vm=vm.cpu())
    assert torch.equal(vs.cpu(), vs_ref)


@pytest.mark.parametrize("gran", ["per_thread", "per_warp"])
def test_c3_headline_config_vs_oracle(sa, gran):
    """BASELINE configs[2] = (4,32,8192,128), INT8 QK^T + FP16 PV -- the shape bench.py times, through the 8-wave
    instantiation it times -- with a random V at full size against the oracle on two whole heads (first head of the
    first batch, last head of the last): |do| <= 2e-3 (fp16 output, fp32 accumulation on both sides; the tile order
    and the lazy rescale differ), LSE <= 2e-3.  Non-causal and causal."""
    from oracle import sage_oracle as O
    torch.manual_seed(5)
    B, H, N, D = 4, 32, 8192, 128
    q = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda") + torch.randn(1, H, 1, D, dtype=torch.float16, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.float16, device="cuda")
    for causal in (False, True):
        o, lse = sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v, is_causal=causal, qk_quant_gran=gran, return_lse=True)
        for sl in ((slice(0, 1), slice(0, 1)), (slice(3, 4), slice(31, 32))):
            oo, ol = O.sageattn_oracle(q[sl].cpu(), k[sl].cpu(), v[sl].cpu(), qk_quant_gran=gran, is_causal=causal,
                                       return_lse=True)
            assert (o[sl].cpu().float() - oo.float()).abs().max() < 2e-3
            assert (lse[sl].cpu() - ol).abs().max() < 2e-3


@pytest.mark.parametrize("layout", ["HND", "NHD"])
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_k_smooth_quant_is_bit_identical_to_mean_plus_quantizer(sa, layout, dt):
    """sage_k_smooth_quant (one call; for <= 16 chunks of 256 rows the quantizer finishes the mean itself) against
    sage_k_mean + sage_quant_qk_int8 (pinned by the reference fixtures above): km, scales and int8 values bit-identical,
    on ragged and long shapes, both K granularities and roundings."""
    from sageattention_amd import _lib as L
    from sageattention_amd.quant import _quant, k_mean, k_smooth_quant
    for i, (B, H, N, D) in enumerate([(1, 1, 1, 64), (2, 3, 63, 128), (1, 2, 257, 64), (2, 4, 1000, 128), (1, 5, 4096, 64),
                                      (1, 2, 4097, 128), (2, 8, 8192, 128), (4, 32, 1024, 64), (3, 32, 2048, 64),
                                      (4, 24, 1000, 128), (2, 48, 1, 64), (2, 50, 1537, 64), (4, 32, 1024, 128), (3, 40, 65, 128)]):
        g = torch.Generator(device="cuda").manual_seed(300 + i)
        shape = (B, H, N, D) if layout == "HND" else (B, N, H, D)
        k = (torch.randn(shape, device="cuda", generator=g) * 2 + torch.randn((1, 1, 1, D), device="cuda", generator=g) * 3).to(dt)
        for gran, rnd in ((L.GRAN_PER_THREAD, L.ROUND_TRITON), (L.GRAN_PER_BLOCK, L.ROUND_CUDA), (L.GRAN_PER_BLOCK, L.ROUND_TRITON)):
            km = k_mean(k, layout)
            k8, ks, _ = _quant(k, layout, gran, True, 64, 64, 1.0, rnd, mean=km, dense_heads=True)
            k8b, ksb, kmb = k_smooth_quant(k, layout, gran, rnd)
            assert torch.equal(kmb, km) and torch.equal(ksb, ks) and torch.equal(k8b, k8), (B, H, N, D, gran, rnd)


@pytest.mark.parametrize("layout", ["HND", "NHD"])
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_kv_prepare_fp8_is_bit_identical_to_the_separate_quantizers(sa, layout, dt):
    """sage_kv_prepare_fp8 (K smoothing + INT8 and the FP8 V^T in one call of two launches; launch B one unit per workgroup
    or, with enough work, the streaming kernel) against sage_k_smooth_quant and
    sage_quant_v_fp8 without V smoothing: every output bit-identical, ragged / single-row / long shapes, V outliers."""
    from sageattention_amd import _lib as L
    from sageattention_amd.quant import k_smooth_quant, kv_prepare_fp8, per_channel_fp8
    for i, (B, H, N, D) in enumerate([(1, 1, 1, 64), (2, 3, 63, 128), (1, 2, 257, 64), (2, 4, 1000, 128), (1, 5, 4096, 64),
                                      (2, 2, 4096, 128), (1, 2, 4097, 128), (1, 4, 8192, 64), (3, 2, 2050, 64),
                                      # enough (b, h, unit) work for the STREAMING K + V quantizer (several units per workgroup), ragged tails
                                      (4, 32, 2050, 64), (3, 40, 4100, 128), (2, 64, 8191, 128)]):
        g = torch.Generator(device="cuda").manual_seed(700 + i)
        shape = (B, H, N, D) if layout == "HND" else (B, N, H, D)
        k = (torch.randn(shape, device="cuda", generator=g) * 2 + torch.randn((1, 1, 1, D), device="cuda", generator=g) * 3).to(dt)
        v = (torch.randn(shape, device="cuda", generator=g) * (1 + 5 * torch.rand((1, 1, 1, D), device="cuda", generator=g))).to(dt)
        for gran, rnd in ((L.GRAN_PER_THREAD, L.ROUND_TRITON), (L.GRAN_PER_BLOCK, L.ROUND_CUDA)):
            k8, ks, km = k_smooth_quant(k, layout, gran, rnd)
            v8, vs, _ = per_channel_fp8(v, tensor_layout=layout, smooth_v=False)
            k8b, ksb, kmb, v8b, vsb = kv_prepare_fp8(k, v, layout, gran, rnd)
            assert torch.equal(kmb, km) and torch.equal(ksb, ks) and torch.equal(k8b, k8), (B, H, N, D, gran, rnd)
            assert torch.equal(vsb, vs), (B, H, N, D)
            assert v8b.shape == v8.shape and v8b.stride() == v8.stride()
            # pad columns beyond N are written as zeros by both; compare the raw bytes
            assert torch.equal(v8b.view(torch.uint8)[..., :N], v8.view(torch.uint8)[..., :N]), (B, H, N, D)
