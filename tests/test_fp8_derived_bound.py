"""FP8-PV operator against its oracle with a DERIVED per-element bound, for the rows the flat 0.06 tolerance cannot cover.

P is rounded to OCP e4m3 (RNE, 4 significant bits: every rounded weight is within 2^-4 relative of p).  The kernel rounds
p * 2^d -- its running maximum is rescaled lazily, d in [0, 3] and not an integer -- the oracle rounds p itself: two rounding
instances, so the two P differ by at most 2 * 2^-4 * p, the numerators sum(P v) by at most 2^-3 * sum(p |v|), and both
divide by the same fp32 sum of the UNROUNDED p.  With the normalised weights W of the quantized operands:

    |o_kernel - o_oracle| <= 2^-3 * (W @ |v_dequantized|) + 2 output ulps          for every row and channel.

A row with hundreds of keys averages the roundings out and stays far inside it (there the test uses the 6-sigma form of the
same rounding model, which is what the flat 0.06 of test_gpu_parity.py approximates); a causal row with a handful of keys does not -- tools/random_parity_sweep.py found 3 such configurations in 1 400
(0.07-0.09) in round 2.  They are pinned here by name, together with a sweep of short causal shapes, against the bound that
holds for ANY e4m3 neighbour of the oracle's P and would still catch a wrong weight (an error of one key's whole p*v is 8x
the bound of that key)."""
import pytest
import torch

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import softmax_weights

pytestmark = pytest.mark.gpu

LOG2E = 1.44269504

# The three out-of-tolerance configurations the random sweep found (tools/random_parity_sweep.py; (seed, index, maxlen) of
# tests/sweep_configs.py names shapes, flags and tensors): |o - oracle| 0.091 / 0.076 / 0.0625 against the flat 0.06 ...
SWEEP_CASES = [(77, 297, 900), (5, 35, 200), (5, 59, 200)]
# ... and short shapes in which EVERY row has few keys
SHORT_CASES = [
    dict(layout="NHD", dt=torch.bfloat16, Hq=4, Hk=2, D=128, causal=True, M=7, N=7, B=2, gran="per_warp", smooth_k=True, nw=4, kbias=2.0, seed=5),
    dict(layout="HND", dt=torch.float16, Hq=4, Hk=1, D=64, causal=True, M=33, N=33, B=1, gran="per_thread", smooth_k=False, nw=8, kbias=0.0, seed=6),
    dict(layout="HND", dt=torch.float16, Hq=2, Hk=2, D=128, causal=True, M=5, N=5, B=1, gran="per_thread", smooth_k=True, nw=0, kbias=2.0, seed=7),
    dict(layout="NHD", dt=torch.float16, Hq=6, Hk=3, D=64, causal=True, M=70, N=70, B=2, gran="per_warp", smooth_k=True, nw=0, kbias=2.0, seed=8),
    dict(layout="HND", dt=torch.bfloat16, Hq=2, Hk=1, D=64, causal=True, M=130, N=130, B=1, gran="per_thread", smooth_k=True, nw=4, kbias=0.0, seed=9),
    dict(layout="HND", dt=torch.float16, Hq=2, Hk=2, D=64, causal=False, M=100, N=3, B=1, gran="per_thread", smooth_k=True, nw=0, kbias=2.0, seed=10),
    dict(layout="HND", dt=torch.float16, Hq=2, Hk=2, D=128, causal=False, M=64, N=1, B=2, gran="per_thread", smooth_k=True, nw=8, kbias=0.0, seed=11),
]


def _cases():
    from sweep_configs import config
    out = [pytest.param(config(*sc), id=f"sweep-seed{sc[0]}-case{sc[1]}") for sc in SWEEP_CASES]
    out += [pytest.param(c, id=f"short-{c['layout']}-D{c['D']}-{'causal' if c['causal'] else 'full'}-{c['M']}x{c['N']}") for c in SHORT_CASES]
    return out


@pytest.mark.parametrize("c", _cases())
def test_fp8_pv_within_the_derived_rounding_bound(c):
    import sageattention_amd as sa
    from oracle import sage_oracle as O
    from sageattention_amd import _lib as L
    from sweep_configs import tensors
    layout, dt, Hq, Hk, D, causal, M, N, gran, smooth_k, nw = (c[x] for x in ("layout", "dt", "Hq", "Hk", "D", "causal", "M", "N",
                                                                              "gran", "smooth_k", "nw"))
    assert D in (64, 128)
    q, k, v = tensors(c)
    L.lib().sage_set_tuning(0, nw)
    try:
        o, lse = sa.sageattn_qk_int8_pv_fp8_cuda(q.cuda(), k.cuda(), v.cuda(), tensor_layout=layout, is_causal=causal,
                                                 qk_quant_gran=gran, smooth_k=smooth_k, return_lse=True)
        torch.cuda.synchronize()
    finally:
        L.lib().sage_set_tuning(0, 0)
    oo, ol = O.sageattn_oracle(q, k, v, tensor_layout=layout, is_causal=causal, qk_quant_gran=gran, pv="fp8", smooth_k=smooth_k,
                               return_lse=True)
    hnd = (lambda x: x) if layout == "HND" else (lambda x: x.transpose(1, 2))
    # the quantized operands both sides multiply (quantizers are bit-exact: test_gpu_parity.py) and the weights they approximate
    km = O.k_mean(k, layout) if smooth_k else None
    quant = O.per_thread_int8 if gran == "per_thread" else O.per_warp_int8
    q8, qs, k8, ks = quant(q, k, km, tensor_layout=layout)
    W = softmax_weights(hnd(q8), hnd(k8), O.expand_q_scale(qs, M, gran), O.expand_k_scale(ks, N, gran), D ** -0.5 * LOG2E, causal)
    v8, v_scale, _ = O.per_channel_fp8(v, tensor_layout=layout, smooth_v=False)
    v8h = v8 if layout == "HND" else v8.transpose(1, 2)                                   # [B,Hk,D,Npad]
    v_deq = (v8h.float()[..., :N] * v_scale.unsqueeze(-1)).transpose(2, 3)                # [B,Hk,N,D]
    wv = W @ v_deq.abs().repeat_interleave(Hq // Hk, dim=1)
    of, oof = hnd(o.cpu()).float(), hnd(oo).float()
    ulp = 2.0 ** -10 if dt == torch.float16 else 2.0 ** -7
    bound = 2.0 ** -3 * wv + 2 * ulp * oof.abs().clamp(min=0.25)
    assert ((of - oof).abs() <= bound).all(), ((of - oof).abs() / bound).max()
    # rows with many keys: the roundings average out.  Each of the two roundings errs uniformly within +-2^-4 relative
    # (variance 2^-8 / 3), independently per key: sigma^2 = (2 * 2^-8 / 3) * sum((p v)^2) / l^2 per element.  6 sigma holds for
    # every one of the ~1e5 elements of a case (it is the tighter bound from a few dozen keys on; the flat 0.06 of the
    # operator's other tests is this bound for typical rows)
    sigma = torch.sqrt((2.0 ** -7 / 3) * ((W * W) @ (v_deq * v_deq).repeat_interleave(Hq // Hk, dim=1)))
    stat = 6 * sigma + 2 * ulp * oof.abs().clamp(min=0.25)
    assert ((of - oof).abs() <= torch.minimum(bound, stat)).all(), ((of - oof).abs() / torch.minimum(bound, stat)).max()
    # the LSE does not see P's rounding: fp32 sums of the unrounded p on both sides
    assert (lse.cpu() - ol).abs().max() < 3e-3
