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
SWEEP_CASES = [(77, 297, 900), (5, 35, 200), (5, 59, 200),
               # round-3 sweeps of the final library (seeds 301-303, 1 050 configurations): four more of the same class
               (301, 50, 900), (302, 335, 900), (303, 257, 900), (303, 315, 900)]
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
    # |o - oracle| / bound over all elements, bound = min(2^-3 (W |v_deq|), 6 sigma) + 2 output ulps: the hard form for rows with
    # few keys (each of the two roundings errs by at most 2^-4 of p), the 6-sigma form of the same rounding model (uniform
    # errors, variance 2^-8 / 3 each, independent per key) where a row has enough keys to average -- the flat 0.06 of the
    # operator's other tests is that bound for typical rows.  Shared with tools/random_parity_sweep.py (sweep_configs.py).
    from sweep_configs import fp8_bound_ratio
    ratio = fp8_bound_ratio(c, q, k, v, o.cpu(), oo)
    assert ratio <= 1.0, ratio
    # the LSE does not see P's rounding: fp32 sums of the unrounded p on both sides
    assert (lse.cpu() - ol).abs().max() < 3e-3
