"""Random operator configurations shared by tools/random_parity_sweep.py and the named regression cases of
tests/test_fp8_derived_bound.py: ``configs(seed0, count, maxlen)`` replays the sweep's random stream, so (seed0, index)
names one configuration -- shapes, flags AND tensors -- for good."""
import random

import torch


def configs(seed0, count, maxlen=900):
    rng = random.Random(seed0)
    for it in range(count):
        layout = rng.choice(["HND", "NHD"])
        dt = rng.choice([torch.float16, torch.bfloat16])
        Hk = rng.choice([1, 2, 3]); Hq = Hk * rng.choice([1, 2, 4])
        D = rng.choice([64, 128, 64, 128, 40, 96])
        causal = rng.random() < 0.5
        M = rng.randint(1, maxlen)
        N = M if (causal and rng.random() < 0.7) else rng.randint(1, maxlen)
        B = rng.choice([1, 2])
        pv = rng.choice(["fp16", "fp8"])
        gran = rng.choice(["per_warp", "per_thread"])
        smooth_k = rng.random() < 0.8
        nw = rng.choice([0, 4, 8])
        kbias = rng.choice([0.0, 2.0])
        yield dict(it=it, layout=layout, dt=dt, Hq=Hq, Hk=Hk, D=D, causal=causal, M=M, N=N, B=B, pv=pv, gran=gran,
                   smooth_k=smooth_k, nw=nw, kbias=kbias, seed=seed0 * 1000 + it)


def tensors(c):
    g = torch.Generator().manual_seed(c["seed"])
    B, D = c["B"], c["D"]
    mk = (lambda h, n: (B, h, n, D)) if c["layout"] == "HND" else (lambda h, n: (B, n, h, D))
    q = torch.randn(mk(c["Hq"], c["M"]), generator=g).to(c["dt"])
    k = (torch.randn(mk(c["Hk"], c["N"]), generator=g) + c["kbias"] * torch.randn(mk(c["Hk"], 1), generator=g)).to(c["dt"])
    v = torch.randn(mk(c["Hk"], c["N"]), generator=g).to(c["dt"])
    return q, k, v


def config(seed0, index, maxlen=900):
    for c in configs(seed0, index + 1, maxlen):
        pass
    return c


def fp8_bound_ratio(c, q, k, v, o, oo):
    """max over all output elements of |o - oracle| / bound, with the DERIVED rounding bound of the FP8-PV operator
    (tests/test_fp8_derived_bound.py): min(2^-3 (W |v_deq|), 6 sigma) + 2 output ulps.  <= 1 means inside the bound."""
    from conftest import softmax_weights
    from oracle import sage_oracle as O
    layout, dt, Hq, Hk, D, causal, M, N, gran, smooth_k = (c[x] for x in ("layout", "dt", "Hq", "Hk", "D", "causal", "M", "N",
                                                                          "gran", "smooth_k"))
    D_og = D
    if D not in (64, 128):  # padded head dims: the operands the kernels multiply are the padded ones
        pad = (64 if D < 64 else 128) - D
        q, k, v = (torch.nn.functional.pad(t, (0, pad)) for t in (q, k, v))
        D = D + pad
    hnd = (lambda x: x) if layout == "HND" else (lambda x: x.transpose(1, 2))
    km = O.k_mean(k, layout) if smooth_k else None
    quant = O.per_thread_int8 if gran == "per_thread" else O.per_warp_int8
    q8, qs, k8, ks = quant(q, k, km, tensor_layout=layout)
    W = softmax_weights(hnd(q8), hnd(k8), O.expand_q_scale(qs, M, gran), O.expand_k_scale(ks, N, gran), D_og ** -0.5 * 1.44269504, causal)
    v8, v_scale, _ = O.per_channel_fp8(v, tensor_layout=layout, smooth_v=False)
    v8h = v8 if layout == "HND" else v8.transpose(1, 2)
    v_deq = (v8h.float()[..., :N] * v_scale.unsqueeze(-1)).transpose(2, 3)[..., :D_og]
    rep = Hq // Hk
    wv = W @ v_deq.abs().repeat_interleave(rep, dim=1)
    sigma = torch.sqrt((2.0 ** -7 / 3) * ((W * W) @ (v_deq * v_deq).repeat_interleave(rep, dim=1)))
    of, oof = hnd(o).float(), hnd(oo).float()
    ulp = 2.0 ** -10 if dt == torch.float16 else 2.0 ** -7
    bound = torch.minimum(2.0 ** -3 * wv, 6 * sigma) + 2 * ulp * oof.abs().clamp(min=0.25)
    return float(((of - oof).abs() / bound).max())
