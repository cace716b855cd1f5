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
