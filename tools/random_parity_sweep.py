import sys, os, random, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import sageattention_amd as sa
from oracle import sage_oracle as O
from sageattention_amd import _lib as L
seed0 = int(sys.argv[1]); count = int(sys.argv[2])
maxlen = int(sys.argv[3]) if len(sys.argv) > 3 else 900
rng = random.Random(seed0)
t0 = time.time(); worst = {}
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
    g = torch.Generator().manual_seed(seed0 * 1000 + it)
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
    cfg = (it, layout, str(dt), Hq, Hk, D, causal, M, N, B, pv, gran, smooth_k, nw)
    tol = {("fp16", torch.float16): 2e-3, ("fp16", torch.bfloat16): 1.6e-2, ("fp8", torch.float16): 0.06,
           ("fp8", torch.bfloat16): 0.07}[(pv, dt)]
    eo = (o.cpu().float() - oo.float()).abs().max().item(); el = (lse.cpu() - ol).abs().max().item()
    key = (pv, str(dt)); worst[key] = max(worst.get(key, 0), eo)
    if not (o.shape == q.shape and torch.isfinite(o).all() and eo < tol and el < 3e-3):
        print("FAIL", cfg, eo, el, flush=True)
print("done", count, "configs in", round(time.time() - t0, 1), "s; worst |o - oracle| per (pv, dtype):", worst)
