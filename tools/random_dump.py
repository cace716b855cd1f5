# usage: rand_dump.py <seed> <count> <out.pt>   (run with cwd = the tree to test)
import sys, os, random
sys.path.insert(0, os.getcwd())
import torch
import sageattention_amd as sa
from sageattention_amd import _lib as L
seed0 = int(sys.argv[1]); count = int(sys.argv[2])
rng = random.Random(seed0)
outs = []
for it in range(count):
    layout = rng.choice(["HND", "NHD"])
    dt = rng.choice([torch.float16, torch.bfloat16])
    Hk = rng.choice([1, 2, 3]); Hq = Hk * rng.choice([1, 2, 4])
    D = rng.choice([64, 128, 64, 128, 40, 96])
    causal = rng.random() < 0.5
    M = rng.randint(1, 700)
    N = M if (causal and rng.random() < 0.7) else rng.randint(1, 900)
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
    outs.append((o.cpu(), lse.cpu()))
torch.save(outs, sys.argv[3])
print("dumped", count, "from", os.getcwd(), sa.__file__)
