import sys, os, random
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import sageattention_amd as sa
from oracle import sage_oracle as O
seed0, target = 7, int(sys.argv[1])
rng = random.Random(seed0)
for it in range(target + 1):
    layout = rng.choice(["HND", "NHD"]); dt = rng.choice([torch.float16, torch.bfloat16])
    Hk = rng.choice([1, 2, 3]); Hq = Hk * rng.choice([1, 2, 4]); D = rng.choice([64, 128, 64, 128, 40, 96])
    causal = rng.random() < 0.5; M = rng.randint(1, 700)
    N = M if (causal and rng.random() < 0.7) else rng.randint(1, 900)
    B = rng.choice([1, 2]); pv = rng.choice(["fp16", "fp8"]); gran = rng.choice(["per_warp", "per_thread"])
    smooth_k = rng.random() < 0.8; nw = rng.choice([0, 4, 8])
    kb = rng.choice([0.0, 2.0])
g = torch.Generator().manual_seed(seed0 * 1000 + target)
mk = (lambda h, n: (B, h, n, D)) if layout == "HND" else (lambda h, n: (B, n, h, D))
q = torch.randn(mk(Hq, M), generator=g).to(dt)
k = (torch.randn(mk(Hk, N), generator=g) + kb * torch.randn(mk(Hk, 1), generator=g)).to(dt)
v = torch.randn(mk(Hk, N), generator=g).to(dt)
print(layout, dt, Hq, Hk, D, causal, M, N, B, pv, gran, smooth_k)
o, lse = sa.sageattn_qk_int8_pv_fp8_cuda(q.cuda(), k.cuda(), v.cuda(), tensor_layout=layout, is_causal=causal, qk_quant_gran=gran, smooth_k=smooth_k, return_lse=True, pv_accum_dtype="fp32")
oo, ol = O.sageattn_oracle(q, k, v, tensor_layout=layout, is_causal=causal, qk_quant_gran=gran, pv="fp8", smooth_k=smooth_k, return_lse=True)
ref, _ = O.sdpa_fp32(q if layout=="HND" else q.transpose(1,2), k if layout=="HND" else k.transpose(1,2), v if layout=="HND" else v.transpose(1,2), is_causal=causal, return_lse=True)
oc = o.cpu().float(); of = oo.float()
if layout == "NHD": oc, of = oc.transpose(1, 2), of.transpose(1, 2)
d = (oc - of).abs()
print("max |gpu - oracle|", d.max().item(), " max |gpu - fp32 attention|", (oc - ref).abs().max().item(), " max |oracle - fp32 attention|", (of - ref).abs().max().item())
idx = torch.nonzero(d > 0.05)
print("elements off by > 0.05:", idx.shape[0], "of", d.numel())
rows = sorted(set(idx[:, 2].tolist()))
print("query rows involved:", rows[:40])
for t in idx[:6].tolist():
    b, h, r, c = t
    print(t, "gpu", oc[b, h, r, c].item(), "oracle", of[b, h, r, c].item(), "fp32", ref[b, h, r, c].item())
