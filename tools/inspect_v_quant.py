import sys, os, random
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import sageattention_amd as sa
from oracle import sage_oracle as O
g = torch.Generator().manual_seed(7 * 1000 + 110)
B, Hq, Hk, M, N, D = 1, 2, 2, 415, 415, 128
q = torch.randn((B, Hq, M, D), generator=g).to(torch.bfloat16)
k = (torch.randn((B, Hk, N, D), generator=g) + 0.0 * torch.randn((B, Hk, 1, D), generator=g)).to(torch.bfloat16)
v = torch.randn((B, Hk, N, D), generator=g).to(torch.bfloat16)
v8, vs, _ = sa.quant.per_channel_fp8(v.cuda(), tensor_layout="HND", smooth_v=False)
o8, os_, _ = O.per_channel_fp8(v, "HND", smooth_v=False)
print("v_scale equal:", torch.equal(vs.cpu(), os_), (vs.cpu() - os_).abs().max().item())
from sageattention_amd.quant import fp8_token_order
perm = fp8_token_order()
npad = v8.shape[-1]
g8 = v8.cpu().view(torch.uint8).view(B, Hk, D, npad // 64, 64)
inv = torch.empty(64, dtype=torch.long); inv[perm] = torch.arange(64)   # token t sits at position inv[t]
g8t = g8[..., inv].reshape(B, Hk, D, npad)                            # back to token order
o8u = o8.view(torch.uint8)
diff = (g8t[..., :N] != o8u[..., :N])
print("fp8 bytes differing:", int(diff.sum()), "of", diff.numel())
idx = torch.nonzero(diff)[:8]
for t in idx.tolist():
    b, h, d, n = t
    amax = os_[b, h, d].item() * 448.0
    x = v[b, h, n, d].float() * (torch.tensor(448.0) / (os_[b, h, d] * 448.0))
    print(t, "v", v[b, h, n, d].item(), "v_scale", os_[b, h, d].item(), "x(oracle formula, recomputed)", x.item(),
          "gpu byte", int(g8t[b, h, d, n]), "oracle byte", int(o8u[b, h, d, n]))
