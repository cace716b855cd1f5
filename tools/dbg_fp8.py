import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import sageattention_amd as sa
from oracle import sage_oracle as O
from conftest import Golden
g = Golden("c1_hnd")
v8, vs, _ = sa.quant.per_channel_fp8(g.v.cuda(), smooth_v=False)
r8, rs, _ = O.per_channel_fp8(g.v, smooth_v=False)
perm = sa.quant.fp8_token_order()
nblk = v8.shape[-1] // 64
idx = (torch.arange(nblk).view(-1, 1) * 64 + perm.view(1, -1)).reshape(-1)
got = v8.cpu().view(torch.uint8); want = r8.view(torch.uint8)[..., idx]
bad = (got != want).nonzero()
vt = g.v.float().transpose(2, 3)[..., idx]
amax = g.v.float().abs().amax(2)
y = vt * (448.0 / amax).unsqueeze(-1)
for b in bad[:12]:
    b = tuple(b.tolist())
    print(b, "y=%.9g" % y[b].item(), "got", got[b].item(), v8.cpu()[b].float().item(), "want", want[b].item(), r8[..., idx][b].float().item())
