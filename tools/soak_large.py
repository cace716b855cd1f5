import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sageattention_amd as sa
torch.manual_seed(11)
bad = 0
for (B,H,N,D,causal,pv) in [(4,32,8192,128,False,"fp16"),(4,32,8192,128,True,"fp16"),(4,32,16384,128,True,"fp8"),(4,32,8192,128,False,"fp8"),
                            (4,32,8192,64,True,"fp16"),(4,32,8192,64,False,"fp8"),(4,32,4096,64,True,"fp8"),(8,32,2048,64,True,"fp16"),
                            (8,32,2048,128,True,"fp8"),(4,32,2048,64,False,"fp16")]:
    q,k,v=(torch.randn(B,H,N,D,dtype=torch.float16,device="cuda") for _ in range(3))
    fn = sa.sageattn_qk_int8_pv_fp16_cuda if pv=="fp16" else sa.sageattn_qk_int8_pv_fp8_cuda
    o0,l0 = fn(q,k,v,is_causal=causal,return_lse=True)
    nd = 0
    for _ in range(25):
        o,l = fn(q,k,v,is_causal=causal,return_lse=True)
        nd += int(not (torch.equal(o,o0) and torch.equal(l,l0)))
    fin = bool(torch.isfinite(o0.float()).all())
    print((B,H,N,D,causal,pv), "nondeterministic", nd, "finite", fin, flush=True)
    bad += nd + (0 if fin else 1)
print("TOTAL BAD", bad)
