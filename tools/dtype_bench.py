import sys, torch, statistics
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sageattention_amd as sa
B,H,N,D=4,32,8192,128
torch.manual_seed(0)
for dt in (torch.float16, torch.bfloat16):
    q,k,v=(torch.randn(B,H,N,D,dtype=dt,device="cuda") for _ in range(3))
    for name,fn in (("fp16pv",sa.sageattn_qk_int8_pv_fp16_cuda),("fp8pv",sa.sageattn_qk_int8_pv_fp8_cuda)):
        for _ in range(3): fn(q,k,v)
        ts=[]
        for r in range(5):
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn(q,k,v)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/5)
        m=statistics.median(ts)
        print(dt, name, f"{m:.3f} ms  {4*B*H*N*N*D/m/1e9:.0f} TFLOPS e2e")
