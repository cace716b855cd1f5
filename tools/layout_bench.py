import sys, os, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sageattention_amd as sa
B,H,N,D=4,32,8192,128
torch.manual_seed(0)
def t(f,n=10):
    for _ in range(3): f()
    ts=[]
    for _ in range(5):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/n)
    return statistics.median(ts)
for layout in ("HND","NHD"):
    shp=(B,H,N,D) if layout=="HND" else (B,N,H,D)
    q,k,v=(torch.randn(shp,dtype=torch.float16,device="cuda") for _ in range(3))
    for name,fn in (("fp16",sa.sageattn_qk_int8_pv_fp16_cuda),("fp8",sa.sageattn_qk_int8_pv_fp8_cuda)):
        ms=t(lambda: fn(q,k,v,tensor_layout=layout))
        print(layout,name,f"{ms:.3f} ms {4*B*H*N*N*D/ms/1e9:.0f} TFLOPS")
