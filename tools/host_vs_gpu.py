import time, torch, sys
sys.path.insert(0, '.')
import sageattention_amd as sa
for (B,H,N,D) in [(4,32,1024,64),(4,32,2048,64),(1,8,256,64),(4,32,1024,128)]:
    q,k,v=(torch.randn(B,H,N,D,dtype=torch.float16,device='cuda') for _ in range(3))
    for name,fn in [("fp16",sa.sageattn_qk_int8_pv_fp16_cuda),("fp8",sa.sageattn_qk_int8_pv_fp8_cuda),("auto",sa.sageattn)]:
        for _ in range(5): fn(q,k,v)
        torch.cuda.synchronize()
        # CPU submit time: no sync inside
        t0=time.perf_counter()
        for _ in range(200): fn(q,k,v)
        t1=time.perf_counter()
        torch.cuda.synchronize()
        t2=time.perf_counter()
        print(f"{(B,H,N,D)} {name}: cpu submit {(t1-t0)/200*1e6:.1f} us/call, total {(t2-t0)/200*1e6:.1f} us/call")
