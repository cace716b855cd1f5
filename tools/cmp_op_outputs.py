import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from sageattention_amd import _lib as L
from sageattention_amd.core import _GRAN_CODE
libs=[]
for path in sys.argv[1:]:
    l = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in L.SIGNATURES.items():
        if hasattr(l, name):
            fn = getattr(l, name); fn.restype, fn.argtypes = res, args
    libs.append(l)
st = torch.cuda.current_stream().cuda_stream
B,H,N,D=2,4,256,64
torch.manual_seed(0)
q,k,v=(torch.randn(B,H,N,D,dtype=torch.float16,device="cuda") for _ in range(3))
for fuse in (1,0):
    outs=[]
    for l in libs:
        opts = L.OpOpts(_GRAN_CODE["per_thread"], 32, 1, fuse, 0)
        nb = l.sage_sageattn_workspace_bytes(0, B, H, H, N, N, D, 0, opts)
        ws = torch.zeros(nb, dtype=torch.uint8, device="cuda"); o = torch.zeros_like(q)
        r = l.sage_sageattn_pv_f16(L.desc(q,"HND"),L.desc(k,"HND"),L.desc(v,"HND"),0,L.desc(o,"HND"),None,B,H,H,N,N,D,0,D**-0.5,opts,ws.data_ptr(),nb,st)
        torch.cuda.synchronize(); outs.append((o.clone(), ws.clone()))
    d=(outs[0][0].float()-outs[1][0].float()).abs()
    print("fuse",fuse,"o max diff",d.max().item(),"n diff",(d>0).sum().item(),"of",d.numel(), "ws equal", torch.equal(outs[0][1],outs[1][1]), (outs[0][1]!=outs[1][1]).sum().item())
