#!/usr/bin/env python3
"""Host-side cost of one operator call (python + ctypes + allocator), by cProfile on a tiny shape where the GPU is idle most
of the time.  usage: host_profile.py [fp16|fp8]"""
import cProfile, pstats, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sageattention_amd as sa
which = sys.argv[1] if len(sys.argv) > 1 else "fp16"
fn = sa.sageattn_qk_int8_pv_fp16_cuda if which == "fp16" else sa.sageattn_qk_int8_pv_fp8_cuda
q, k, v = (torch.randn(1, 8, 256, 64, dtype=torch.float16, device="cuda") for _ in range(3))
for _ in range(20): fn(q, k, v)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(2000): fn(q, k, v)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
