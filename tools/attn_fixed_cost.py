#!/usr/bin/env python3
"""One fused-Q operator call per (head_dim, N) for `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES ...`:
the instruction counts per wave at N and 2N separate the per-tile cost from the fixed cost (prologue + epilogue) of a wave:
  attn_fixed_cost.py            (run under rocprofv3; then tools/pmc_dispatches.py <dir> | grep attn_i8)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sageattention_amd as sa
for D in (64, 128):
    for N in (512, 1024, 2048, 4096):
        q, k, v = (torch.randn(4, 32, N, D, dtype=torch.float16, device="cuda") for _ in range(3))
        for fn in (sa.sageattn_qk_int8_pv_fp16_cuda, sa.sageattn_qk_int8_pv_fp8_cuda):
            fn(q, k, v)
        torch.cuda.synchronize()
        print("ran", D, N, flush=True)
