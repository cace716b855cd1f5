#!/usr/bin/env python3
"""Accuracy of the operators against exact fp32 attention (same fp16/bf16 inputs), the metrics the SageAttention papers
report: cosine similarity, relative L1, RMSE.  Random normal inputs, and with per-channel K outliers (what smooth_k is for)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sageattention_amd as sa

def metrics(o, ref):
    o, ref = o.float().flatten(), ref.float().flatten()
    cos = torch.nn.functional.cosine_similarity(o, ref, dim=0).item()
    l1 = ((o - ref).abs().sum() / ref.abs().sum()).item()
    rmse = (o - ref).pow(2).mean().sqrt().item()
    return cos, l1, rmse

print("| inputs | head_dim | causal | operator | cos sim | rel. L1 | RMSE |")
print("|---|---|---|---|---|---|---|")
torch.manual_seed(0)
B, H, N = 2, 8, 4096
for dt in (torch.float16, torch.bfloat16):
    for D in (64, 128):
        for outl in (False, True):
            for causal in (False, True):
                q = torch.randn(B, H, N, D, device="cuda")
                k = torch.randn(B, H, N, D, device="cuda") + (8.0 * torch.randn(1, H, 1, D, device="cuda") if outl else 0.0)
                v = torch.randn(B, H, N, D, device="cuda")
                q, k, v = q.to(dt), k.to(dt), v.to(dt)
                ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float(), is_causal=causal)
                for name, fn in (("INT8/FP16-PV", sa.sageattn_qk_int8_pv_fp16_cuda), ("INT8/FP8-PV", sa.sageattn_qk_int8_pv_fp8_cuda)):
                    c, l1, r = metrics(fn(q, k, v, is_causal=causal), ref)
                    tag = ("bf16" if dt == torch.bfloat16 else "fp16") + (", K outliers" if outl else "")
                    print(f"| {tag} | {D} | {int(causal)} | {name} | {c:.6f} | {l1:.4f} | {r:.5f} |", flush=True)
