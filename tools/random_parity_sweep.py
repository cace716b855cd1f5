"""Random configurations of the public operators against the oracle: random_parity_sweep.py <seed> <count> [maxlen].
The configurations come from tests/sweep_configs.py, so (seed, index) of a FAIL line names a regression case."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import sageattention_amd as sa
from oracle import sage_oracle as O
from sageattention_amd import _lib as L
from sweep_configs import configs, fp8_bound_ratio, tensors
seed0 = int(sys.argv[1]); count = int(sys.argv[2])
maxlen = int(sys.argv[3]) if len(sys.argv) > 3 else 900
t0 = time.time(); worst = {}
for c in configs(seed0, count, maxlen):
    q, k, v = tensors(c)
    pv, dt = c["pv"], c["dt"]
    fn = sa.sageattn_qk_int8_pv_fp16_cuda if pv == "fp16" else sa.sageattn_qk_int8_pv_fp8_cuda
    L.lib().sage_set_tuning(0, c["nw"])
    try:
        o, lse = fn(q.cuda(), k.cuda(), v.cuda(), tensor_layout=c["layout"], is_causal=c["causal"], qk_quant_gran=c["gran"],
                    smooth_k=c["smooth_k"], return_lse=True, pv_accum_dtype="fp32")
        torch.cuda.synchronize()
    finally:
        L.lib().sage_set_tuning(0, 0)
    oo, ol = O.sageattn_oracle(q, k, v, tensor_layout=c["layout"], is_causal=c["causal"], qk_quant_gran=c["gran"], pv=pv,
                               smooth_k=c["smooth_k"], return_lse=True)
    tol = {("fp16", torch.float16): 2e-3, ("fp16", torch.bfloat16): 1.6e-2, ("fp8", torch.float16): 0.06,
           ("fp8", torch.bfloat16): 0.07}[(pv, dt)]
    eo = (o.cpu().float() - oo.float()).abs().max().item(); el = (lse.cpu() - ol).abs().max().item()
    key = (pv, str(dt)); worst[key] = max(worst.get(key, 0), eo)
    if not (o.shape == q.shape and torch.isfinite(o).all() and eo < tol and el < 3e-3):
        note = ""
        if pv == "fp8" and o.shape == q.shape and torch.isfinite(o).all() and el < 3e-3:
            # over the FLAT tolerance: inside the bound DERIVED from e4m3's rounding (rows with few keys)?
            r = fp8_bound_ratio(c, q, k, v, o.cpu(), oo)
            note = f" derived-bound ratio {r:.2f}" + (" (inside: a few-key row, not a failure)" if r <= 1 else " (OUTSIDE)")
        print("FAIL" if "OUTSIDE" in note or not note else "OVER-FLAT", (seed0, c["it"], maxlen), {k_: str(v_) for k_, v_ in c.items()}, eo, el, note, flush=True)
print("done", count, "configs in", round(time.time() - t0, 1), "s; worst |o - oracle| per (pv, dtype):", worst)
