"""CPU stand-in for sageattention_amd.ring.HipRingBackend, built on the oracle, so that the ring PROTOCOL (rotation,
double buffering, causal skipping, LSE merge) can run under gloo without a GPU.  Test infrastructure only."""
import torch

from oracle import sage_oracle as O


class OracleRingBackend:
    def __init__(self, pv="fp16", qk_quant_gran="per_thread"):
        self.pv, self.gran = pv, qk_quant_gran

    def prepare_q(self, q, sm_scale):
        if self.gran == "per_thread":
            gid, n = O.gid_per_thread_q(q.shape[2])
            q8, qs = O.quant_int8_grouped(q, gid, n, rounding="triton", scale_eps=1e-7)
        else:
            gid, n = O.gid_per_warp_q(q.shape[2], 128, 32)
            q8, qs = O.quant_int8_grouped(q, gid, n, rounding="cuda")
        return {"q": q, "q8": q8, "qs": qs, "sm_scale": sm_scale}

    def prepare_kv(self, k, v):
        km = O.k_mean(k, "HND")  # [B,H,1,D]
        if self.gran == "per_thread":
            gid, n = O.gid_per_thread_k(k.shape[2])
            k8, ks = O.quant_int8_grouped(k, gid, n, mean=km, rounding="triton", scale_eps=1e-7)
        else:
            gid, n = O.gid_per_block(k.shape[2], 64)
            k8, ks = O.quant_int8_grouped(k, gid, n, mean=km, rounding="cuda")
        parts = {"k8": k8.contiguous(), "ks": ks, "km": km.squeeze(2).contiguous()}
        if self.pv == "fp16":
            parts["v"] = v.contiguous()
        else:
            v8, vs, _ = O.per_channel_fp8(v, "HND", smooth_v=False)
            parts["v"], parts["vs"] = v8, vs
        return parts

    def slice_q(self, qstate, r0, r1):
        per = 32 if self.gran == "per_thread" else 4
        return {"q": qstate["q"][:, :, r0:r1], "q8": qstate["q8"][:, :, r0:r1],
                "qs": qstate["qs"][:, :, r0 // 128 * per:-(-r1 // 128) * per], "sm_scale": qstate["sm_scale"]}

    def slice_kv(self, kv, r0, r1):
        per = 4 if self.gran == "per_thread" else 1
        out = {"k8": kv["k8"][:, :, r0:r1], "ks": kv["ks"][:, :, r0 // 64 * per:-(-r1 // 64) * per], "km": kv["km"]}
        if self.pv == "fp16":
            out["v"] = kv["v"][:, :, r0:r1]
        else:
            out["v"], out["vs"] = kv["v"][..., r0:-(-r1 // 64) * 64], kv["vs"]
        return out

    def block_attn(self, qstate, kv, causal):
        q, q8, qs, sm = qstate["q"], qstate["q8"], qstate["qs"], qstate["sm_scale"]
        M, N = q8.shape[2], kv["k8"].shape[2]
        qrows = O.expand_q_scale(qs, M, self.gran)
        kcols = O.expand_k_scale(kv["ks"], N, self.gran)
        o, lse2 = O.attn_tile_loop(q8, kv["k8"], kv["v"], qrows, kcols, logit_mult=sm * O.LOG2E, is_causal=causal,
                                   pv=self.pv, v_scale=kv.get("vs"), out_dtype=q.dtype)
        corr = O.lse_correction(q, kv["km"], "HND")
        return o, lse2 / O.LOG2E + corr * sm

    def merge_all(self, blocks):
        lses = torch.stack([b[1] for b in blocks])                       # [P,B,H,M]
        lse = torch.logsumexp(lses, dim=0)
        w = torch.exp(lses - lse).nan_to_num(0.0)                        # blocks with lse = -inf weigh 0
        o = sum(b[0].float() * w[i].unsqueeze(-1) for i, b in enumerate(blocks))
        return o.to(blocks[0][0].dtype), lse
