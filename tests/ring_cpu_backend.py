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


class _Slots:
    def __init__(self, buf, rows, meta):
        self.buf, self.rows, self.meta = buf, rows, meta


class OracleGatherBackend:
    """CPU stand-in for sageattention_amd.ring.HipGatherBackend (schedule "gather"): whole-sequence smoothing mean and
    V scale from exchanged statistics, shards quantized into one byte slot each, gathered slots attended as ONE sequence.
    Same method names and call order as the HIP backend; a slot is a plain concatenation of the tensors' bytes."""

    def __init__(self, pv="fp8", qk_quant_gran="per_thread"):
        self.pv, self.gran = pv, qk_quant_gran

    def stats(self, k, v):
        def one(x):
            xf = x.float()
            B, H, n, D = xf.shape
            return torch.stack([xf.amax(2), xf.amin(2), xf.sum(2)], dim=2).reshape(B * H, 3, D)
        return torch.stack([one(k), one(v)] if self.pv == "fp8" else [one(k)])

    def reduce(self, all_stats, world, n_total, k, v):
        B, Hk, _, D = k.shape
        ksum = torch.zeros(B * Hk, D)
        for p in range(world):                       # fixed order, as sage_kv_stats_reduce
            ksum = ksum + all_stats[p, 0, :, 2, :]
        self.km4 = (ksum / float(n_total)).to(k.dtype).view(B, Hk, 1, D)
        self.km = self.km4.squeeze(2).contiguous()
        if self.pv == "fp8":
            vmax = all_stats[:, 1, :, 0, :].amax(0).view(B, Hk, D)
            vmin = all_stats[:, 1, :, 1, :].amin(0).view(B, Hk, D)
            self.amax = torch.maximum(vmax.abs(), vmin.abs())
            self.v_scale = self.amax / O.FP8_E4M3_MAX

    def _quantize_parts(self, k, v):
        n = k.shape[2]
        if self.gran == "per_thread":
            gid, ng = O.gid_per_thread_k(n)
            k8, ks = O.quant_int8_grouped(k, gid, ng, mean=self.km4, rounding="triton", scale_eps=1e-7)
        else:
            gid, ng = O.gid_per_block(n, 64)
            k8, ks = O.quant_int8_grouped(k, gid, ng, mean=self.km4, rounding="cuda")
        parts = [k8.contiguous(), ks.contiguous()]
        if self.pv == "fp8":
            y = v.float().transpose(2, 3) * O._ieee_div(O.FP8_E4M3_MAX, self.amax).unsqueeze(-1)   # quant.py:318-321 with the GLOBAL amax
            parts.append(y.clamp(-O.FP8_E4M3_MAX, O.FP8_E4M3_MAX).to(torch.float8_e4m3fn).contiguous())
        else:
            parts.append(v.contiguous())
        return parts

    def new_slots(self, slots, B, Hk, rows, D, device):
        dt = torch.float16
        probe = self._quantize_parts(torch.zeros(B, Hk, rows, D, dtype=self.km.dtype), torch.zeros(B, Hk, rows, D, dtype=self.km.dtype))
        meta = [(t.shape, t.dtype) for t in probe]
        nbytes = sum(t.numel() * t.element_size() for t in probe)
        return _Slots(torch.zeros((slots, nbytes), dtype=torch.uint8), rows, meta)

    def quantize(self, S, k, v):
        parts = self._quantize_parts(k, v)
        assert [(t.shape, t.dtype) for t in parts] == S.meta
        S.buf[0] = torch.cat([t.view(torch.uint8).reshape(-1) for t in parts])

    def setup(self, all_stats, world, k, v):
        B, Hk, n, D = k.shape
        self.reduce(all_stats, world, n * world, k, v)
        S = self.new_slots(world, B, Hk, n, D, k.device)
        self.quantize(S, k, v)
        return S

    def _unpack(self, S, p):
        out, off = [], 0
        for shape, dt in S.meta:
            nb = int(torch.tensor(shape).prod()) * torch.empty((), dtype=dt).element_size()
            out.append(S.buf[p][off:off + nb].view(dt).view(shape))
            off += nb
        return out

    def prepare_q(self, q, sm_scale, want_corr):
        if self.gran == "per_thread":
            gid, n = O.gid_per_thread_q(q.shape[2])
            q8, qs = O.quant_int8_grouped(q, gid, n, rounding="triton", scale_eps=1e-7)
        else:
            gid, n = O.gid_per_warp_q(q.shape[2], 128, 32)
            q8, qs = O.quant_int8_grouped(q, gid, n, rounding="cuda")
        corr = O.lse_correction(q, self.km, "HND") if want_corr else None
        return {"q": q, "q8": q8, "qs": qs, "sm_scale": sm_scale, "corr": corr}

    def slice_q(self, qstate, r0, r1):
        per = 32 if self.gran == "per_thread" else 4
        corr = qstate["corr"]
        return {"q": qstate["q"][:, :, r0:r1], "q8": qstate["q8"][:, :, r0:r1],
                "qs": qstate["qs"][:, :, r0 // 128 * per:r1 // 128 * per], "sm_scale": qstate["sm_scale"],
                "corr": None if corr is None else corr[:, :, r0:r1]}

    def attend(self, qstate, S, pos0, npos, causal):
        sl = [self._unpack(S, p) for p in range(pos0, pos0 + npos)]
        k8 = torch.cat([s[0] for s in sl], dim=2)
        ks = torch.cat([s[1] for s in sl], dim=2)
        v = torch.cat([s[2] for s in sl], dim=3 if self.pv == "fp8" else 2)
        q, q8, qs, sm = qstate["q"], qstate["q8"], qstate["qs"], qstate["sm_scale"]
        qrows = O.expand_q_scale(qs, q8.shape[2], self.gran)
        kcols = O.expand_k_scale(ks, k8.shape[2], self.gran)
        return O.attn_tile_loop(q8, k8, v, qrows, kcols, logit_mult=sm * O.LOG2E, is_causal=causal, pv=self.pv,
                                v_scale=self.v_scale if self.pv == "fp8" else None, out_dtype=q.dtype)

    def merge(self, parts, qstate, want_lse):
        lses = torch.stack([b[1] for b in parts]) / O.LOG2E
        lse = torch.logsumexp(lses, dim=0)
        w = torch.exp(lses - lse).nan_to_num(0.0)
        o = sum(b[0].float() * w[i].unsqueeze(-1) for i, b in enumerate(parts)).to(parts[0][0].dtype)
        if not want_lse:
            return o, None
        return o, lse + qstate["corr"] * qstate["sm_scale"]
