"""Ring (sequence-parallel) SageAttention over ``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm).

New component: the reference has no parallelism code of its own -- it only exposes ``return_lse`` for exactly this use
(sageattention/core.py:122-124) and delegates ring/Ulysses to xDiT (example/parallel_sageattn_cogvideo.py:44-58).

Scheme.  Rank r owns the query rows and the key/value rows of sequence shard r.  Every rank
quantizes its OWN shard once (INT8 K with its local smoothing mean km_r, FP16 or FP8 V); the quantized shard plus
km_r and the scales is one contiguous byte buffer that travels around the ring with one send/recv pair per step,
double buffered, while the fused attention kernel runs on the shard that is already local.  Each step yields
(o_s, lse_s) for KV shard s, where lse_s is the natural-log LSE of the true logits (the smooth-K correction
(q.km_s)*sm_scale is applied per shard, core.py:651), and the partial results are merged ONCE at the end of the step with
    lse = log(sum_s exp(lse_s));  o = sum_s o_s*exp(lse_s-lse)     (sage_merge_attn_states_multi).
Causal: shards are contiguous in the sequence, so a shard from a later rank is skipped, the rank's own shard is
causal, earlier shards are full attention.  Per step and rank the ring moves Hk*n*D bytes of K + 2*Hk*n*D (fp16 V)
or Hk*n*D (fp8 V) over one xGMI link.

Causal load balance ("zigzag"): with contiguous shards rank 0 computes one block and rank P-1 computes P.
``causal_layout="zigzag"`` cuts the sequence into 2P chunks and gives rank r the chunks r and 2P-1-r (its local rows
are their concatenation; ``zigzag_split`` / ``zigzag_merge`` convert).  Chunk a attends chunk b fully when b < a, on
the diagonal when b == a and not at all when b > a, so against the shard of rank s a rank computes
    s < r:  (lo,lo) full + (hi,lo) full      s > r:  (hi, lo+hi) full      s == r:  (lo,lo) diag + (hi,lo) full + (hi,hi) diag
-- two half-block products per step on EVERY rank.  Shards are quantized once, whole; half-blocks are row slices
of the quantized tensors and of their scale vectors (the chunk length must be a multiple of 128 rows).

Three exchange schedules.  "gather" (default) changes the arithmetic for the better, the other two are bit-identical to
each other:
  "gather" ONE smoothing mean and ONE V scale for the whole sequence: the ranks first exchange their per-channel
           statistics (K column sums, V max/min: ~100 KB, one all-gather), so every rank quantizes its shard exactly as
           the unsharded operator would.  Each rank quantizes STRAIGHT INTO its slot of a tile-major exchange buffer
           (one record per 64-key tile: K int8 | V fp8/fp16 | k scales of all heads), posts that slot to all peers at
           once (7 xGMI links in parallel), attends its own shard while the exchange is in flight and then all remote
           shards with ONE more launch of the attention kernel over the gathered records (which form one sequence for
           every head), and merges the two partial results once.  No per-shard LSE corrections (the single correction
           q.km comes out of the Q quantizer in fp32), no staging copy, no per-shard outputs.  Non-causal, causal with
           contiguous shards, and causal with the zigzag layout (half-shard slots, _gather_zigzag).
  "ring"   P-1 rotation steps, each overlapped with one block of compute; one xGMI link per direction is busy.
  "direct" the 8 GPUs of an MI355X node are fully connected by xGMI (7 links per GPU), so every rank posts its shard to
           ALL peers at once (P-1 isend + P-1 irecv in one RCCL group), computes its local block meanwhile and then
           consumes the peers' shards in ring order as they land: all 7 links carry traffic concurrently and the
           exchange is paid once instead of P-1 times.  Costs P-1 receive buffers (C5: 7 x ~100 MB, trivial in 288 GB).
At C5 (n = 8192 rows per rank, D = 128, 32 heads) one block is ~1.1 TFLOP (~0.85 ms) while a ring step moves ~100 MB
over a single link (>1.3 ms): the ring would be communication bound, the all-link schedules are not.  Default: "gather"
(``schedule="direct"`` is the documented opt-out: per-shard smoothing, results bit-identical to "ring").
"""
import ctypes
from typing import Any, Optional

import torch
import torch.distributed as dist

from . import _lib as L
from . import _qattn
from .quant import _quant, k_mean, per_channel_fp8

__all__ = ["ring_sageattn", "HipRingBackend", "HipGatherBackend", "zigzag_split", "zigzag_merge"]


def zigzag_split(x: torch.Tensor, world: int, rank: int, dim: int = 2) -> torch.Tensor:
    """Rows of rank ``rank`` under the zigzag layout: chunks ``rank`` and ``2*world-1-rank`` of 2*world equal chunks."""
    chunks = x.chunk(2 * world, dim=dim)
    return torch.cat([chunks[rank], chunks[2 * world - 1 - rank]], dim=dim)


def zigzag_merge(parts, dim: int = 2) -> torch.Tensor:
    """Inverse of zigzag_split over the list of per-rank tensors (rank order)."""
    world = len(parts)
    chunks = [None] * (2 * world)
    for r, t in enumerate(parts):
        lo, hi = t.chunk(2, dim=dim)
        chunks[r], chunks[2 * world - 1 - r] = lo, hi
    return torch.cat(chunks, dim=dim)


class HipRingBackend:
    """Device-side steps of the ring, all through the C ABI (include/sageattn_hip.h).  Tests substitute a CPU backend
    with the same methods to exercise the ring protocol under gloo."""

    def __init__(self, pv: str = "fp16", qk_quant_gran: str = "per_thread"):
        assert pv in ("fp16", "fp8") and qk_quant_gran in ("per_warp", "per_thread")
        self.pv, self.gran = pv, qk_quant_gran
        self.code = L.GRAN_PER_THREAD if qk_quant_gran == "per_thread" else L.GRAN_PER_WARP
        self.rnd = L.ROUND_TRITON if qk_quant_gran == "per_thread" else L.ROUND_CUDA

    # -- queries: quantized once
    def prepare_q(self, q, sm_scale):
        gq = L.GRAN_PER_THREAD if self.gran == "per_thread" else L.GRAN_PER_WARP
        q8, qs, _ = _quant(q, "HND", gq, False, 128, 32, 1.0, self.rnd)
        return {"q": q, "q8": q8, "qs": qs, "sm_scale": sm_scale}

    # -- local KV shard -> dict of tensors that travel (all contiguous)
    def prepare_kv(self, k, v):
        km = k_mean(k, "HND")
        kg = L.GRAN_PER_THREAD if self.gran == "per_thread" else L.GRAN_PER_BLOCK
        k8, ks, _ = _quant(k, "HND", kg, True, 64, 64, 1.0, self.rnd, mean=km)
        parts = {"k8": k8.contiguous(), "ks": ks, "km": km}
        if self.pv == "fp16":
            parts["v"] = v.contiguous()
        else:
            v8, vsc, _ = per_channel_fp8(v, tensor_layout="HND", smooth_v=False)
            parts["v"] = v8
            parts["vs"] = vsc
        return parts

    # -- row ranges of the quantized tensors (zigzag half-blocks); r0 % 128 == 0
    def slice_q(self, qstate, r0, r1):
        per = 32 if self.gran == "per_thread" else 4  # q scales per 128 rows (BLKQ 128, WARPQ 32)
        return {"q": qstate["q"][:, :, r0:r1], "q8": qstate["q8"][:, :, r0:r1],
                "qs": qstate["qs"][:, :, r0 // 128 * per:-(-r1 // 128) * per].contiguous(), "sm_scale": qstate["sm_scale"]}

    def slice_kv(self, kv, r0, r1):
        per = 4 if self.gran == "per_thread" else 1   # k scales per 64 rows
        out = {"k8": kv["k8"][:, :, r0:r1], "ks": kv["ks"][:, :, r0 // 64 * per:-(-r1 // 64) * per].contiguous(),
               "km": kv["km"]}
        if self.pv == "fp16":
            out["v"] = kv["v"][:, :, r0:r1]
        else:  # V^T [B,Hk,D,N_pad]: tokens are the last axis, permuted only inside groups of 64
            out["v"] = kv["v"][..., r0:-(-r1 // 64) * 64]
            out["vs"] = kv["vs"]
        return out

    def lse_corrections(self, qstate, kvs):
        """Smooth-K corrections (q . km_s) for several shards in ONE pass over q: fp32 [P,B,H,M]."""
        q = qstate["q"]
        kms = torch.stack([kv["km"] for kv in kvs])                       # [P,B,Hk,D]
        g = q.shape[1] // kms.shape[2]
        if g > 1:
            kms = kms.repeat_interleave(g, dim=2)
        # fp32 products and sums (the single-GPU path accumulates q.km in fp32 inside the quantizer; 16-bit outputs would
        # cost up to ~0.2 in |q.km| with large channel means, i.e. percent-level merge weights)
        return torch.einsum("bhmd,pbhd->pbhm", q.float(), kms.float()).contiguous()

    def block_attn(self, qstate, kv, causal: bool, corr=None):
        """(o_blk [B,H,M,D] in q's dtype, lse [B,H,M] natural log of the true logits) for one KV shard.
        corr: precomputed (q . km_s) [B,H,M] fp32 (lse_corrections), else computed here."""
        q, q8, qs, sm = qstate["q"], qstate["q8"], qstate["qs"], qstate["sm_scale"]
        o = torch.empty(q.shape, dtype=q.dtype, device=q.device)
        if self.pv == "fp16":
            lse2 = _qattn._attn_f16(q8, kv["k8"], kv["v"], o, qs, kv["ks"], None, 1, int(causal), self.code, sm, 1)
        else:
            lse2 = _qattn._attn_f8(q8, kv["k8"], kv["v"], o, qs, kv["ks"], kv["vs"], None, 1, int(causal), self.code, sm, 1)
        # smooth-K correction of THIS shard: (q . km_s) * sm_scale   (core.py:613-617, 651)
        if corr is None:
            g = q.shape[1] // kv["km"].shape[1]
            km = kv["km"].repeat_interleave(g, dim=1) if g > 1 else kv["km"]
            corr = torch.einsum("bhmd,bhd->bhm", q.float(), km.float())
        corr = corr.contiguous()
        lse = torch.empty_like(lse2)
        L.check(L.lib().sage_finish_lse(lse2.data_ptr(), corr.data_ptr(), float(sm), lse.data_ptr(), lse2.numel(),
                                        L.stream_ptr(q.device)), "sage_finish_lse")
        return o, lse

    MERGE_MAX = 16  # SAGE_MERGE_MAX

    def merge_all(self, blocks):
        """(o, lse) of the union of the KV shards from the per-shard results [(o_blk, lse_blk), ...]: one multi-way
        merge pass (sage_merge_attn_states_multi) instead of one fp32 accumulator round trip per block."""
        import ctypes
        while len(blocks) > 1:
            grp, rest = blocks[:self.MERGE_MAX], blocks[self.MERGE_MAX:]
            o0 = grp[0][0]
            o = torch.empty(o0.shape, dtype=o0.dtype, device=o0.device)
            lse = torch.empty(grp[0][1].shape, dtype=torch.float32, device=o0.device)
            op = (ctypes.c_void_p * len(grp))(*[b[0].data_ptr() for b in grp])
            lp = (ctypes.c_void_p * len(grp))(*[b[1].data_ptr() for b in grp])
            L.check(L.lib().sage_merge_attn_states_multi(op, lp, len(grp), L.dtype_code(o0.dtype), o.data_ptr(), lse.data_ptr(),
                                                         lse.numel(), o0.shape[-1], L.stream_ptr(o0.device)),
                    "sage_merge_attn_states_multi")
            blocks = [(o, lse)] + rest
        return blocks[0]


class KvSlots:
    """Exchange buffer of the gather schedule: ``buf`` uint8 [slots, bytes], slot p = the quantized K/V records of
    ``rows`` sequence rows of one rank (tile-major, see HipGatherBackend)."""

    def __init__(self, buf, rows, B, Hk, D):
        self.buf, self.rows, self.B, self.Hk, self.D = buf, rows, B, Hk, D


class HipGatherBackend:
    """Device steps of the "gather" schedule, all through the C ABI (include/sageattn_hip.h, "sequence-parallel
    building blocks").  Tests substitute a CPU backend with the same methods (tests/ring_cpu_backend.py)."""

    def __init__(self, pv: str = "fp8", qk_quant_gran: str = "per_thread"):
        assert pv in ("fp16", "fp8") and qk_quant_gran in ("per_warp", "per_thread")
        self.pv, self.gran = pv, qk_quant_gran
        self.code = L.GRAN_PER_THREAD if qk_quant_gran == "per_thread" else L.GRAN_PER_WARP
        self.kcode = L.GRAN_PER_THREAD if qk_quant_gran == "per_thread" else L.GRAN_PER_BLOCK
        self.rnd = L.ROUND_TRITON if qk_quant_gran == "per_thread" else L.ROUND_CUDA
        self.pt = 4 if qk_quant_gran == "per_thread" else 1   # k scales per 64-key tile

    # -- per-channel statistics of the local shard: fp32 [c][B*Hk][3][D], c = 1 (K) or 2 (K, V)
    def stats(self, k, v):
        B, Hk, n, D = k.shape
        if n % 64:
            raise ValueError("the gather schedule needs shard lengths that are a multiple of 64 rows")
        lib, st = L.lib(), L.stream_ptr(k.device)
        c = 2 if self.pv == "fp8" else 1
        out = torch.empty((c, B * Hk, 3, D), dtype=torch.float32, device=k.device)
        ws = torch.empty(max(1, lib.sage_seq_stats_workspace_bytes(B, Hk, n, D) // 4) * c, dtype=torch.float32, device=k.device)
        for i, x in enumerate((k, v)[:c]):
            L.check(lib.sage_seq_stats(L.desc(x, "HND"), L.dtype_code(x.dtype), B, Hk, n, D, out[i].data_ptr(),
                                       ws[i * (ws.numel() // c):].data_ptr(), st), "sage_seq_stats")
        return out

    def reduce(self, all_stats, world, n_total, k, v):
        """Whole-sequence smoothing mean (and V scale) from the gathered statistics [world][c][B*Hk][3][D]."""
        B, Hk, _, D = k.shape
        lib, st, dev = L.lib(), L.stream_ptr(k.device), k.device
        c, BH = all_stats.shape[1], B * Hk
        fp8 = self.pv == "fp8"
        self.km = torch.empty((B, Hk, D), dtype=k.dtype, device=dev)
        self.v_scale = torch.empty((B, Hk, D), dtype=torch.float32, device=dev) if fp8 else None
        self.v_coef = torch.empty((B, Hk, 2, D), dtype=torch.float32, device=dev) if fp8 else None
        self.v_dtype = v.dtype
        L.check(lib.sage_kv_stats_reduce(all_stats.data_ptr(), all_stats[0, 1].data_ptr() if fp8 else None, world,
                                         c * BH * 3 * D, BH, D, n_total, L.dtype_code(k.dtype), 448.0, self.km.data_ptr(),
                                         L.ptr(self.v_scale), L.ptr(self.v_coef), st), "sage_kv_stats_reduce")

    # -- record layout of an exchange slot: per 64-key tile [K int8 of all heads | V | k scales], R bytes
    def _layout(self, B, Hk, D):
        BH = B * Hk
        kb = BH * 64 * D
        vb = kb * (2 if self.pv == "fp16" else 1)
        sb = (BH * self.pt * 4 + 15) // 16 * 16
        return BH, kb, vb, kb + vb + sb

    def new_slots(self, slots, B, Hk, rows, D, device):
        _, _, _, R = self._layout(B, Hk, D)
        return KvSlots(torch.empty((slots, rows // 64 * R), dtype=torch.uint8, device=device), rows, B, Hk, D)

    def quantize(self, S, k, v):
        """K (with the whole-sequence mean) and V of ``S.rows`` local rows straight into slot 0 of S."""
        B, Hk, n, D = k.shape
        assert n == S.rows
        BH, kb, vb, R = self._layout(B, Hk, D)
        T = n // 64
        lib, st = L.lib(), L.stream_ptr(k.device)
        base = S.buf.data_ptr()
        strides = (ctypes.c_int64 * 3)(Hk * self.pt, self.pt, R // 4)
        L.check(lib.sage_quant_k_int8_kvtiles(L.desc(k, "HND"), L.dtype_code(k.dtype), B, Hk, n, D, self.km.data_ptr(),
                                              L.SageTensor(base, Hk * 64 * D, 64 * D, D), R, base + kb + vb, strides,
                                              self.kcode, self.rnd, st), "sage_quant_k_int8_kvtiles")
        if self.pv == "fp8":
            L.check(lib.sage_quant_v_fp8_apply(L.desc(v, "HND"), L.dtype_code(v.dtype), B, Hk, n, D,
                                               L.SageTensor(base + kb, Hk * D * 64, D * 64, 64), R, self.v_coef.data_ptr(), st),
                    "sage_quant_v_fp8_apply")
        else:  # fp16/bf16 V: one strided copy into the tile records (a send buffer must be contiguous anyway)
            dst = torch.as_strided(S.buf.view(v.dtype), (T, B, Hk, 64, D), (R // 2, Hk * 64 * D, 64 * D, D, 1), kb // 2)
            dst.copy_(v.reshape(B, Hk, T, 64, D).permute(2, 0, 1, 3, 4))

    def setup(self, all_stats, world, k, v):
        """reduce + new_slots + quantize for whole shards (one slot per rank)."""
        B, Hk, n, D = k.shape
        self.reduce(all_stats, world, n * world, k, v)
        S = self.new_slots(world, B, Hk, n, D, k.device)
        self.quantize(S, k, v)
        return S

    def prepare_q(self, q, sm_scale, want_corr):
        """Queries quantized once; the smooth-K correction q.km (fp32, core.py:613-617) falls out of the same pass."""
        g = q.shape[1] // self.km.shape[1]
        q8, qs, corr = _quant(q, "HND", self.code, False, 128, 32, 1.0, self.rnd,
                              dot_vec=self.km if want_corr else None, dot_group=g)
        return {"q": q, "q8": q8, "qs": qs, "sm_scale": sm_scale, "corr": corr}

    def slice_q(self, qstate, r0, r1):
        """Query rows [r0, r1) (multiples of 128) of a prepared query state."""
        per = 32 if self.gran == "per_thread" else 4  # q scales per 128 rows (BLKQ 128, WARPQ 32)
        corr = qstate["corr"]
        return {"q": qstate["q"][:, :, r0:r1], "q8": qstate["q8"][:, :, r0:r1],
                "qs": qstate["qs"][:, :, r0 // 128 * per:r1 // 128 * per].contiguous(), "sm_scale": qstate["sm_scale"],
                "corr": None if corr is None else corr[:, :, r0:r1].contiguous()}

    def attend(self, qstate, S, pos0, npos, causal):
        """One launch of the attention kernel over the records of slots [pos0, pos0+npos) of S: (o, raw base-2 LSE)."""
        B, Hk, n, D = S.B, S.Hk, S.rows, S.D
        BH, kb, vb, R = self._layout(B, Hk, D)
        q, q8, qs = qstate["q"], qstate["q8"], qstate["qs"]
        Hq, M = q.shape[1], q.shape[2]
        lib, st = L.lib(), L.stream_ptr(q.device)
        base = S.buf.data_ptr() + pos0 * (n // 64) * R
        o = torch.empty(q.shape, dtype=q.dtype, device=q.device)
        lse2 = torch.empty((B, Hq, M), dtype=torch.float32, device=q.device)
        k8 = L.SageTensor(base, Hk * 64 * D, 64 * D, D)
        args = (B, Hq, Hk, M, npos * n, D, int(causal), self.code, 128, 32, float(qstate["sm_scale"]), st)
        if self.pv == "fp8":
            lay = L.KvLayout(R, R, Hk * self.pt, self.pt, R // 4)
            L.check(lib.sage_attn_qk_int8_pv_f8_kvtiles(L.desc(q8, "HND"), k8, L.SageTensor(base + kb, Hk * D * 64, D * 64, 64),
                                                         L.desc(o, "HND"), L.dtype_code(o.dtype), qs.data_ptr(), base + kb + vb,
                                                         self.v_scale.data_ptr(), lay, lse2.data_ptr(), *args),
                    "sage_attn_qk_int8_pv_f8_kvtiles")
        else:
            lay = L.KvLayout(R, R // 2, Hk * self.pt, self.pt, R // 4)
            L.check(lib.sage_attn_qk_int8_pv_f16_kvtiles(L.desc(q8, "HND"), k8, L.SageTensor(base + kb, Hk * 64 * D, 64 * D, D),
                                                          L.dtype_code(self.v_dtype), L.desc(o, "HND"), L.dtype_code(o.dtype),
                                                          qs.data_ptr(), base + kb + vb, lay, lse2.data_ptr(), *args),
                    "sage_attn_qk_int8_pv_f16_kvtiles")
        return o, lse2

    def merge(self, parts, qstate, want_lse):
        """(o, lse): the partial results share one smoothing vector, so they merge on their raw LSE and the correction
        (q.km)*sm_scale of core.py:651 is added once."""
        lib, st = L.lib(), L.stream_ptr(parts[0][0].device)
        corr, sm = qstate["corr"], float(qstate["sm_scale"])
        if len(parts) == 1:
            o, lse2 = parts[0]
            if not want_lse:
                return o, None
            lse = torch.empty_like(lse2)
            L.check(lib.sage_finish_lse(lse2.data_ptr(), L.ptr(corr), sm, lse.data_ptr(), lse2.numel(), st), "sage_finish_lse")
            return o, lse
        o0 = parts[0][0]
        o = torch.empty(o0.shape, dtype=o0.dtype, device=o0.device)
        lse = torch.empty(parts[0][1].shape, dtype=torch.float32, device=o0.device) if want_lse else None
        op = (ctypes.c_void_p * len(parts))(*[b[0].data_ptr() for b in parts])
        lp = (ctypes.c_void_p * len(parts))(*[b[1].data_ptr() for b in parts])
        L.check(lib.sage_merge_attn_states_multi_ex(op, lp, len(parts), L.dtype_code(o0.dtype), o.data_ptr(), L.ptr(lse),
                                                    parts[0][1].numel(), o0.shape[-1], 1.0 / 1.44269504,
                                                    L.ptr(corr) if want_lse else None, sm, st),
                "sage_merge_attn_states_multi_ex")
        return o, lse


def _all_gather_stats(st, world, group):
    if world == 1:
        return st.unsqueeze(0)
    flat = torch.empty(world * st.numel(), dtype=st.dtype, device=st.device)
    dist.all_gather_into_tensor(flat, st.reshape(-1), group=group)
    return flat.view((world,) + tuple(st.shape))


def _exchange_batches(world, nbatch):
    """Slot offsets p = 1 .. world-1 cut into at most ``nbatch`` contiguous groups, nearest ranks first: [(p_lo, p_hi), ...)
    with p_hi exclusive.  The cut depends on ``world`` only, so the send a rank posts in batch i is the receive of its peer
    in the SAME batch i (RCCL matches sends and receives of a pair in issue order)."""
    n = world - 1
    nbatch = max(1, min(nbatch, n))
    edges = [1 + (n * i + nbatch - 1) // nbatch for i in range(nbatch + 1)]
    return [(edges[i], edges[i + 1]) for i in range(nbatch) if edges[i + 1] > edges[i]]


GATHER_BATCHES = 3   # exchange batches of the gather schedule (see _gather_schedule)


def _gather_schedule(be, q, k, v, is_causal, sm_scale, return_lse, group, world, rank, peer, nbatch=None, delay=None):
    """schedule="gather" (module docstring).  Slot p of a rank's exchange buffer holds the shard of rank (rank - p) mod P:
    its own shard first, then the ranks before it -- exactly the shards a causal rank needs, contiguously.

    The exchange is posted in ``nbatch`` batches of neighbouring slots (one RCCL group each, all posted up front), and the
    rank attends the slots of a batch as soon as THAT batch has landed: own shard | batch 1 | batch 2 | ... each with one
    launch of the attention kernel over a contiguous run of slots, one multi-way merge at the end.  Round 2 waited for every
    request and then issued one launch over all remote slots, so only the local 1/P of the work overlapped the exchange
    and the step ended one whole remote launch after the slowest peer; now everything but the last batch's launch does.
    The launches and the merge order are fixed by (world, nbatch), never by arrival order: results are bit-identical
    whatever the timing (tests/test_ring_gloo.py delays one rank).  ``delay``: test hook, called before the sends are posted."""
    all_stats = _all_gather_stats(be.stats(k, v), world, group)
    S = be.setup(all_stats, world, k, v)
    G = S.buf
    qstate = be.prepare_q(q, sm_scale, return_lse)
    nremote = (rank if is_causal else world - 1) if world > 1 else 0           # slots 1 .. nremote are used
    if delay is not None:
        delay()
    batches = _exchange_batches(world, GATHER_BATCHES if nbatch is None else nbatch) if world > 1 else []
    pending = []
    for lo, hi in batches:
        ops = []
        for p_ in range(lo, hi):
            dst, src = (rank + p_) % world, (rank - p_) % world
            if not (is_causal and dst < rank):      # rank dst attends our shard (its slot p_)
                ops.append(dist.P2POp(dist.isend, G[0], peer(dst), group))
            if p_ <= nremote:
                ops.append(dist.P2POp(dist.irecv, G[p_], peer(src), group))
        pending.append((lo, min(hi, nremote + 1), dist.batch_isend_irecv(ops) if ops else []))
    parts = [be.attend(qstate, S, 0, 1, is_causal)]       # own shard, while the exchange is in flight
    for lo, hi, reqs in pending:
        for r in reqs:
            r.wait()
        if hi > lo:
            parts.append(be.attend(qstate, S, lo, hi - lo, False))  # the remote shards of this batch in one launch
    return be.merge(parts, qstate, return_lse)


def _gather_zigzag(be, q, k, v, sm_scale, return_lse, group, world, rank, peer):
    """Causal, zigzag layout (rank r owns chunks r and 2P-1-r of 2P, local rows = [lo; hi]) on the gather schedule.
    Two exchange buffers of half-shard slots:
        LO  slot p = lo half (chunk s) of rank s = (r - p) mod P          every rank needs every lo half
        HI  slot p = hi half (chunk 2P-1-s) of rank s = r + p             only ranks before s need its hi half
    and five launches per rank, the same 2P half-block products on every rank:
        lo rows (chunk r):        own lo, causal | lo halves of the ranks before r          (LO slots 1..r)
        hi rows (chunk 2P-1-r):   own hi, causal | ALL lo halves (LO slots 0..P-1) | hi halves of the ranks after r (HI 1..P-1-r)
    The two causal launches run while the exchange is in flight."""
    n = q.size(2)
    h = n // 2
    B, Hk, _, D = k.shape
    all_stats = _all_gather_stats(be.stats(k, v), world, group)
    be.reduce(all_stats, world, n * world, k, v)
    LO = be.new_slots(world, B, Hk, h, D, k.device)
    HI = be.new_slots(max(1, world - rank), B, Hk, h, D, k.device)
    be.quantize(LO, k[:, :, :h], v[:, :, :h])
    be.quantize(HI, k[:, :, h:], v[:, :, h:])
    qstate = be.prepare_q(q, sm_scale, return_lse)
    q_lo, q_hi = be.slice_q(qstate, 0, h), be.slice_q(qstate, h, n)
    ops = []
    for p_ in range(1, world):
        dst, src = (rank + p_) % world, (rank - p_) % world
        ops.append(dist.P2POp(dist.isend, LO.buf[0], peer(dst), group))      # everybody needs our lo half
        ops.append(dist.P2POp(dist.irecv, LO.buf[p_], peer(src), group))
    for s in range(rank + 1, world):                                          # hi halves of the ranks after us
        ops.append(dist.P2POp(dist.irecv, HI.buf[s - rank], peer(s), group))
    for d in range(rank):                                                     # ranks before us need our hi half
        ops.append(dist.P2POp(dist.isend, HI.buf[0], peer(d), group))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    lo_parts = [be.attend(q_lo, LO, 0, 1, True)]
    hi_parts = [be.attend(q_hi, HI, 0, 1, True)]
    for r in reqs:
        r.wait()
    if rank > 0:
        lo_parts.append(be.attend(q_lo, LO, 1, rank, False))
    hi_parts.append(be.attend(q_hi, LO, 0, world, False))
    if rank < world - 1:
        hi_parts.append(be.attend(q_hi, HI, 1, world - 1 - rank, False))
    o_lo, l_lo = be.merge(lo_parts, q_lo, return_lse)
    o_hi, l_hi = be.merge(hi_parts, q_hi, return_lse)
    o = torch.cat([o_lo, o_hi], dim=2)
    return o, (torch.cat([l_lo, l_hi], dim=2) if return_lse else None)


def _pack(parts):
    """One contiguous uint8 buffer holding every travelling tensor (16-B aligned slots) + the views into it."""
    names = sorted(parts)
    sizes = [(parts[n].numel() * parts[n].element_size() + 255) // 256 * 256 for n in names]
    dev = parts[names[0]].device
    buf = torch.empty(sum(sizes), dtype=torch.uint8, device=dev)
    views, off = {}, 0
    for n, sz in zip(names, sizes):
        t = parts[n]
        nb = t.numel() * t.element_size()
        view = buf[off:off + nb].view(t.dtype).view(t.shape)
        view.copy_(t)
        views[n] = view
        off += sz
    return buf, views


def _views_like(buf, views):
    out, off = {}, 0
    for n in sorted(views):
        t = views[n]
        nb = t.numel() * t.element_size()
        out[n] = buf[off:off + nb].view(t.dtype).view(t.shape)
        off += (nb + 255) // 256 * 256
    return out


@torch.compiler.disable
def ring_sageattn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, tensor_layout: str = "HND", is_causal: bool = False,
                  sm_scale: Optional[float] = None, group: Optional[dist.ProcessGroup] = None, pv: str = "auto",
                  qk_quant_gran: str = "per_thread", return_lse: bool = False, backend: Any = None,
                  schedule: str = "gather", causal_layout: str = "contiguous", **kwargs: Any):
    """SageAttention over a sequence sharded across the ranks of ``group``.  Equal shard lengths; rank r holds rows
    [r*n, (r+1)*n) of q, k, v (``causal_layout="contiguous"``) or the zigzag rows ``zigzag_split(x, P, r)``
    (``"zigzag"``, balances causal work; chunk length n/2 must be a multiple of 128).  Same tensor conventions as
    ``sageattn``; returns this rank's output rows (and their LSE) in the same local order.
    ``gather_batches`` (keyword, schedule "gather"): number of exchange batches (default GATHER_BATCHES)."""
    if tensor_layout == "NHD":
        q, k, v = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    elif tensor_layout != "HND":
        raise ValueError(f"Unknown tensor layout: {tensor_layout}")
    if schedule not in ("ring", "direct", "gather"):
        raise ValueError(f"Unknown schedule: {schedule}")
    if causal_layout not in ("contiguous", "zigzag"):
        raise ValueError(f"Unknown causal_layout: {causal_layout}")
    zigzag = is_causal and causal_layout == "zigzag"
    if zigzag and (q.size(2) % 256 or k.size(2) != q.size(2)):
        raise ValueError("zigzag layout needs equal q/kv shard lengths that are a multiple of 256 rows")
    if is_causal and q.size(2) != k.size(2):
        # every schedule launches the rank's own shard causal with M = q rows, N = k rows: the diagonal only lines up
        # when the two are equal (the reference asserts the same for its causal kernels, attn_qk_int8_per_block_causal.py:150)
        raise ValueError("causal sequence-parallel attention needs equal q and kv shard lengths")
    D = q.size(-1)
    if D not in (64, 128):
        raise ValueError(f"ring_sageattn supports head_dim 64 or 128, got {D}")
    if sm_scale is None:
        sm_scale = D ** -0.5
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if pv == "auto":  # the dispatcher's rule (core.dispatch_pv) on the WHOLE sequence: FP8 PV from a few thousand keys per row
        from .core import dispatch_pv
        pv = dispatch_pv(q, k, "HND", is_causal, n_kv=k.size(2) * world)

    def peer(r):  # group rank -> global rank for P2POp
        return dist.get_global_rank(group, r) if (world > 1 and group is not None) else r

    if schedule == "gather":
        gather_ok = (k.size(2) % 64 == 0 and q.size(0) == k.size(0) and (backend is None or hasattr(backend, "setup")))
        if gather_ok and backend is None:  # 32-bit tile offsets span the whole gathered buffer of one launch
            per_tile = k.size(0) * k.size(1) * 64 * D * (3 if pv == "fp16" else 2) + k.size(0) * k.size(1) * 16 + 16
            gather_ok = per_tile * (k.size(2) // 64) * world < (1 << 31) - (1 << 22)
        if gather_ok:
            be = backend if backend is not None else HipGatherBackend(pv, qk_quant_gran)
            if zigzag:
                o, lse = _gather_zigzag(be, q, k, v, sm_scale, return_lse, group, world, rank, peer)
            else:
                o, lse = _gather_schedule(be, q, k, v, is_causal, sm_scale, return_lse, group, world, rank, peer,
                                          nbatch=kwargs.get("gather_batches"), delay=kwargs.get("_gather_delay"))
            if tensor_layout == "NHD":
                o = o.transpose(1, 2)
            return (o, lse) if return_lse else o
        schedule = "direct"   # ragged shards, very large batches: the per-shard path
    be = backend if backend is not None else HipRingBackend(pv, qk_quant_gran)

    qstate = be.prepare_q(q, sm_scale)
    cur_buf, cur = _pack(be.prepare_kv(k, v))
    blocks = []  # per-shard (o, lse), merged once at the end

    n_loc = q.size(2)
    half = n_loc // 2
    if zigzag:  # separate result lists for the two half-blocks of query rows
        q_part = {"lo": be.slice_q(qstate, 0, half), "hi": be.slice_q(qstate, half, n_loc)}
        zblocks = {"lo": [], "hi": []}
        kv_rng = {"lo": (0, half), "hi": (half, n_loc), "all": (0, n_loc)}

    def skips(src, dst):  # rank dst never touches the shard of rank src
        return is_causal and not zigzag and src > dst

    def consume(views, src, corr=None):
        if skips(src, rank):
            return
        if not zigzag:
            blocks.append(be.block_attn(qstate, views, is_causal and src == rank, **({} if corr is None else {"corr": corr})))
            return
        if src < rank:
            pairs = (("lo", "lo", False), ("hi", "lo", False))
        elif src > rank:
            pairs = (("hi", "all", False),)
        else:
            pairs = (("lo", "lo", True), ("hi", "lo", False), ("hi", "hi", True))
        for qa, kb, diag in pairs:
            kv = views if kb == "all" else be.slice_kv(views, *kv_rng[kb])
            rows = slice(0, half) if qa == "lo" else slice(half, n_loc)
            zblocks[qa].append(be.block_attn(q_part[qa], kv, diag, **({} if corr is None else {"corr": corr[:, :, rows]})))

    if schedule == "ring" or world == 1:
        nxt_buf = torch.empty_like(cur_buf) if world > 1 else None
        for step in range(world):
            src = (rank - step) % world  # owner of the shard held in cur
            reqs = []
            if step + 1 < world:
                # rotate while computing: send the shard we hold, receive the next one into the other buffer
                ops = [dist.P2POp(dist.isend, cur_buf, peer((rank + 1) % world), group),
                       dist.P2POp(dist.irecv, nxt_buf, peer((rank - 1) % world), group)]
                reqs = dist.batch_isend_irecv(ops)
            consume(cur, src)
            for r in reqs:
                r.wait()
            if step + 1 < world:
                cur_buf, nxt_buf = nxt_buf, cur_buf
                cur = _views_like(cur_buf, cur)
    else:
        # direct exchange over the fully connected xGMI fabric: one send + one receive per peer, all posted at once
        order = [(rank - s) % world for s in range(1, world)]  # same consumption order as the ring
        rbufs = {src: torch.empty_like(cur_buf) for src in order}
        need = [src for src in order if not skips(src, rank)]          # shards this rank will use
        wanted_by = [dst for dst in order if not skips(rank, dst)]      # ranks that will use OUR shard
        ops = [dist.P2POp(dist.isend, cur_buf, peer(dst), group) for dst in wanted_by]
        ops += [dist.P2POp(dist.irecv, rbufs[src], peer(src), group) for src in need]
        reqs = dist.batch_isend_irecv(ops) if ops else []
        consume(cur, rank)
        for r in reqs:  # RCCL completes the group as a whole; shards are consumed in ring order afterwards
            r.wait()
        views = [_views_like(rbufs[src], cur) for src in need]
        # the smooth-K corrections of all remote shards in one pass over q (backends without the batched form do it per block)
        corrs = be.lse_corrections(qstate, views) if (views and hasattr(be, "lse_corrections")) else None
        for i, src in enumerate(need):
            consume(views[i], src, None if corrs is None else corrs[i])

    if zigzag:
        (o_lo, l_lo), (o_hi, l_hi) = be.merge_all(zblocks["lo"]), be.merge_all(zblocks["hi"])
        o, lse = torch.cat([o_lo, o_hi], dim=2), torch.cat([l_lo, l_hi], dim=2)
    else:
        o, lse = be.merge_all(blocks)
    if tensor_layout == "NHD":
        o = o.transpose(1, 2)
    return (o, lse) if return_lse else o
