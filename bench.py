#!/usr/bin/env python3
"""bench.py -- BASELINE metric: attention forward TFLOPS at head_dim=128, seqlen=8K (+ speedup vs FA2-ROCm).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|ring]

A "step" is one pass of the whole hot path (K mean -> INT8 Q/K quantizers -> fused attention kernel) over one
batch of synthetic (q,k,v) already resident in HBM.  TFLOPS = 4*B*H*M*N*D / t (/2 causal), the reference's own
formula (bench/bench_baseline.py:31).  N=1 default workload: C3 = qk_int8_pv_fp16, (B,H,N,D)=(4,32,8192,128), the
configuration the metric is quoted on.  N>1: ring sequence-parallel `sageattn` over RCCL on (1,32,65536,128)
(BASELINE configs[4]; at 64K keys the dispatcher's choice is the INT8-QK / FP8-PV operator), strong scaling.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# dense MFMA peaks (MI355X_MICROARCH.md): int8 5.0 POPS, fp16 2.5 PF, MX-scaled fp8 5.0 PF.  Half the flops are the
# int8 QK^T, half the PV product: P_mix = 1/(0.5/P_qk + 0.5/P_pv)  
P_MIX_TFLOPS = {"fp16": 3333.0, "fp8": 5000.0}

WORKLOADS = {
    # name: (B, H, N, D, causal, variant)
    "c2": (4, 32, 2048, 64, False, "fp16"),
    "c3": (4, 32, 8192, 128, False, "fp16"),
    "c4": (4, 32, 16384, 128, True, "fp8"),
    "ring": (1, 32, 65536, 128, False, "fp8"),   # configs[4] names `sageattn`: at 64K keys its dispatcher picks the FP8-PV operator
}


def flops(B, H, M, N, D, causal):
    return 4.0 * B * H * M * N * D / (2.0 if causal else 1.0)


def time_events(fn, steps, warmup):
    """Average ms per call measured with HIP events on the current stream (the stream the kernels launch on)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(steps):
        fn()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) / steps


def cpu_baseline(D, causal):
    """The oracle (a CPU port of the same quantized algorithm) and torch SDPA fp32 on the host cores, on a bounded
    sample of the workload (same head_dim, shorter sequence, fewer heads)."""
    from oracle import sage_oracle as O
    # the oracle is elementwise-heavy torch code: beyond ~32 threads it only gets slower; state what was used
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    B, H, N = 1, 16, 8192  # half the heads of one batch element of C3 (1/8 of the workload), ~10 s of CPU work
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, H, N, D, generator=g).to(torch.float16)
    k = torch.randn(B, H, N, D, generator=g).to(torch.float16)
    v = torch.randn(B, H, N, D, generator=g).to(torch.float16)
    t0 = time.perf_counter()
    O.sageattn_oracle(q, k, v, qk_quant_gran="per_thread", is_causal=causal)
    t_port = time.perf_counter() - t0
    qf, kf, vf = q.float(), k.float(), v.float()
    t0 = time.perf_counter()
    torch.nn.functional.scaled_dot_product_attention(qf, kf, vf, is_causal=causal)
    t_sdpa = time.perf_counter() - t0
    fl = flops(B, H, N, N, D, causal)
    sample = f"(B,H,N,D)=({B},{H},{N},{D}) fp16 inputs, same head_dim, 1 pass"
    return ({"value": round(fl / t_port / 1e12, 5), "unit": "TFLOPS", "cores": cores, "kind": "port", "sample": sample,
             "seconds": round(t_port, 2)},
            {"value": round(fl / t_sdpa / 1e12, 5), "unit": "TFLOPS", "cores": cores, "kind": "torch-sdpa-fp32",
             "sample": sample, "seconds": round(t_sdpa, 2)})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 20 (200 for c2, whose step is 0.17 ms: a 4 ms timed "
                    "region is over before the clocks have settled and reads 15-20 %% low)")
    ap.add_argument("--warmup", type=int, default=None, help="default 5 (20 for c2)")
    ap.add_argument("--workload", default=None, choices=list(WORKLOADS))
    ap.add_argument("--gran", default="per_thread", choices=["per_warp", "per_thread"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fa2", action="store_true")
    ap.add_argument("--pv", default=None, choices=["fp16", "fp8"], help="override the PV precision of the workload")
    ap.add_argument("--causal", default=None, type=int, choices=[0, 1], help="override the workload's causal flag")
    ap.add_argument("--schedule", default="gather", choices=["gather", "direct", "ring"],
                    help="N>1: KV exchange schedule (gather: whole-sequence smoothing, one launch over all remote shards)")
    ap.add_argument("--sp", default="ring", choices=["ring", "ulysses"], help="N>1: sequence-parallel scheme (ring = "
                    "BASELINE configs[4]; ulysses = head-parallel all-to-all, SURVEY 8 f4)")
    ap.add_argument("--causal-layout", default="zigzag", choices=["zigzag", "contiguous"],
                    help="N>1 causal: zigzag half-blocks (balanced, default) or contiguous shards")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL (default); gloo only to "
                    "rehearse the multi-process path with several ranks on ONE GPU")
    ap.add_argument("--seq", type=int, default=None, help="override the workload's sequence length (rehearsals)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the RCCL process group even at world size 1")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or (world == 1 and args.gpus == 1), f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.backend == "gloo":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or args.force_dist
    saved_stdout = None
    if use_dist:
        # RCCL prints a version banner on the process's stdout (fd 1) at communicator creation; the contract is ONE JSON
        # line on stdout, so everything the libraries write goes to stderr until the result line is printed
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import sageattention_amd as sa
    from sageattention_amd import _lib as L, _qattn
    from sageattention_amd import core as sacore

    wl = args.workload or ("c3" if world == 1 else "ring")
    B, H, N, D, causal, variant = WORKLOADS[wl]
    variant = args.pv or variant
    N = args.seq or N
    causal = bool(args.causal) if args.causal is not None else causal
    # default windows: long enough for the clocks to settle under the operator (C3 on one box, same process state:
    # 20 steps after 5 warm-up steps 1302-1309 TFLOPS, 100 after 20: 1317-1323, 300 after 50: 1320-1326) and short enough
    # for the whole default run to take seconds; the ring workload (tens of ms per step) keeps 20 + 5
    if args.steps is None:
        args.steps = 200 if wl == "c2" else 20 if wl == "ring" else 100
    if args.warmup is None:
        args.warmup = 5 if wl == "ring" else 20
    torch.manual_seed(0)

    if wl == "ring" and use_dist:
        from sageattention_amd import ring
        n_local = N // world
        torch.manual_seed(1000 + rank)   # every rank its own shard (identical shards would hide a mixed-up exchange)
        q = torch.randn(B, H, n_local, D, dtype=torch.float16, device=dev)
        k = torch.randn(B, H, n_local, D, dtype=torch.float16, device=dev)
        v = torch.randn(B, H, n_local, D, dtype=torch.float16, device=dev)

        if args.sp == "ulysses":
            from sageattention_amd import ulysses

            def step():
                return ulysses.ulysses_sageattn(q, k, v, is_causal=causal, pv=variant, qk_quant_gran=args.gran)
            parallelism = f"ulysses{world}"
        else:
            def step():
                return ring.ring_sageattn(q, k, v, is_causal=causal, pv=variant, schedule=args.schedule,
                                          causal_layout=args.causal_layout)
            parallelism = f"seq-parallel{world}-{args.schedule}" + (f"-{args.causal_layout}" if causal else "")
    else:
        q = torch.randn(B, H, N, D, dtype=torch.float16, device=dev)
        k = torch.randn(B, H, N, D, dtype=torch.float16, device=dev)
        v = torch.randn(B, H, N, D, dtype=torch.float16, device=dev)
        entry = sa.sageattn_qk_int8_pv_fp16_cuda if variant == "fp16" else sa.sageattn_qk_int8_pv_fp8_cuda

        def step():
            return entry(q, k, v, is_causal=causal, qk_quant_gran=args.gran)
        parallelism = "single"

    # ---- the contract's timed region: W warmup, K steps, barrier + synchronize on both sides, max over ranks
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    total_flops = flops(B, H, N, N, D, causal)
    value = total_flops / (ms_per_step * 1e-3) / 1e12

    out = {
        "metric": f"attention fwd TFLOPS at head_dim={D} seqlen={N} (INT8 QK^T + {variant.upper()} PV, quantizers included)",
        "value": round(value, 2), "unit": "TFLOPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
        "dtype": "int8+fp16" if variant == "fp16" else "int8+fp8", "data": "synthetic",
        "config": {"workload": f"{wl}: qk_int8_pv_{variant} (B,H,N,D)=({B},{H},{N},{D}) causal={causal} "
                               f"qk_quant_gran={args.gran}", "global_batch": B, "seq_len": N, "head_dim": D,
                   "heads": H, "parallelism": parallelism},
    }

    if rank == 0 and world == 1:
        # ---- dominant kernel alone (pre-quantized inputs), HIP events on the launch stream
        with torch.cuda.device(dev):
            km = sa.quant.k_mean(k)
            q8, qs, k8, ks, _ = sacore._quant_qk(q, k, km, "HND", args.gran, D ** -0.5, 32, False, H, H)
            o = torch.empty_like(q)
            code = L.GRAN_PER_THREAD if args.gran == "per_thread" else L.GRAN_PER_WARP
            if variant == "fp16":
                def kern():
                    _qattn._attn_f16(q8, k8, v, o, qs, ks, None, 1, int(causal), code, D ** -0.5, 0)
            else:
                v8, vsc, _ = sa.quant.per_channel_fp8(v, smooth_v=False)

                def kern():
                    _qattn._attn_f8(q8, k8, v8, o, qs, ks, vsc, None, 1, int(causal), code, D ** -0.5, 0)
            k_ms = time_events(kern, args.steps, args.warmup)
            k_tflops = total_flops / (k_ms * 1e-3) / 1e12
            peak = P_MIX_TFLOPS[variant]
            # HBM bytes per launch from the committed PMC passes of this kernel on this workload (separate
            # FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md)
            traffic = None
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_attn_{wl}_pmc.json")))  # latest round last
            if cands:
                c = json.load(open(cands[-1]))
                if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                    traffic = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
            out["roofline"] = {"bound": "mfma", "achieved": round(k_tflops, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(k_tflops / peak, 4), "traffic": traffic,
                               # `traffic` is NOT measured by this run (PMC counters need rocprofv3 around the process): it is
                               # read from the committed counter passes of this kernel on this workload, named here
                               "traffic_from": (os.path.relpath(cands[-1], ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                                "passes, tools/profile_round.sh)") if cands else None,
                               "kernel": f"attn_i8_kernel<D={D}, pv={variant}>",
                               "kernel_ms": round(k_ms, 4), "flops_per_launch": total_flops,
                               # context, measured with tools/mfma_power.hip on random operands:
                               # the chip is power limited under this operator; TFLOP/s of (a) nothing but the
                               # kernel's MFMAs, (b) a dependency-free loop of its whole instruction mix
                               "power_limited_context": {"mfma_only": 2180, "instruction_mix": 1477}
                               if variant == "fp16" and D == 128 else None}
            pre_ms = time_events(lambda: sacore._quant_qk(q, k, sa.quant.k_mean(k), "HND", args.gran, D ** -0.5, 32,
                                                          False, H, H), args.steps, args.warmup)
            # quantizer pre-pass: algorithmic bytes = K read twice (mean, quant) + Q read once + int8 written
            pre_bytes = (q.numel() * 2 + k.numel() * 2 * 2 + q.numel() + k.numel())
            out["prepass"] = {"ms": round(pre_ms, 4), "GBps": round(pre_bytes / (pre_ms * 1e-3) / 1e9, 1),
                              "bound": "hbm", "peak_GBps": 8000}
            # the same workload with bf16 tensors (what video models run in): a bf16 V is multiplied as bf16
            # (v_mfma_f32_32x32x16_bf16), nothing is converted -- context, not the metric (the reference benches in fp16)
            qb, kb, vb = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
            b_ms = time_events(lambda: entry(qb, kb, vb, is_causal=causal, qk_quant_gran=args.gran), args.steps, args.warmup)
            out["bf16_inputs"] = {"value": round(total_flops / (b_ms * 1e-3) / 1e12, 2), "unit": "TFLOPS", "ms_per_step": round(b_ms, 4)}
            del qb, kb, vb
            if not args.no_fa2:
                try:
                    from torch.nn.attention import SDPBackend, sdpa_kernel
                    with sdpa_kernel(SDPBackend.FLASH_ATTENTION):
                        fa_ms = time_events(lambda: torch.nn.functional.scaled_dot_product_attention(
                            q, k, v, is_causal=causal), args.steps, args.warmup)
                    fa_tflops = total_flops / (fa_ms * 1e-3) / 1e12
                    out["fa2_rocm"] = {"tflops": round(fa_tflops, 2), "ms": round(fa_ms, 4),
                                       "speedup_end_to_end": round(value / fa_tflops, 3),
                                       "speedup_kernel_only": round(k_tflops / fa_tflops, 3)}
                except Exception as e:  # comparator only
                    out["fa2_rocm"] = {"error": repr(e)[:200]}
        if not args.no_cpu_baseline:
            port, sdpa = cpu_baseline(D, causal)
            out["cpu_baseline"] = port
            out["cpu_sdpa"] = sdpa
    if use_dist and world > 1 and wl == "ring":
        # context for the driver's scaling table (N = 1 runs C3, a different workload): the SAME problem -- the very tensors
        # of the timed run, gathered on rank 0 -- on ONE GPU through the single-GPU operator, a few launches after the timed
        # region (the other ranks wait at the barrier).  The first real multi-GPU run thereby VALIDATES itself: rank 0's rows
        # of the sequence-parallel result are compared with the same rows of the one-GPU result.
        o_loc = step()
        o_loc = (o_loc[0] if isinstance(o_loc, tuple) else o_loc).contiguous()
        full = []
        for t in (q, k, v):
            flat = torch.empty(world * t.numel(), dtype=t.dtype, device=dev)
            dist.all_gather_into_tensor(flat, t.contiguous().reshape(-1))
            full.append(flat.view((world,) + tuple(t.shape)))
        if rank == 0:
            try:
                zig = causal and args.sp == "ring" and args.causal_layout == "zigzag"
                if zig:
                    qa, ka, va = (ring.zigzag_merge(list(t.unbind(0))) for t in full)
                else:  # contiguous shards (ulysses shards the sequence the same way)
                    qa, ka, va = (torch.cat(list(t.unbind(0)), dim=2) for t in full)
                del full
                entry1 = sa.sageattn_qk_int8_pv_fp16_cuda if variant == "fp16" else sa.sageattn_qk_int8_pv_fp8_cuda
                o_one = entry1(qa, ka, va, is_causal=causal, qk_quant_gran=args.gran)
                mine = ring.zigzag_split(o_one, world, 0) if zig else o_one[:, :, :n_local]
                diff = float((o_loc.float() - mine.float()).abs().max())
                # same quantized operands up to the smoothing statistics' reduction order; different key order and merge
                tol = (0.06 if causal else 0.03) if variant == "fp8" else 4e-3
                out["validation"] = {"max_abs_diff_vs_one_gpu": round(diff, 6), "rows": int(mine.shape[2]), "tolerance": tol,
                                     "ok": bool(diff < tol)}
                one_ms = time_events(lambda: entry1(qa, ka, va, is_causal=causal, qk_quant_gran=args.gran), 3, 1)
                one_tf = total_flops / (one_ms * 1e-3) / 1e12
                out["one_gpu_same_problem"] = {"ms": round(one_ms, 4), "tflops": round(one_tf, 2),
                                               "speedup": round(value / one_tf, 3), "of_ideal": round(value / one_tf / world, 3)}
                del qa, ka, va
            except Exception as e:  # context only
                out["one_gpu_same_problem"] = {"error": repr(e)[:200]}
        torch.cuda.synchronize()
        dist.barrier()
    if use_dist:
        dist.destroy_process_group()
    if saved_stdout is not None:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
