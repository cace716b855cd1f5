/*
 * sageattn_hip.h -- C ABI of the MI355X (gfx950) native SageAttention hot path.
 *
 * Drop-in boundary for the path sageattn() / sageattn_qk_int8_pv_{fp16,fp8}_* of
 * eliotwang/SageAttention (sageattention/core.py:80-905).  Every entry point below replaces one
 * or more functions that the reference binds through pybind (module names
 * sageattention._fused and sageattention._qattn_{sm80,sm89,rocm}); the reference interface each
 * one replaces is cited as file:line of the reference tree.  INTEGRATION.md shows the
 * reference-side binding.
 *
 * Conventions
 *  - plain C: device pointers, sizes, element strides; no torch types.  All pointers are DEVICE
 *    pointers of the current HIP device unless stated otherwise.
 *  - tensors are 4-D [B,H,N,D] views with unit stride on D, described by sage_tensor (strides in
 *    ELEMENTS): this covers both reference layouts, tensor_layout 1 = "HND" [B,H,N,D] and
 *    0 = "NHD" [B,N,H,D] (core.py:585; strides selected as in qk_int_sv_f16_cuda_sm80.cu:728-764).
 *  - ownership as in the reference: the caller allocates every buffer, the callee
 *    keeps no state; work is enqueued on `stream` and never synchronised.
 *  - every function returns SAGE_OK (0) or a negative sage_status; nothing is printed, nothing
 *    aborts (the reference raises through TORCH_CHECK / std::invalid_argument, utils.cuh:20-38).
 */
#ifndef SAGEATTN_HIP_H
#define SAGEATTN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAGEATTN_HIP_ABI_VERSION 3

typedef void* sage_stream_t; /* hipStream_t */

typedef enum sage_status {
  SAGE_OK = 0,
  SAGE_ERR_INVALID_ARGUMENT = -1, /* bad enum / null pointer / inconsistent sizes          */
  SAGE_ERR_UNSUPPORTED_HEAD_DIM = -2, /* head_dim not in {64,128} (dispatch_utils.h:23-34)  */
  SAGE_ERR_UNSUPPORTED = -3,      /* valid in the reference, not built here                */
  SAGE_ERR_TOO_LARGE = -4,        /* a (b,h) slice exceeds the 2^31-byte buffer window       */
  SAGE_ERR_LAUNCH = -5            /* hipGetLastError() != hipSuccess after the launch       */
} sage_status;

typedef enum sage_dtype { SAGE_F16 = 0, SAGE_BF16 = 1 } sage_dtype;

/* csrc/qattn/attn_utils.cuh:49-54 (QuantGranularity); values cross the reference's pybind ABI as
 * ints (core.py:587: per_warp = 2, per_thread = 3). */
typedef enum sage_qk_gran {
  SAGE_GRAN_PER_BLOCK = 1,
  SAGE_GRAN_PER_WARP = 2,
  SAGE_GRAN_PER_THREAD = 3
} sage_qk_gran;

/* Quantizer numerics: the reference has two statements of the INT8 quantizer that differ in the
 * last bit. */
typedef enum sage_rounding {
  SAGE_ROUND_TRITON = 0, /* scale=amax/127 (+eps per-thread); q=trunc(x/scale + 0.5*sign);
                            mean subtracted in the input dtype  (triton/quant_per_block.py:39-54) */
  SAGE_ROUND_CUDA = 1    /* amax=max(1e-7,amax); q=rint_even(x*(127/amax)) saturated;
                            mean subtracted in fp32              (csrc/fused/fused.cu:119-184)    */
} sage_rounding;

typedef struct sage_tensor {
  void* data;
  int64_t stride_b, stride_h, stride_n; /* elements */
} sage_tensor;

/* ---- library ------------------------------------------------------------------------------ */
int sage_abi_version(void);
const char* sage_status_string(int status);
/* Compile-time target of the device code in this library ("gfx950"). */
const char* sage_target_arch(void);

/* Tuning knob of the CALLING HOST THREAD (speed only, never results): it applies to the attention launches this thread
 * makes afterwards and to no other thread's.  key SAGE_TUNE_NWAVES: waves per workgroup of the attention kernels, value
 * in {0 = the library's measured choice, 4, 8}.  The one-call operators (sage_sageattn_*) take the same choice per call
 * in their options struct instead. */
typedef enum sage_tune_key { SAGE_TUNE_NWAVES = 0 } sage_tune_key;
int sage_set_tuning(int key, int value);
int sage_get_tuning(int key); /* the calling thread's value; -1 for an unknown key */

/* ---- K smoothing ---------------------------------------------------------------------------
 * km[b,h,:] = mean over n of k[b,h,n,:], fp32 accumulation, one rounding to `dtype`.
 * Replaces the torch op `k.mean(dim=seq_dim, keepdim=True)` at core.py:612,794.
 * workspace: fp32, at least sage_k_mean_workspace_bytes(B,H,N,D) bytes (deterministic two-pass
 * reduction, no atomics).  km: [B,H,D] contiguous, same dtype as k. */
size_t sage_k_mean_workspace_bytes(int B, int H, int N, int D);
int sage_k_mean(const sage_tensor* k, int dtype, int B, int H, int N, int D,
                void* km, void* workspace, sage_stream_t stream);

/* ---- INT8 Q/K quantizer ---------------------------------------------------------------------
 * Replaces, by (gran, rounding, mean, mult):
 *   quant_per_block_int8_cuda (2 overloads)        csrc/fused/fused.cu:429-592, pybind.cpp:23-25
 *   quant_per_block_int8_fuse_sub_mean_cuda         csrc/fused/fused.cu:594-682
 *   quant_per_warp_int8_cuda                        csrc/fused/fused.cu:685-768
 *   triton per_block_int8 / per_thread_int8         sageattention/triton/quant_per_block.py:48,
 *                                                   sageattention/triton/quant_per_thread.py:158
 * x: [B,H,N,D] fp16/bf16;  out: int8 same logical shape;  scale: fp32 [B,H,G] contiguous with
 *   per_block : G = ceil(N/blk)                     one scale per blk rows
 *   per_warp  : G = ceil(N/blk)*(blk/warp)          one scale per warp rows
 *   per_thread: is_key=0: G = ceil(N/blk)*(blk/warp)*8, rows with equal r%8 inside a warp group
 *               is_key=1: G = ceil(N/blk)*(blk/warp)*4, rows with equal (r%8)/2
 * mean: optional [B,H,D] (same dtype as x, contiguous) subtracted before quantization (smooth_k).
 * mult: multiplied into x (fp32) before quantization (Q of the per-block path: sm_scale*log2e).
 * lse_dot/lse_dot_vec: optional; lse_dot[b,h,n] = sum_d x[b,h,n,d]*lse_dot_vec[b,h/dot_group,d]
 *   in fp32 (the `q @ km^T` LSE correction of core.py:613-617), fp32 [B,H,N] contiguous.
 * blk in {64,128}; warp in {16,32,64} and divides blk; D in {64,128}. */
int sage_quant_qk_int8(const sage_tensor* x, int dtype, int B, int H, int N, int D,
                       const void* mean, const sage_tensor* out, float* scale,
                       int gran, int is_key, int blk, int warp, float mult, int rounding,
                       const void* lse_dot_vec, int dot_group, float* lse_dot,
                       sage_stream_t stream);

/* ---- V smoothing for the fp16-accumulate path ---------------------------------------------------
 * out = fp16(v - vm) with vm [B,H,D] (dtype of v).  Replaces sub_mean_cuda, fused.cu:770-848. */
int sage_sub_mean_f16(const sage_tensor* v, int dtype, int B, int H, int N, int D,
                      const void* vm, const sage_tensor* out, sage_stream_t stream);

/* ---- FP8 V quantizer ------------------------------------------------------------------------
 * Replaces transpose_pad_permute_cuda + scale_fuse_quant_cuda / mean_scale_fuse_quant_cuda
 * (csrc/fused/fused.cu:850-1083, sageattention/quant.py:225-322) in one fused pair of kernels.
 * v: [B,H,N,D] fp16/bf16.  v_fp8: OCP e4m3fn bytes, logical [B,H,D,Npad] with Npad=ceil64(N),
 * described by (stride_b, stride_h, stride_n:=stride of the D index); zero beyond N.
 * v_scale[b,h,d] = amax_d/scale_max (fp32 [B,H,D]); v_mean (optional, fp32 [B,H,D]): when non
 * null the channel mean (sum/ceil16(N), fused.cu:335,381) is subtracted first.
 * The reference's 16-row permutation (quant.py:234) is an NVIDIA mma-fragment artefact and is
 * not applied (the fork's HIP port disables it too, fused.hip:362-367).  Its gfx950 counterpart IS:
 * inside every 64-token block position pos holds token
 *   32*((pos&31)>>4) + (pos&3) + 8*((pos&15)>>2) + 4*(pos>>5)
 * ("MFMA order": the k order in which v_mfma_scale_f32_32x32x64_f8f6f4 receives P^T out of the S^T
 * accumulators), so v_fp8 is an opaque operand of sage_attn_qk_int8_pv_f8.  Columns of tokens >= N
 * are zero.
 * workspace: sage_quant_v_fp8_workspace_bytes(B,H,N,D) bytes. */
size_t sage_quant_v_fp8_workspace_bytes(int B, int H, int N, int D);
int sage_quant_v_fp8(const sage_tensor* v, int dtype, int B, int H, int N, int D,
                     const sage_tensor* v_fp8, float* v_scale, float* v_mean, float scale_max,
                     void* workspace, sage_stream_t stream);

/* ---- fused attention, INT8 QK^T + FP16 PV ------------------------------------------------------
 * Replaces qk_int8_sv_f16_accum_f32_attn, qk_int8_sv_f16_accum_f16_attn,
 * qk_int8_sv_f16_accum_f16_attn_inst_buf, qk_int8_sv_f16_accum_f16_fuse_v_mean_attn
 * (csrc/qattn/attn_cuda_sm80.h:19-65, qk_int_sv_f16_cuda_sm80.cu:674-1379).  On gfx950 the PV
 * MFMA accumulates in fp32 for every pv_accum_dtype the reference names.
 *   q8 [B,Hq,M,D] int8, k8 [B,Hk,N,D] int8, v [B,Hk,N,D] fp16 or bf16 (v_dtype), o [B,Hq,M,D] fp16/bf16 (o_dtype).
 *   A bf16 v is used as it is: P is rounded to bf16 and P.V runs on the bf16 MFMA with fp32 accumulation (the reference
 *   converts v to fp16 first, core.py:633 `v.to(float16)`; a caller who wants exactly that passes the converted tensor).
 *   Precision note: v_dtype = BF16 selects the bf16 P (8 significant bits) whatever o_dtype is -- with o_dtype = F16 the
 *   result carries 3 bits less in P than the fp16-V call (|do| <= 2^-8 * sum(p |v|) / l per element); bf16 outputs round
 *   at that size anyway.
 *   q_scale / k_scale: fp32 [B,Hq,Gq] / [B,Hk,Gk] with the shapes sage_quant_qk_int8 produces for
 *   (gran, blkq, warpq, blkk=64, warpk=64)  (…sm80.cu:796-805).
 *   sm_scale: logits are multiplied by sm_scale*log2(e) inside the kernel (…sm80.cu:92); must be
 *   positive and finite (SAGE_ERR_INVALID_ARGUMENT otherwise: the integer row max and the masks
 *   assume a positive dequantisation scale).
 *   logit_mult_is_one != 0: the scales already contain sm_scale*log2e (triton per_block path,
 *   attn_qk_int8_per_block.py:47) and sm_scale is ignored.
 *   v_mean: optional fp32 [B,Hk,D] added to the output rows (fuse_v_mean).
 *   lse: optional fp32 [B,Hq,M]; receives log2-domain lse of the scaled, smoothed logits
 *   (…sm80.cu:657-668); the caller applies core.py:651.
 *   is_causal: kv_idx > q_idx masked (top-left aligned, attn_utils.cuh:296-323). */
int sage_attn_qk_int8_pv_f16(const sage_tensor* q8, const sage_tensor* k8, const sage_tensor* v,
                             int v_dtype, const sage_tensor* o, int o_dtype,
                             const float* q_scale, const float* k_scale, const float* v_mean,
                             float* lse, int B, int Hq, int Hk, int M, int N, int D,
                             int is_causal, int qk_gran, int blkq, int warpq,
                             float sm_scale, int logit_mult_is_one, sage_stream_t stream);

/* ---- fused attention, INT8 QK^T + FP8 PV -------------------------------------------------------
 * Replaces qk_int8_sv_f8_accum_f32[_fuse_v_scale][_fuse_v_mean]_attn[_inst_buf] and
 * qk_int8_sv_f8_accum_f16_* (csrc/qattn/attn_cuda_sm89.h, qk_int_sv_f8_cuda_sm89.cuh:44-713) and the
 * fork's unfused qk_int8_sv_f8_accum_f32_attn (csrc/qattn/rocm/attn_rocm_gfx942.h:20-32).
 *   v_fp8: as produced by sage_quant_v_fp8 ([B,Hk,D,Npad], OCP e4m3fn);  v_scale fp32 [B,Hk,D]
 *   multiplied into the output columns (fuse_v_scale), v_mean optional. */
int sage_attn_qk_int8_pv_f8(const sage_tensor* q8, const sage_tensor* k8, const sage_tensor* v_fp8,
                            const sage_tensor* o, int o_dtype,
                            const float* q_scale, const float* k_scale, const float* v_scale,
                            const float* v_mean, float* lse, int B, int Hq, int Hk, int M, int N,
                            int D, int is_causal, int qk_gran, int blkq, int warpq,
                            float sm_scale, int logit_mult_is_one, sage_stream_t stream);

/* ---- fused attention with the Q quantizer folded into the kernel --------------------------------
 * Same operators as sage_attn_qk_int8_pv_{f16,f8}, but `q` is the fp16/bf16 query tensor: each wave
 * quantizes its own query rows in the kernel prologue with exactly the arithmetic of
 * sage_quant_qk_int8 (qk_gran per_warp -> SAGE_ROUND_CUDA, per_thread -> SAGE_ROUND_TRITON, blk 128),
 * which removes one launch and the int8 round trip of Q through HBM.  km (optional,
 * [B,Hk,D], dtype of q): when given, `lse` receives the FINAL natural-log LSE
 * lse2/log2(e) + (q.km)*sm_scale of core.py:651; otherwise lse2/log2(e).
 * Replaces the pair {quant_per_warp_int8_cuda | per_thread_int8 (Q half), qk_int8_sv_*_attn}. */
int sage_attn_fusedq_pv_f16(const sage_tensor* q, int q_dtype, const sage_tensor* k8, const sage_tensor* v,
                            int v_dtype, const sage_tensor* o, int o_dtype, const float* k_scale,
                            const void* km, const float* v_mean, float* lse, int B, int Hq, int Hk,
                            int M, int N, int D, int is_causal, int qk_gran, int warpq, float sm_scale,
                            sage_stream_t stream);
int sage_attn_fusedq_pv_f8(const sage_tensor* q, int q_dtype, const sage_tensor* k8,
                           const sage_tensor* v_fp8, const sage_tensor* o, int o_dtype,
                           const float* k_scale, const void* km, const float* v_scale,
                           const float* v_mean, float* lse, int B, int Hq, int Hk, int M, int N, int D,
                           int is_causal, int qk_gran, int warpq, float sm_scale, sage_stream_t stream);

/* ---- attention with an explicit attn_mask ----------------------------------------------------------
 * The attn_mask argument of sageattn_qk_int8_pv_fp16_triton (core.py:249-251,306-318; kernels
 * triton/attn_qk_int8_per_block.py:33-52, attn_qk_int8_per_thread.py:37-75).  Non-causal, 16-bit PV (v fp16 or
 * bf16, multiplied in its own type exactly as by sage_attn_qk_int8_pv_f16).
 * attn_mask: device pointer to a [B,Hq,M,N] VIEW given by mask_strides[4] in ELEMENTS (host array; 0 =
 * broadcast dimension).  mask_kind 1: bool/uint8, zero = masked (the reference adds -1e6 to the base-2
 * logit); 2: fp16, 3: bf16 additive mask, added to the base-2 logit exactly as the reference does (after
 * its sm_scale*log2(e) scaling).  Rows without any allowed key are undefined in the reference (they
 * depend on its 128x64 tile skipping) and here. */
int sage_attn_qk_int8_pv_f16_masked(const sage_tensor* q8, const sage_tensor* k8, const sage_tensor* v,
                                    int v_dtype, const sage_tensor* o, int o_dtype,
                                    const float* q_scale, const float* k_scale, const void* attn_mask,
                                    int mask_kind, const int64_t* mask_strides, float* lse, int B, int Hq,
                                    int Hk, int M, int N, int D, int qk_gran, int blkq, int warpq,
                                    float sm_scale, int logit_mult_is_one, sage_stream_t stream);

/* ---- packed variable-length sequences (sageattn_varlen, core.py:363-477) ------------------------
 * q/k/v/o are packed [total_tokens, H, D] tensors described as sage_tensor with stride_b unused;
 * sequence s owns rows [cu_seqlens[s], cu_seqlens[s+1]) (int32, device memory, num_seqs+1 entries).
 * Quantization blocks restart at every sequence start (triton/quant_per_block_varlen.py:21-58); the
 * scale tensors are [num_seqs, H, G(max_seqlen)] (a private layout: the reference's cumulative block
 * offsets, quant_per_block_varlen.py:73-80, are not needed).  The mean vector is per (head, channel)
 * over ALL packed tokens, [1,H,D] (core.py:461).  FP16 PV only and no LSE, as in the reference.
 * Replace the Triton varlen quantizer + attn_qk_int8_block_varlen.py / attn_qk_int8_per_block_causal_varlen.py. */
int sage_quant_qk_int8_varlen(const sage_tensor* x, int dtype, const int* cu_seqlens, int num_seqs, int H,
                              int max_seqlen, int D, const void* mean, const sage_tensor* out, float* scale,
                              int gran, int is_key, int blk, int warp, float mult, int rounding,
                              sage_stream_t stream);
int sage_attn_qk_int8_pv_f16_varlen(const sage_tensor* q8, const sage_tensor* k8, const sage_tensor* v,
                                    int v_dtype, const sage_tensor* o, int o_dtype,
                                    const float* q_scale, const float* k_scale,
                                    const int* cu_seqlens_q, const int* cu_seqlens_k, int num_seqs,
                                    int Hq, int Hk, int max_seqlen_q, int max_seqlen_k, int D,
                                    int is_causal, int qk_gran, int blkq, int warpq, float sm_scale,
                                    int logit_mult_is_one, sage_stream_t stream);

/* ---- ring attention merge (new; the reference only exposes return_lse, core.py:122-124) -------
 * In place: (o_acc, lse_acc) <- merge((o_acc, lse_acc), (o_blk, lse_blk)) with
 *   lse = logaddexp(lse_a, lse_b);  o = o_a*exp(lse_a-lse) + o_b*exp(lse_b-lse).
 * o_acc: fp32 [rows, D] contiguous; lse_acc / lse_blk: fp32 [rows] natural log;
 * o_blk: fp16/bf16 [rows, D] contiguous. */
int sage_merge_attn_states(float* o_acc, float* lse_acc, const void* o_blk, int o_dtype,
                           const float* lse_blk, int64_t rows, int D, sage_stream_t stream);

/* Multi-way form: (o_out, lse_out) = merge of `count` block results (1 <= count <= SAGE_MERGE_MAX) in one pass,
 *   lse = log(sum_i exp(lse_i));  o = sum_i o_i * exp(lse_i - lse)   (blocks with lse_i = -inf weigh 0).
 * o_blks[i]: fp16/bf16 [rows, D] contiguous (o_dtype), lse_blks[i]: fp32 [rows] natural log; HOST arrays of
 * device pointers.  o_out: [rows, D] in o_dtype; lse_out: fp32 [rows] or NULL.  A ring step over P shards otherwise
 * passes the fp32 accumulator through HBM P times. */
#define SAGE_MERGE_MAX 16
int sage_merge_attn_states_multi(const void* const* o_blks, const float* const* lse_blks, int count,
                                 int o_dtype, void* o_out, float* lse_out, int64_t rows, int D,
                                 sage_stream_t stream);

/* lse_out[i] = lse2[i]/log2(e) + (corr ? corr[i]*sm_scale : 0)   (core.py:651), n elements. */
int sage_finish_lse(const float* lse2, const float* corr, float sm_scale, float* lse_out,
                    int64_t n, sage_stream_t stream);

/* ---- K smoothing + quantization in one call ------------------------------------------
 * = sage_k_mean + sage_quant_qk_int8(is_key = 1, mean = km, blk 64, mult 1), bit-identical, in TWO launches at every length:
 * the sequence is cut into at most 16 chunks whose column sums the quantizer finishes itself (one launch less than the pair).  gran: SAGE_GRAN_PER_BLOCK or SAGE_GRAN_PER_THREAD; km: [B,H,D] out (dtype of k); workspace as
 * sage_k_mean.  Replaces `k.mean` (core.py:612) + quant_per_block_int8_fuse_sub_mean_cuda (fused.cu:594-682) / the K half
 * of per_thread_int8 (triton/quant_per_thread.py:48-102,162-163). */
int sage_k_smooth_quant(const sage_tensor* k, int dtype, int B, int H, int N, int D, const sage_tensor* out,
                        float* scale, void* km, int gran, int rounding, void* workspace, sage_stream_t stream);

/* The whole K/V side of the FP8-PV operator's pre-pass as ONE call: km + INT8 K (as sage_k_smooth_quant) and the FP8 V^T
 * with its per-channel scale (as sage_quant_v_fp8 with v_mean = NULL, i.e. smooth_v = False), k and v of equal shape
 * [B,H,N,D].  At every length it runs as two launches (K and V column statistics per chunk; both quantizers, each finishing
 * its own statistics) instead of five.  Results are bit-identical to the two separate entry points.  Replaces core.py:612 + :621-624 (K half) + per_channel_fp8 (quant.py:225-322, fused.cu:262-427). */
size_t sage_kv_prepare_fp8_workspace_bytes(int B, int H, int N, int D);
int sage_kv_prepare_fp8(const sage_tensor* k, const sage_tensor* v, int dtype, int B, int H, int N, int D,
                        const sage_tensor* k_int8, float* k_scale, void* km, int gran, int rounding,
                        const sage_tensor* v_fp8, float* v_scale, float scale_max, void* workspace, sage_stream_t stream);

/* ---- one-call operators ---------------------------------------------------------------------------------------------
 * The whole body of sageattn_qk_int8_pv_fp16_cuda (core.py:604-651) / sageattn_qk_int8_pv_fp8_cuda (core.py:786-905)
 * below their argument checks as ONE call with ONE caller-provided workspace: km = mean(k) and the INT8 K quantizer,
 * the FP8 V^T quantizer (pv_f8), the Q quantizer (folded into the attention kernel's prologue unless fuse_q = 0),
 * the fused attention kernel and the LSE fix of core.py:651.  It sequences the library's own entry points on `stream`
 * (sage_k_smooth_quant | sage_kv_prepare_fp8, sage_quant_qk_int8, sage_attn_{fusedq,qk_int8}_pv_*, sage_finish_lse), so
 * results are bit-identical to calling those one by one.  q, k, v, o: fp16 or bf16 (one dtype), head_dim 64 or 128 (the
 * caller pads, core.py:592-601); lse: optional fp32 [B,Hq,M], receives the natural-log LSE of the un-smoothed logits.
 * opts: quantization granularity of the reference's `qk_quant_gran` (per_warp -> CUDA quantizer numerics, per_thread ->
 * Triton numerics, core.py:621-624); warpq 32 (16 = the reference's "fp16+fp32" per-warp variant, core.py:622); smooth_k
 * must be 1 (core.py:612; without smoothing use the separate entry points); fuse_q -1 = the library's choice; nwaves =
 * the per-call form of SAGE_TUNE_NWAVES (0 = the library's measured choice).
 * workspace: at least sage_sageattn_workspace_bytes(...) bytes, 16-byte aligned, private to the call until it has
 * finished on `stream`. */
typedef struct sage_op_opts {
  int qk_gran;  /* SAGE_GRAN_PER_WARP | SAGE_GRAN_PER_THREAD */
  int warpq;    /* 32 (or 16) */
  int smooth_k; /* 1 */
  int fuse_q;   /* -1 | 0 | 1 */
  int nwaves;   /* 0 | 4 | 8 */
  int reserved[3];
} sage_op_opts;
size_t sage_sageattn_workspace_bytes(int pv_fp8, int B, int Hq, int Hk, int M, int N, int D, int want_lse,
                                     const sage_op_opts* opts);
int sage_sageattn_pv_f16(const sage_tensor* q, const sage_tensor* k, const sage_tensor* v, int dtype,
                         const sage_tensor* o, float* lse, int B, int Hq, int Hk, int M, int N, int D, int is_causal,
                         float sm_scale, const sage_op_opts* opts, void* workspace, size_t workspace_bytes,
                         sage_stream_t stream);
int sage_sageattn_pv_f8(const sage_tensor* q, const sage_tensor* k, const sage_tensor* v, int dtype,
                        const sage_tensor* o, float* lse, int B, int Hq, int Hk, int M, int N, int D, int is_causal,
                        float sm_scale, float scale_max, const sage_op_opts* opts, void* workspace,
                        size_t workspace_bytes, sage_stream_t stream);

/* ==== sequence-parallel building blocks (new: the reference has no parallelism code; its hook is
 * return_lse, core.py:122-124, and its multi-GPU launcher delegates to xDiT, example/parallel_sageattn_cogvideo.py:40-52).
 * With ONE smoothing mean and ONE V scale for the whole sequence (statistics exchanged first: a few KB), the quantized
 * K/V shards of all ranks are exactly the operands of the unsharded operator, so a rank attends the gathered shards
 * with plain launches of the attention kernel: no per-shard LSE corrections, no per-shard outputs.
 * The exchange buffers are TILE-MAJOR: tile j (64 keys) of every (b, h_kv) is one contiguous block,
 *   k8 [tiles][B][Hk][64][D] int8, v fp16 [tiles][B][Hk][64][D] / v fp8 [tiles][B][Hk][D][64], k_scale [tiles][B][Hk][4|1],
 * so the tiles received from all ranks, stored one rank after the other, form one sequence for every head. ==== */

/* KV tile layout of an attention call: distance between consecutive 64-key tiles of one (b, h_kv).
 * k_tile_stride: int8 elements (= bytes) in k8; v_tile_stride: elements of v (fp16: 2 bytes each; fp8: bytes);
 * 0 = dense (64 * stride_n; fp8: 64).  Within a tile rows keep the sage_tensor's stride_n.
 * ks_stride_{b,h,tile}: floats between the k scales of consecutive batches / kv heads / 64-key tiles; all 0 = the dense
 * [B,Hk,Gk] layout.  per_thread scales are read 16 B at a time: multiples of 4. */
typedef struct sage_kv_layout {
  int64_t k_tile_stride, v_tile_stride;
  int64_t ks_stride_b, ks_stride_h, ks_stride_tile;
} sage_kv_layout;

/* sage_attn_qk_int8_pv_{f16,f8} on K/V operands in an explicit tile layout (non-causal or causal; no v_mean).
 * The raw base-2 LSE is returned as there. */
int sage_attn_qk_int8_pv_f16_kvtiles(const sage_tensor* q8, const sage_tensor* k8, const sage_tensor* v, int v_dtype,
                                     const sage_tensor* o, int o_dtype, const float* q_scale, const float* k_scale,
                                     const sage_kv_layout* kv_layout, float* lse, int B, int Hq, int Hk, int M, int N,
                                     int D, int is_causal, int qk_gran, int blkq, int warpq, float sm_scale,
                                     sage_stream_t stream);
int sage_attn_qk_int8_pv_f8_kvtiles(const sage_tensor* q8, const sage_tensor* k8, const sage_tensor* v_fp8,
                                    const sage_tensor* o, int o_dtype, const float* q_scale, const float* k_scale,
                                    const float* v_scale, const sage_kv_layout* kv_layout, float* lse, int B, int Hq,
                                    int Hk, int M, int N, int D, int is_causal, int qk_gran, int blkq, int warpq,
                                    float sm_scale, sage_stream_t stream);

/* Per-channel statistics of a [B,H,N,D] fp16/bf16 tensor over its N rows: stats fp32 [B,H,3,D] = (max, min, sum),
 * deterministic two-level reduction.  The local halves of `k.mean` (core.py:612) and of the per-channel amax of
 * per_channel_fp8 (quant.py:225-322, fused.cu:316-427).  workspace: sage_seq_stats_workspace_bytes bytes. */
size_t sage_seq_stats_workspace_bytes(int B, int H, int N, int D);
int sage_seq_stats(const sage_tensor* x, int dtype, int B, int H, int N, int D, float* stats, void* workspace,
                   sage_stream_t stream);

/* Combine the statistics of `parts` sequence shards (k_stats / v_stats: fp32 [BH][3][D] per shard, shard p at
 * + p*part_stride floats, e.g. an all-gather of sage_seq_stats results; v_stats may be NULL) into the operands of the
 * quantizers:
 *   km [BH][D] (dtype) = sum of sums / n_total                      (core.py:612 over the WHOLE sequence)
 *   v_scale fp32 [BH][D] = max |v| / scale_max,  v_coef fp32 [BH][2][D] = (0, scale_max / max |v|)   (quant.py:228,318-321)
 * Fixed summation order: every rank computes identical bits from the same gathered statistics. */
int sage_kv_stats_reduce(const float* k_stats, const float* v_stats, int parts, int64_t part_stride, int BH, int D,
                         int64_t n_total, int dtype, float scale_max, void* km, float* v_scale, float* v_coef,
                         sage_stream_t stream);

/* sage_quant_qk_int8 for K with tile-major results: out row r of (b,h) is written at
 * out->data + b*stride_b + h*stride_h + (r/64)*out_tile_stride + (r%64)*stride_n, its scales at
 * scale + b*scale_strides[0] + h*scale_strides[1] + (r/64)*scale_strides[2] (+ 0..3).  blk = 64. */
int sage_quant_k_int8_kvtiles(const sage_tensor* k, int dtype, int B, int H, int N, int D, const void* mean,
                              const sage_tensor* out, int64_t out_tile_stride, float* scale,
                              const int64_t* scale_strides, int gran, int rounding, sage_stream_t stream);

/* Second half of sage_quant_v_fp8 alone: v -> e4m3 with GIVEN coefficients v_coef [B,H,2,D] = (mean, scale_max/amax)
 * (sage_kv_stats_reduce), token block t of (b,h) written at v_fp8->data + b*stride_b + h*stride_h + t*out_tile_stride
 * (bytes; 0 = 64: the dense [B,H,D,Npad] layout), channel d at + d*stride_n. */
int sage_quant_v_fp8_apply(const sage_tensor* v, int dtype, int B, int H, int N, int D, const sage_tensor* v_fp8,
                           int64_t out_tile_stride, const float* v_coef, sage_stream_t stream);

/* sage_merge_attn_states_multi for partial results that share one smoothing vector: the inputs' LSE are multiplied by
 * lse_in_mult first (1/log2(e) for the raw base-2 LSE of the attention entry points) and
 * lse_out = log(sum) + (corr ? corr*corr_mult : 0)   (core.py:651 applied once, after the merge). */
int sage_merge_attn_states_multi_ex(const void* const* o_blks, const float* const* lse_blks, int count, int o_dtype,
                                    void* o_out, float* lse_out, int64_t rows, int D, float lse_in_mult,
                                    const float* corr, float corr_mult, sage_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SAGEATTN_HIP_H */
