/*
 * fa_mi355.h -- C ABI of the MI355X (gfx950) FlashAttention-forward library.
 *
 * This is the drop-in boundary for the reference's forward path.  Every entry point
 * names the reference interface it replaces (paths relative to the reference repo
 * santiweide/flash-attention-impls @ 2025-10-31):
 *
 *   fa_fwd                 <- _FlashAttnFn.forward + _fwd_kernel launch
 *                             code/triton_fa2/FA2-triton.py:175-205 (tensors + strides in,
 *                             O / m,l out; here LSE = m + ln l replaces the m,l pair)
 *   fa_fwd_dispatch        <- flash_attention_cutlass_dispatch(Q,K,V,O,B,H,N,D,stream)
 *                             code/cutlass_cuda_fa1/run/flash_attn_cutlass.cu:519-544
 *                             (same argument order; contiguous [B,H,N,D]; int status
 *                             instead of void + fprintf)
 *   fa_supported           <- the head_dim switch of that dispatcher (:530-543) and the
 *                             asserts of FA2-triton.py:176-178
 *   fa_last_error          <- fprintf(stderr, ...) at flash_attn_cutlass.cu:510-514,540-542
 *   fa_bwd                 <- _FlashAttnFn.backward + _bwd_kernel launch
 *                             code/triton_fa2/FA2-triton.py:207-237 (q,k,v, the forward's statistics and dO in;
 *                             dQ,dK,dV out -- written once each, no zero-fill and no atomics needed)
 *   fa_fwd_ex, fa_fwd_fp8_ex, fa_bwd_ex
 *                          <- no reference counterpart (its operator takes B, H, N, D from q for all three tensors,
 *                             FA2-triton.py:176,185-194): the same three entry points for K / V with fewer heads than Q
 *                             (grouped-query / multi-query attention) and / or another length
 *
 * Conventions
 *   - all tensor pointers are DEVICE pointers owned by the caller; the library never
 *     allocates, frees or synchronises (flash_attn_cutlass.cu has the same ownership rule);
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return value 0 = success, negative = error (see FA_ERR_*); a human-readable message
 *     for the calling thread is available from fa_last_error();
 *   - re-entrant: no mutable global state besides the one-time, per-device kernel attribute setup (a process may drive
 *     several GPUs: the current device at the time of the call is the one used).
 */
#ifndef FA_MI355_H
#define FA_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FA_VERSION 138          /* 0.1.38: wide-head backward images stored in 16-byte pieces (results unchanged); 0.1.37: wide-head forward on 8 waves x 16 rows (results unchanged, +17 .. 28 %); 0.1.36: backward for head_dim 144 .. 256 (fa_bwd_wide_ds + the caller's three GEMMs); 0.1.35: fp8 forward pass end of the 16-bit kernel (one barrier, continuous ring across a causal pair; results unchanged); 0.1.34: fp8 forward without LSE checks the whole row (sampled bound), fa_build_is_default; 0.1.33: dS hand-off backward (fa_bwd_ds_workspace_bytes); 0.1.32: fa_diag_mfma_loop, fa_device_cus; 0.1.31: head_dim 144 .. 256 forward (16-bit types); 0.1.3: fp8 P V on fp8 MFMAs, fa_fp8_pv_native (0.1.2: + extended entry points (H_kv, S_k); 0.1.1: + backward) */

/* element types of Q/K/V (and of O unless stated otherwise) */
#define FA_DTYPE_BF16     0
#define FA_DTYPE_FP16     1
#define FA_DTYPE_FP8_E4M3 2     /* OCP e4m3fn inputs, bf16 output: through fa_fwd_fp8 (needs a workspace) */

#define FA_OK               0
#define FA_ERR_BAD_DTYPE   -1
#define FA_ERR_BAD_HEAD_DIM -2
#define FA_ERR_BAD_SHAPE   -3
#define FA_ERR_BAD_STRIDE  -4
#define FA_ERR_NULL_PTR    -5
#define FA_ERR_LAUNCH      -6
#define FA_ERR_TOO_LARGE   -7

/* Library version (FA_VERSION of the build). */
int fa_version(void);

/* 1 if (dtype, head_dim) is served: dtype as below, head_dim % 16 == 0 and 16 <= head_dim <= 128 (what the reference
 * accepts, FA2-triton.py:178; head_dim <= 64 runs on the head_dim-64 kernel, larger on the head_dim-128 kernel), and --
 * beyond the reference, forward only, bf16 / fp16 -- 144 <= head_dim <= 256 (a plain 128-row kernel; the backward entry
 * points reject head_dim > 128). */
int fa_supported(int dtype, int head_dim);

/* Message describing the last error on the calling thread ("" if none). */
const char* fa_last_error(void);

/*
 * Fused attention forward:  O = softmax(scale * Q K^T  [+ causal mask]) V
 *
 *   q, k, v : [B, H, S, D] with element strides (stride_b, stride_h, stride_s); the
 *             head_dim stride must be 1.  A NULL strides pointer means contiguous.
 *             Base pointers and row strides must be 16-byte aligned.
 *   o       : same logical shape; element type = dtype (bf16 for FA_DTYPE_FP8_E4M3).
 *   lse     : nullable; contiguous [B, H, S] fp32, natural log-sum-exp of the scaled scores.
 *   causal  : non-zero = mask keys j > query i (top-left aligned, FA2-triton.py:70-73).
 *   softmax_scale : <= 0 selects 1/sqrt(D) (FA2-triton.py:183).
 *   descale : nullable, HOST pointer to 3 floats {q,k,v} per-tensor scales: scores are multiplied by
 *             descale[0]*descale[1], the output by descale[2] (NULL = 1.0).
 *   dtype FA_DTYPE_FP8_E4M3 is rejected here (FA_ERR_BAD_DTYPE): use fa_fwd_fp8.
 */
int fa_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
           int B, int H, int S, int D,
           const int64_t* q_strides, const int64_t* k_strides,
           const int64_t* v_strides, const int64_t* o_strides,
           int dtype, int causal, float softmax_scale,
           const float* descale, void* stream);

/*
 * fp8 (OCP e4m3fn) Q/K/V, bf16 O.  head_dim > 64: all three tensors are consumed as fp8 by the matrix cores -- both
 * Q K^T and P V run on v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (the whole head_dim, resp. 128 keys, per
 * instruction; P is rounded to e4m3 after the fp32 softmax, the row sums and the LSE use the unrounded P) -- and no
 * workspace is needed.  head_dim <= 64: all three tensors are converted (exactly) to bf16 into the caller-provided
 * workspace by a HIP pre-pass and the bf16 kernel runs.  The q, k dequantisation scales are folded into the softmax
 * scale, the v scale into the output.  No reference counterpart (the reference is fp16 only); BASELINE.json
 * config 5.  Stated accuracy for fp8 inputs: relative Frobenius error <= 5 %.
 *   Where P sits in e4m3's range is chosen per row from the scores of its FIRST 128 KEYS (reference = their maximum; window
 *   opened downwards by how far the rest of the sample lies below it).  No finite sample can see a tail that lies under the
 *   window behind a first block that is dominant as a whole, so every variant checks the WHOLE row and redoes a workgroup
 *   whose rows lost weight with its exact running-maximum loop (results then as accurate as on ordinary data):
 *     lse != NULL: the exact fp32 row sums exist anyway; a rounded row sum under 0.92 of the exact one triggers the redo;
 *     lse == NULL: (the default of the Python entry) behind the first block one score in 32 adds min(P, 2^-9) to a sampled
 *                  bound on the weight below e4m3's smallest subnormal; an estimate above 1/64 of the row sum triggers the
 *                  redo (ordinary rows stay under 1/200; a heavy tail under the window IS that weight, and must be > 1600
 *                  keys long to matter).  Cost: < 1 % of the kernel.
 *   The stated 5 % bound holds for both (tests/test_cfg5_gpu.py: first block dominant as a whole at S = 4096, 32768, 65536
 *   with 8 - 55 % of each row's weight in the tail; tests/golden/fp8_d128_block0_dominant_*.npz from the reference's own
 *   sdpa_reference).
 *   strides are in elements (= bytes) with unit head_dim stride; rows and bases 16-byte aligned.
 *   workspace: device buffer of at least fa_fp8_workspace_bytes(B,H,S,D) bytes, 16-byte aligned (0 bytes for
 *              head_dim > 64: the pointer may then be NULL; three bf16 tensors otherwise).
 * fa_fp8_pv_native(): 1 if P V of the fp8 entry runs on fp8 MFMAs (this build's default), 0 if on bf16 MFMAs.
 */
size_t fa_fp8_workspace_bytes(int B, int H, int S, int D);
int fa_fp8_pv_native(void);
int fa_fwd_fp8(const void* q, const void* k, const void* v, void* o, float* lse,
               int B, int H, int S, int D,
               const int64_t* q_strides, const int64_t* k_strides,
               const int64_t* v_strides, const int64_t* o_strides,
               int causal, float softmax_scale, const float* descale,
               void* workspace, size_t workspace_bytes, void* stream);

/*
 * Diagnostics (no reference counterpart; the reference's protocol asks for "% of peak", code/README.md:41-43, and this is
 * the measured ceiling that fraction can be read against):
 *   fa_device_cus()     compute units of the current device as the launch heuristics see them (256 on a whole MI355X).
 *   fa_diag_mfma_loop() enqueues ONE launch of a kernel that does nothing but back-to-back dense MFMAs of the given element
 *                       type (FA_DTYPE_BF16 / FA_DTYPE_FP16: v_mfma_f32_16x16x32; FA_DTYPE_FP8_E4M3:
 *                       v_mfma_scale_f32_16x16x128_f8f6f4 with unit scales) on every CU, two waves per SIMD, 64 x iters
 *                       MFMAs per wave, operand fragments taken from `operands` (>= 64 KiB of device memory holding data of
 *                       that type: the matrix cores' power draw depends on it) and kept in registers.  *flops_out = FLOPs of
 *                       the launch; `sink` = one device float (never written in practice).  Timed by the caller
 *                       (bench.py: `roofline.attainable`).
 */
int fa_device_cus(void);
/* 1 if this library was compiled from the sources with no experiment / ablation / diagnostic switch set (csrc/fa_build_guard.hpp). */
int fa_build_is_default(void);
int fa_diag_mfma_loop(int dtype, int iters, const void* operands, float* sink, double* flops_out, void* stream);

/*
 * Contiguous [B,H,N,D] convenience entry with the reference dispatcher's argument order
 * (flash_attn_cutlass.cu:519-529), non-causal, scale 1/sqrt(D).
 */
int fa_fwd_dispatch(const void* Q, const void* K, const void* V, void* O,
                    int batch_size, int num_heads, int seq_len, int head_dim,
                    int dtype, void* stream);

/*
 * Attention backward:  given dO, the forward's inputs, its output O and its LSE, writes
 *   dV = P^T dO,  dQ = scale * dS K,  dK = scale * dS^T Q,   P = exp(scale*Q K^T [masked] - LSE),
 *   dS = P o (dO V^T - delta),  delta = rowsum(dO o O)
 * (the recompute backward of FA2-triton.py:98-170 with the softmax Jacobian in its correct form; the
 * reference's :160-161 is defective and is not reproduced -- see DESIGN.md §2).
 *   q,k,v,o,d_o : [B,H,S,D] inputs of element type dtype (FA_DTYPE_BF16 / FA_DTYPE_FP16), strides as for fa_fwd
 *   lse         : contiguous [B,H,S] fp32, as written by fa_fwd (natural log-sum-exp of the scaled scores)
 *   dq,dk,dv    : [B,H,S,D] outputs of element type dtype; every element is written (no need to zero them)
 *   causal, softmax_scale : must equal the forward call's
 *   workspace   : device buffer of at least fa_bwd_workspace_bytes(B,H,S) bytes, 16-byte aligned
 * Three launches on `stream`: row statistics pre-pass, dQ kernel, dK/dV kernel -- or, with the larger workspace of
 * fa_bwd_ds_workspace_bytes (below): pre-pass, dK/dV kernel (which also writes dS), dQ GEMM.  Deterministic (no atomics).
 */
size_t fa_bwd_workspace_bytes(int B, int H, int S);
int fa_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
           void* dq, void* dk, void* dv,
           int B, int H, int S, int D,
           const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
           const int64_t* o_strides, const int64_t* do_strides,
           const int64_t* dq_strides, const int64_t* dk_strides, const int64_t* dv_strides,
           int dtype, int causal, float softmax_scale,
           void* workspace, size_t workspace_bytes, void* stream);

/*
 * Extended entry points: key/value head count and key count of their own.
 *   H_kv : Q, O, dO, dQ and LSE have H heads, K, V, dK, dV have H_kv heads, H % H_kv == 0; query head h attends to
 *          key/value head h / (H / H_kv) (grouped-query / multi-query attention: consecutive query heads share one).
 *   S_k  : Q, O, dO, dQ, LSE have S_q rows, K, V, dK, dV have S_k rows (S_k >= 1).  The causal mask is then
 *          bottom-right aligned -- key j is visible to query i iff j <= i + S_k - S_q (the last query sees every key:
 *          chunked prefill / decoding against a longer key/value history).  For S_k < S_q the first S_q - S_k queries
 *          see no key: their O rows are 0, their LSE -inf, and they contribute no gradient.
 * Everything else is as in the entry point of the same name; the strides of K, V, dK, dV address [B, H_kv, S_k, D]
 * tensors (NULL = contiguous).  dK and dV are the sums over the query heads of the group, accumulated in registers by
 * one workgroup per (key block, key/value head) -- or, for small launches, by a few of them whose fp32 partial sums a
 * reduction pass adds in a fixed order: no atomics, no zero-fill, run-to-run deterministic either way.
 * H_kv == H and S_k == S_q is exactly fa_fwd / fa_fwd_fp8 / fa_bwd.
 * Workspaces: fa_fp8_workspace_bytes(B, H, max(S_q, S_k), D) and fa_bwd_workspace_bytes(B, H, S_q) are sufficient.
 * fa_bwd_ex_workspace_bytes is the recommended size for fa_bwd_ex (and fa_bwd): where a launch would have too few dK/dV
 * workgroups for the chip -- few key/value heads and a small batch, multi-query attention above all; or fewer workgroups
 * than CUs -- it adds room for fp32 partial sums, so that the query heads of a group (or the query range of a key block)
 * can be split over several workgroups and summed by a reduction pass (same results up to fp32 summation order); with the
 * smaller fa_bwd_workspace_bytes the unsplit kernel runs.
 */
size_t fa_bwd_ex_workspace_bytes(int B, int H, int H_kv, int S_q, int S_k, int D);
/*
 * The dS hand-off backward (no counterpart in the reference, whose _bwd_kernel recomputes: FA2-triton.py:98-170).
 * A workspace of fa_bwd_ds_workspace_bytes(...) -- fa_bwd_ex_workspace_bytes plus 2 * S_q * S_k bytes per query head,
 * padded to 64-key x 256-query tiles (cfg3: 8 GiB) -- lets fa_bwd_ex / fa_bwd run 5 matrix products instead of 7: the
 * dK/dV kernel writes the 16-bit dS it forms to the workspace and dQ = scale * dS . K becomes one streaming GEMM over it
 * (flash_attention_impls_amd/csrc/fa_bwd_dq_gemm_kernel.hpp), instead of a second kernel that recomputes S, P, dP and dS.
 * HBM capacity and bandwidth for matrix work: cfg3 causal backward 3.67 -> 3.26 ms (profiles/r3_bwd_handoff_sweep.txt).
 * dQ, dK and dV are bitwise those of the recompute path (both form the same 16-bit dS and sum dS . K in the same key order;
 * asserted on a shape grid, on randomly drawn extended shapes and on cfg3 at full size: tests/test_bwd_ds_gpu.py).
 * Deterministic, no atomics.  One head's image may be of any size (both kernels address it through descriptors of two slab
 * rows); a caller short of memory splits the call over batches or over groups of query heads and reuses one workspace.
 * Returns 0 where the hand-off cannot pay (head_dim <= 64 with a dS image of more than 256 MiB: the bytes moved do not
 * shrink with the head_dim, the work saved does; fa_bwd_ex, which knows the mask, takes the hand-off at head_dim <= 64 only
 * while the dS actually written and read -- half the image under the causal mask -- is at most 128 MiB, i.e. stays in the
 * Infinity Cache: profiles/r3_bwd_handoff_sweep_d64.txt); with any smaller workspace, or with FA_MI355_BWD_DS=0 in the
 * environment, the recompute path runs.
 */
size_t fa_bwd_ds_workspace_bytes(int B, int H, int H_kv, int S_q, int S_k, int D);
int fa_fwd_ex(const void* q, const void* k, const void* v, void* o, float* lse,
              int B, int H, int H_kv, int S_q, int S_k, int D,
              const int64_t* q_strides, const int64_t* k_strides,
              const int64_t* v_strides, const int64_t* o_strides,
              int dtype, int causal, float softmax_scale,
              const float* descale, void* stream);
int fa_fwd_fp8_ex(const void* q, const void* k, const void* v, void* o, float* lse,
                  int B, int H, int H_kv, int S_q, int S_k, int D,
                  const int64_t* q_strides, const int64_t* k_strides,
                  const int64_t* v_strides, const int64_t* o_strides,
                  int causal, float softmax_scale, const float* descale,
                  void* workspace, size_t workspace_bytes, void* stream);
int fa_bwd_ex(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
              void* dq, void* dk, void* dv,
              int B, int H, int H_kv, int S_q, int S_k, int D,
              const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
              const int64_t* o_strides, const int64_t* do_strides,
              const int64_t* dq_strides, const int64_t* dk_strides, const int64_t* dv_strides,
              int dtype, int causal, float softmax_scale,
              void* workspace, size_t workspace_bytes, void* stream);

/*
 * Backward for wide heads, head_dim 144 .. 256 (this library's extension: the reference stops at 128, FA2-triton.py:178;
 * SURVEY.md section 8(f) row N2 names 256).  At these head dims the fused backward kernels do not fit a register file, so the
 * work is split where the head_dim stops mattering: fa_bwd_wide_ds runs the row-statistics pre-pass and ONE kernel
 * (flash_attention_impls_amd/csrc/fa_bwd_wide_kernel.hpp) that forms S = Q K^T and dP = dO V^T on the matrix cores and writes
 *   p_image [B,H,S_q,ld]  = P  = exp(scale * S [masked] - LSE)               (element type dtype, zeros where the mask hides a key)
 *   ds_image[B,H,S_q,ld]  = scale * P o (dP - delta)
 * contiguous, row stride ld (a multiple of 8, >= S_k; columns S_k .. ld-1 are written as zeros); every element is written.
 * What remains are three plain GEMMs over the images, the caller's (any BLAS; this package's Python layer calls the vendor's batched GEMM):
 *   dV = P^T dO,   dK = dS^T Q,   dQ = dS K        (per key/value head: summed over its group's query heads)
 * A caller short of memory splits the call over batches or heads and reuses the images (2 * S_q * ld * 2 bytes per query head).
 * workspace: fa_bwd_wide_workspace_bytes(B, H, S_q) bytes, 16-byte aligned.  Deterministic, no atomics.
 */
size_t fa_bwd_wide_workspace_bytes(int B, int H, int S_q);
int fa_bwd_wide_ds(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                   void* p_image, void* ds_image, long long ld,
                   int B, int H, int H_kv, int S_q, int S_k, int D,
                   const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                   const int64_t* o_strides, const int64_t* do_strides,
                   int dtype, int causal, float softmax_scale,
                   void* workspace, size_t workspace_bytes, void* stream);

/*
 * Launch geometry that fa_fwd would use (for benches / profilers): writes grid size,
 * block size and dynamic LDS bytes.  Returns FA_OK or an error.
 */
int fa_fwd_launch_info(int B, int H, int S, int D, int dtype, int causal,
                       int* grid, int* block, int* lds_bytes);

#ifdef __cplusplus
}
#endif
#endif /* FA_MI355_H */
