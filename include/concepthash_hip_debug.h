/*
 * concepthash_hip_debug.h -- test / bench taps of libconcepthash_hip.so (MI355X / gfx950).
 *
 * NOT part of the drop-in boundary (include/concepthash_hip.h): single-kernel launches on caller buffers, kernel-selection
 * overrides and workspace copies that tests/ and tools/ use to pin each kernel by itself.  Same conventions as the main header
 * (0 on success, ch_last_error(), device pointers, `stream` = hipStream_t as void*).
 */
#ifndef CONCEPTHASH_HIP_DEBUG_H
#define CONCEPTHASH_HIP_DEBUG_H

#include "concepthash_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test / bench taps (not part of the product path): one fused-epilogue GEMM launch on caller buffers, and a global
 * override of the GEMM kernel selection (0 auto, 1 = 128x128 two-phase kernel, 2 = 256x256 ping-pong kernel).
 * X [X_rows_alloc, K] bf16, W [N, K] bf16, bias [N] fp32; epi: 0 bias, 1 bias+quick_gelu, 2 bias+gelu,
 * 3 bias + (resid += v) + bf16 out, 4 resid += [addend bf16 [M,N]] + *scale_ptr * (acc + bias). */
int ch_debug_gemm(int32_t variant, const void *X, int64_t X_rows_alloc, const void *W, const float *bias, int32_t M,
                  int32_t N, int32_t K, int32_t epi, void *out_bf16, int32_t ldo, float *resid, int32_t ldr,
                  const float *scale_ptr, const void *addend, void *stream);
/* The LayerNorm-fold epilogues: epi 6 = bias + row statistics of the bf16 output -> stats_out
 * [M, N/64, 2]; 7 = epi 4 + hb_out = bf16(resid) + its row statistics; 8/9/10 = y = rstd*(acc - mean*fold_c) + bias
 * [+ quick_gelu / gelu], mean/rstd from stats_in [M, K/64, 2] (partial sums of the rows of X) and ln_eps. */
int ch_debug_gemm_ln(int32_t variant, const void *X, int64_t X_rows_alloc, const void *W, const float *bias, int32_t M,
                     int32_t N, int32_t K, int32_t epi, void *out_bf16, int32_t ldo, float *resid, int32_t ldr,
                     const float *scale_ptr, const void *addend, const float *stats_in, const float *fold_c, float ln_eps,
                     float *stats_out, void *hb_out, void *stream);
void ch_debug_set_gemm_variant(int32_t variant);
/* 1 when the library was built with CH_BUILD_EXPERIMENTS=1: the non-dispatched experiment kernels (GEMM variants 3 / 5 / 6 of
 * the taps above, the fused adapter kernel behind CH_FUSED_ADAPTER=1 and ch_debug_adapter) exist; 0 in the product build, where
 * those taps return an error. */
int32_t ch_debug_experiments_built(void);
/* How many GEMMs the dispatcher has sent to the 128x128 (which = 0) / 256x256 ping-pong (which = 1) kernel since the
 * library was loaded, and how many launches ran the instance with the non-temporal fp32-residual read-modify-write (which = 2) /
 * the non-temporal bf16 output store (which = 3): lets a parity test prove which kernel produced the output it compared. */
int64_t ch_debug_gemm_dispatch_count(int32_t which);
/* Copy the first nbytes of one activation buffer of the model's workspace, as the last ch_encode / ch_encode_hidden call left
 * it, to `out` (device): which = 0 H fp32 [rows, D] | 1 Xn | 2 QKV [rows, 3D] | 3 AO | 4 A | 5 AD [rows, max(bpad, 128)] |
 * 6 F1 [rows, ffn] (1..6 bf16).  tools/stage_probe.py compares every stage of a layer with the rounding-emulating oracle. */
int ch_debug_copy_buffer(ch_model *m, int32_t which, void *out, int64_t nbytes, void *stream);
/* Let the debug GEMM taps use the split-K tail of the 256x256 kernel (off by default: a split tile sums its K slices in a
 * different order, so it is no longer bit-identical to the 128x128 kernel). */
void ch_debug_set_gemm_splitk(int32_t on);
/* One fused adapter call H += a + scale * (GELU(LN(a) Wd^T + bd) Wu^T + bu) on caller buffers (a [M,D] bf16, H [M,D] fp32,
 * Wd [b,D] fp32, Wu [D, roundup(b,128)] bf16 zero-padded); work_* are caller scratch for the LayerNorm-folded weights. */
int ch_debug_adapter(const void *A, float *H, int32_t M, int32_t D, int32_t b, const float *Wd, const float *bd,
                     const float *gamma, const float *beta, const void *Wu_bf16_padded, const float *bu, const float *scale,
                     void *work_wdf, float *work_c, float *work_d, int32_t dbg, void *stream);
/* qkv [B*ntok, 3*heads*64] bf16 (q | k | v) -> out [B*ntok, heads*64] bf16: softmax(q k^T / 8) v per (image, head). */
int ch_debug_attention(const void *qkv, int32_t B, int32_t ntok, int32_t heads, void *out, void *stream);

/* 1 = the mAP scan passes read the gallery through scalar loads (the round-1 form) instead of 16-row VMEM blocks + DPP row broadcast:
 * same results bit for bit, kept as a cross-check of the row loops (tests/test_hamming_gpu.py).  Process-wide, tests only. */
void ch_debug_set_hamming_scalar_loads(int32_t on);

/* Kernel taps of the training step: see train_kernels.hip / attention_bwd.hip. */
int ch_debug_attention_bwd(const void *qkv, const void *dO, int32_t B, int32_t ntok, int32_t heads, void *dqkv, const float *dpext,
                           int32_t ncon, void *stream);
int ch_debug_wgrad(const void *A, int32_t lda, const void *Bm, int32_t ldb, int64_t rows, int64_t rows_alloc, int32_t N, int32_t K,
                   float *out, void *stream);
int ch_debug_ln_bwd(const void *dyg, const void *x, int64_t rows, int32_t D, float eps, const float *dres_in, float *dres_out,
                    void *out_b, void *xhat_out, void *stream);
int ch_debug_act(const void *g, const void *pre, int64_t n, int32_t act, const float *scale_ptr, int32_t backward, void *out,
                 void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CONCEPTHASH_HIP_DEBUG_H */
