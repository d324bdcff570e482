// Internal launcher declarations shared by the .hip translation units of libconcepthash_hip.
#pragma once
#include "ch_common.h"

// ---- gemm_bf16.hip -------------------------------------------------------------------------------------------
enum GemmEpilogue {
    EPI_BIAS = 0,           // out_bf16 = acc + bias
    EPI_BIAS_QUICKGELU = 1, // out_bf16 = quick_gelu(acc + bias)
    EPI_BIAS_GELU = 2,      // out_bf16 = gelu_erf(acc + bias)
    EPI_BIAS_RESID = 3,     // v = acc + bias; resid += v; out_bf16 = v
    EPI_SCALE_RESID = 4,    // resid += [addend] + *scale_ptr * (acc + bias)
    EPI_PATCH = 5,          // resid[token row of patch m] = acc + pos[1 + patch]
    // ---- LayerNorm folded into the CONSUMER GEMM: the producer of a LayerNorm input emits per-row
    // partial (sum, sum of squares) over each 64-column slice of its bf16-rounded output; the consumer multiplies the RAW
    // rows by W' = bf16(W * gamma) and normalises in its epilogue:  y = rstd_m * (acc - mean_m * c_n) + d_n,
    // c_n = sum_k W'[n][k], d_n = bias_n + sum_k W[n][k] * beta_k.
    EPI_BIAS_STATS = 6,         // EPI_BIAS + row statistics of out_bf16 -> stats_out
    EPI_SCALE_RESID_STATS = 7,  // EPI_SCALE_RESID + hb_out = bf16(resid) + row statistics of hb_out -> stats_out
    EPI_FOLD_BIAS = 8,          // out_bf16 = LN-folded linear (bias = d, fold_c = c, stats_in)
    EPI_FOLD_QUICKGELU = 9,     // ... then quick_gelu
    EPI_FOLD_GELU = 10,         // ... then gelu_erf
    // ---- training step (train.hip): activations applied / differentiated on a bf16 PRE-activation tensor in the epilogue
    EPI_BIAS_DACT_QUICK = 11,   // out_bf16 = bf16(acc + bias) * quick_gelu'(aux)                     (fc2 input-gradient GEMM)
    EPI_BIAS_DACT_GELU = 12,    // out_bf16 = *scale_ptr * bf16(acc + bias) * gelu_erf'(aux)          (adapter up-projection dgrad)
    EPI_FOLD_ACT2_QUICK = 13,   // out_bf16 = pre = LN-folded linear; hb_out = quick_gelu(pre)        (fc1 forward, pre kept for backward)
    EPI_FOLD_ACT2_GELU = 14,    // ... hb_out = gelu_erf(pre)                                          (adapter down-projection forward)
    EPI_COUNT = 15,
};

struct GemmParams {
    const bf16_t *X;       // [X_rows_alloc, K]
    const bf16_t *W;       // [N, K]
    int M, N, K;
    int64_t X_rows_alloc;  // rows actually allocated behind X (>= M rounded up to the block tile)
    const float *bias;     // [N] or nullptr
    bf16_t *out_bf16;      // [M, ldo]
    int ldo;
    float *resid;          // fp32 residual stream [rows, ldr]
    int ldr;
    const float *scale_ptr;  // device scalar (adapter scale)
    const bf16_t *addend;    // EPI_SCALE_RESID: optional bf16 [M, ld_addend] added to the residual as well (or nullptr)
    int ld_addend;
    int group_n;             // n-tiles per L2-resident weight group (set by the launcher)
    const float *pos;        // EPI_PATCH: position embedding [1 + Np, N]
    int tokens_per_img;      // EPI_PATCH: N tokens per image in the residual stream
    int patches_per_img;     // EPI_PATCH: Np
    // LayerNorm fold (EPI_*_STATS producers, EPI_FOLD_* consumers)
    float *stats_out;        // [rows, N/64, 2] partial (sum, sumsq) per 64-column slice of this GEMM's output row
    bf16_t *hb_out;          // EPI_SCALE_RESID_STATS: bf16 copy of the updated residual [rows, ld_hb]
    int ld_hb;
    const float *stats_in;   // [rows, K/64, 2] partials of this GEMM's input rows (written by the producer)
    const float *fold_c;     // [N]
    float ln_eps;
    const bf16_t *aux;       // EPI_BIAS_DACT_*: saved pre-activation [M, ldo] (same layout as out_bf16)
    // split-K tail of the 256x256 kernel (gemm_pp.hip): fp32 slabs [<= 256 units][256*256] + one counter per split tile;
    // nullptr = every tile is computed by one workgroup.  split_full / split_s are set by the launcher.
    float *splitk_ws;
    unsigned *splitk_cnt;
    int split_full, split_s;
    int force_split;         // debug taps: split even though the model's "splitk" option is off
    int rev;                 // 1: walk the m-tiles from the last row tile to the first (serpentine launch order)
    int pp_sched;            // 256x256 kernel: 0 = four phases of 16 MFMAs per K-tile, 1 = two phases of 32 (gemm_pp.hip)
    int pp_min_k;            // dispatcher: smallest K that goes to the 256x256 ping-pong kernel (0 = default 512)
    int64_t footprint_rows;  // rows of the same tensor that CONCURRENT launches (the other micro-batch chains) cover, this launch included; 0 = M.
                             // What the cache-policy choice below is sized on: two chains of 128 images put the tensors of 256 images in flight
    int nt_resid;            // 1 = launch the instance with the non-temporal read-modify-write of the fp32 residual (set by the dispatcher)
    int nt_out;              // 1 = store out_bf16 non-temporally (an output larger than the Infinity Cache that is read once, much later)
    int tag;                 // profiling only: 1 = launch the 256x256 kernel under its second symbol name (gemm_pp.hip, TAG)
    int small_kernel;        // dispatcher, GEMMs that do not go to the 256x256 kernel: 0 = default, 1 = 128x128x64 two-phase, 2 = 128x128x32 ring at every size
    // per-model options handed down by the launch chain (ch_model_set_option); zero = the dispatcher's own choice
    int nt_resid_opt;        // 0 = by tensor size, 1 = non-temporal residual instance, -1 = default-policy instance
    int nt_out_opt;          // the same for the non-temporal bf16 output instance
    int group_n_opt;         // > 0: n-tiles per weight group instead of the ch_gemm_group_n heuristic
    int splitk_opt;          // 1 = split-K tail of the 256x256 kernel (needs splitk_ws / splitk_cnt)
    int rows_opt;            // 1 = whole-row kernel for N = 384 where supported (experiments build)
    int wide_opt;            // 256x384 kernel for N % 384 == 0 (experiments build): 0 = off (default: measured neutral end to end), 1 = wherever supported
};
constexpr size_t CH_SPLITK_WS_BYTES = (size_t)256 * 256 * 256 * 4;  // 64 MiB: at most 256 tail units of one 256x256 fp32 slab
constexpr size_t CH_SPLITK_CNT_BYTES = 256 * sizeof(unsigned);
constexpr int CH_FOLD_LDS_BYTES = 2048;  // per-row (mean, rstd) table of a <= 256-row block tile
int ch_gemm_bf16(const GemmParams &p, int epi, hipStream_t s);      // dispatcher
int ch_gemm_bf16_v1(const GemmParams &p, int epi, hipStream_t s);   // gemm_bf16.hip: 128x128x64, two-phase
int ch_gemm_bf16_pp(const GemmParams &p, int epi, hipStream_t s);   // gemm_pp.hip: 256x256x64, ping-pong 8-phase
bool ch_gemm_pp_supported(const GemmParams &p);
int ch_gemm_bf16_r4(const GemmParams &p, int epi, hipStream_t s);   // gemm_r4.hip: 128x128x32, 4-stage ring, 2 workgroups/CU (small grids)
bool ch_gemm_r4_supported(const GemmParams &p, int epi);
constexpr int64_t CH_RING_MAX_ROWS = 8192;   // up to this many rows the GEMMs of the 128x128 path run the ring kernel (latency-bound grids)

// Experiment kernels that did not beat the dispatched ones: built only with CH_BUILD_EXPERIMENTS=1
// (-DCH_EXPERIMENTS); the product library does not contain them and their taps say so.
#ifdef CH_EXPERIMENTS
int ch_gemm_bf16_pq(const GemmParams &p, int epi, hipStream_t s);   // gemm_pq.hip: 256x128x64, ping-pong, two phases per K-tile
bool ch_gemm_pq_supported(const GemmParams &p);
int ch_gemm_bf16_ppp(const GemmParams &p, int epi, hipStream_t s);  // gemm_ppp.hip: persistent ping-pong (bf16-output epilogues)
bool ch_gemm_ppp_supported(const GemmParams &p, int epi);
int ch_gemm_bf16_dp(const GemmParams &p, int epi, hipStream_t s);   // gemm_dp.hip: 256x128x32, 3-stage ring, 2 workgroups/CU
bool ch_gemm_dp_supported(const GemmParams &p);
int ch_gemm_bf16_rows(const GemmParams &p, int epi, hipStream_t s); // gemm_rows.hip: 128 whole rows x N = 384 per workgroup (adapter down-projection)
bool ch_gemm_rows_supported(const GemmParams &p, int epi);
int ch_gemm_bf16_wide(const GemmParams &p, int epi, hipStream_t s); // gemm_wide.hip: 256x384x32, three-stage ring, two-phase ping-pong (N = 384: adapter bottleneck)
bool ch_gemm_wide_supported(const GemmParams &p, int epi);
int ch_gemm_bf16_wide_dbg(const GemmParams &p, int dbg, hipStream_t s);  // timing-only builds (garbage results)
#else
static inline int ch_experiments_not_built() {
    ch_set_error("experiment kernels are not part of this build (rebuild with CH_BUILD_EXPERIMENTS=1)");
    return 2;
}
static inline int ch_gemm_bf16_pq(const GemmParams &, int, hipStream_t) { return ch_experiments_not_built(); }
static inline int ch_gemm_bf16_ppp(const GemmParams &, int, hipStream_t) { return ch_experiments_not_built(); }
static inline int ch_gemm_bf16_dp(const GemmParams &, int, hipStream_t) { return ch_experiments_not_built(); }
static inline int ch_gemm_bf16_rows(const GemmParams &, int, hipStream_t) { return ch_experiments_not_built(); }
static inline bool ch_gemm_rows_supported(const GemmParams &, int) { return false; }
static inline int ch_gemm_bf16_wide(const GemmParams &, int, hipStream_t) { return ch_experiments_not_built(); }
static inline bool ch_gemm_wide_supported(const GemmParams &, int) { return false; }
static inline int ch_gemm_bf16_wide_dbg(const GemmParams &, int, hipStream_t) { return ch_experiments_not_built(); }
#endif
int ch_gemm_bf16_pp_dbg(const GemmParams &p, int dbg, hipStream_t s);  // timing-only builds (garbage results)
void ch_gemm_count_nt_launch(int kind);  // test tap counters: 0 = non-temporal residual instance, 1 = non-temporal output instance
void ch_gemm_set_variant(int v);
// n-tiles per weight group for a block tile of bn columns: minimises X re-fetches + W re-fetches
int ch_gemm_group_n(int M, int N, int K, int bm, int bn, int forced = 0);

// ---- adapter_fused.hip --------------------------------------------------------------------------------------------
struct AdapterParams {
    const bf16_t *A;      // sub-block output a [M, D] bf16 (read twice: statistics + MFMA operand, and as the residual addend)
    float *H;             // residual stream [M, D] fp32, updated in place
    int M, D, bpad;
    const bf16_t *Wd;     // [bpad, D]  bf16(Wd * gamma), zero rows past b
    const float *c, *d;   // [bpad]     LayerNorm fold constants (see adapter_fused.hip)
    const bf16_t *Wu;     // [D, bpad]  up_proj weight, zero columns past b
    const float *bu;      // [D]
    const float *scale;   // device scalar
    float eps;
    int dbg;              // timing-only ablations (bit 1: skip phase 0, 2: skip phase A loop, 4: skip phase C loop, 8: skip epilogues)
};
#ifdef CH_EXPERIMENTS
bool ch_adapter_fused_supported(int D, int bpad);
int ch_adapter_fused(const AdapterParams &p, hipStream_t s);
#else
static inline bool ch_adapter_fused_supported(int, int) { return false; }
static inline int ch_adapter_fused(const AdapterParams &, hipStream_t) { return ch_experiments_not_built(); }
#endif
// small_f32.hip: LayerNorm fold of a Linear at model build
int ch_fold_ln(const float *Wd, const float *bd, const float *gamma, const float *beta, int b, int bpad, int D, bf16_t *Wdf,
               float *c, float *d, hipStream_t s);

// ---- rowops.hip ----------------------------------------------------------------------------------------------
// im2col for the patch-embed conv (k = s = patch, no bias): out[b*Np + p][c*pp + ky*patch + kx], zero padded to Kp.
int ch_im2col(const void *images, int image_dtype, int B, int image, int patch, int Kp, bf16_t *out, hipStream_t s);
// rows of H: token 0 <- cls + pos[0]; tokens > Np <- ctx[token - Np - 1]; then pre-LN (in place, fp32) and, fused,
// the first encoder layer's LN1 -> xn (bf16).
int ch_assemble_preln(float *H, int B, int ntok, int np, int D, const float *cls_pos0, const float *ctx,
                      const float *pre_w, const float *pre_b, const float *ln_w, const float *ln_b, float eps,
                      bf16_t *xn, hipStream_t s);
// LayerNorm over D of fp32 rows -> bf16
int ch_layernorm_f32(const float *x, int64_t rows, int D, const float *w, const float *b, float eps, bf16_t *out,
                     hipStream_t s);
// LayerNorm over D of bf16 rows -> bf16
int ch_layernorm_bf16(const bf16_t *x, int64_t rows, int D, const float *w, const float *b, float eps, bf16_t *out,
                      hipStream_t s);

// compact copy of the residual rows the hashing head reads: out[b*(1+ncon) + j] = H[b*ntok + (j == 0 ? 0 : ntok-ncon+j-1)]
int ch_gather_head_rows(const float *H, int B, int ntok, int ncon, int D, float *out, hipStream_t s);

// ---- attention.hip -------------------------------------------------------------------------------------------
// qkv [B*ntok, 3D] bf16 (q | k | v, head h at columns h*64), out [B*ntok, D] bf16.  head_dim == 64.
// cattn (optional): [B, heads, ncon, ntok - ncon - 1] fp32 softmax rows of the last `ncon` tokens over tokens 1 .. ntok-ncon-1
// compact: queries are only CLS + the last `ncon` (concept) tokens of every image; out is [B * (1 + ncon), heads * 64]
// rev: process the (image, head) pairs from the last image to the first (serpentine launch order)
int ch_attention(const bf16_t *qkv, int B, int ntok, int heads, bf16_t *out, hipStream_t s, float *cattn = nullptr,
                 int ncon = 0, bool compact = false, bool rev = false);

// ---- head.hip ------------------------------------------------------------------------------------------------
struct HeadParams {
    const float *H;  // residual stream [B*ntok, D]
    int B, ntok, D, Q, nbit, C, P;
    const float *hash_pe;     // [Q, D]
    const float *hash_fc;     // [nbit/Q, D]
    const float *bn_scale;    // [nbit]  gamma / sqrt(var + eps)
    const float *bn_shift;    // [nbit]  beta - mean * scale
    const float *center_l2;   // [C, nbit] l2-normalised projected centres
    const float *center_bin;  // [C, nbit] sign(center_l2) / sqrt(nbit)
    const float *concept_pe;  // [Q, D] or nullptr
    const float *concept_cent_l2;  // [C, D] l2-normalised centroids or nullptr
    const float *post_w, *post_b;  // post_layernorm
    const float *vis_proj;         // [P, D]
    float ln_eps;
    float *out_codes;          // [B, nbit]
    uint64_t *out_packed;      // [B, W] or nullptr
    float *out_logits_cont;    // [B, C] or nullptr
    float *out_logits_bin;     // [B, C] or nullptr
    float *out_logits_concept; // [Q, B, C] or nullptr
    float *out_hash_features;  // [B, Q, D] or nullptr
    float *out_image_features; // [B, P] or nullptr
    float *ws_xn;              // workspace [B*Q, D]: l2(hash_features + concept_pe), left operand of the concept logits (head_dense_kernel)
    float *ws_cls;             // workspace [B, D]: post_layernorm(CLS), left operand of the image features
};
int ch_head(const HeadParams &p, hipStream_t s);
int ch_pack_sign_launch(const float *codes, int64_t rows, int nbit, float thr, uint64_t *out, hipStream_t s);

// ---- small_f32.hip (model-load-time constant folding, fp32) -------------------------------------------------------
// y[r][o] = act(sum_i x[r][i] * W[o][i] + b[o]);  act: 0 none, 1 relu
int ch_small_linear(const float *x, int rows, int in_f, const float *W, const float *b, int out_f, int act, float *y,
                    hipStream_t s);
int ch_small_layernorm(const float *x, int rows, int D, const float *w, const float *b, float eps, float *y,
                       hipStream_t s);
// y = a + b (elementwise)
int ch_small_add(const float *a, const float *b, int64_t n, float *y, hipStream_t s);
// single-sequence multi-head attention core on packed qkv [T, 3P] -> out [T, P]
int ch_small_mha(const float *qkv, int T, int P, int heads, float *out, hipStream_t s);
// rows l2-normalise; optionally also sign(x_l2)/sqrt(cols) into out_bin
int ch_small_l2norm(const float *x, int rows, int cols, float *out_l2, float *out_bin, hipStream_t s);
// fp32 -> bf16 with optional zero padding of the inner dimension: in [rows, cols] -> out [rows, cols_pad]
int ch_convert_bf16(const float *x, int64_t rows, int cols, int cols_pad, bf16_t *out, hipStream_t s);
// bn fold: scale = w / sqrt(var + eps), shift = b - mean * scale
int ch_bn_fold(const float *w, const float *b, const float *mean, const float *var, int n, float eps, float *scale,
               float *shift, hipStream_t s);

// ---- training step (train_kernels.hip, attention_bwd.hip; orchestration in train.hip) -----------------------------------------
// fp32 rows -> bf16 copy + per-64-column (sum, sum of squares) partials [rows, D/64, 2] (what the LN-folded GEMMs consume)
int ch_hb_stats(const float *H, int64_t rows, int D, bf16_t *hb, float *stats, hipStream_t s);
// out = act(pre) on a saved bf16 pre-activation; act: 0 quick_gelu, 1 gelu (erf)
int ch_act_fwd(const bf16_t *pre, int64_t n, int act, bf16_t *out, hipStream_t s);
// out = (*scale_ptr or 1) * g * act'(pre)   (out may alias g)
int ch_act_bwd(const bf16_t *g, const bf16_t *pre, int64_t n, int act, const float *scale_ptr, bf16_t *out, hipStream_t s);
// x_hat = (x - mean) * rstd from the row statistics, as bf16
int ch_normalize_bf16(const bf16_t *x, const float *stats, int64_t rows, int D, float eps, bf16_t *out, hipStream_t s);
// LayerNorm backward per row given dyg = dy * gamma: result = dres_in + rstd (dyg - mean(dyg) - x_hat mean(dyg x_hat)) -> dres_out (fp32,
// may alias dres_in, may be null), out_b (bf16, may be null)
// xhat_out (optional): also write x_hat = (x - mean) * rstd as bf16
int ch_ln_bwd(const bf16_t *dyg, const bf16_t *x, const float *stats, int64_t rows, int D, float eps, const float *dres_in,
              float *dres_out, bf16_t *out_b, hipStream_t s, bf16_t *xhat_out = nullptr);
// out[n][k] = sum_m A[m][n] * B[m][k]  (A [rows, N] ld lda, B [rows, K] ld ldb, bf16; out [N, K] fp32); ws: ch_wgrad_ws_floats floats.
// Rows up to the next multiple of 32 are read: they must be allocated; A's are zeroed by the call, B's must be finite.
// chunks_out (optional): leave the chunk slabs in ws unreduced and report how many there are (-> ch_reduce_partials_multi); out unused
int ch_wgrad_tn(bf16_t *A, int lda, const bf16_t *B, int ldb, int64_t rows, int64_t rows_alloc, int N, int K, float *out,
                float *ws, hipStream_t s, int *chunks_out = nullptr);
// out[i] = sum over chunks of partial[c][i], i < 4 * n4, fixed association; up to four such reductions in one launch
struct ChReduceJob {
    const float *partial;
    float *out;
    int nchunks;
    int n4;
};
int ch_reduce_partials_multi(const ChReduceJob *jobs, int njobs, hipStream_t s);
size_t ch_wgrad_ws_floats(int64_t rows, int N, int K);
// out[n] = sum_m A[m][n]; A bf16 (is_f32 = 0) or fp32 (1); ws: ch_colsum_ws_floats(N) floats
int ch_colsum(const void *A, int is_f32, int lda, int64_t rows, int N, float *out, float *ws, hipStream_t s, int *chunks_out = nullptr);
size_t ch_colsum_ws_floats(int N);
// dst[c][r] = bf16(src[r][c] * (colscale ? colscale[c] : 1)); dst [C, ld_dst]
int ch_transpose_f32_to_bf16(const float *src, int R, int C, int ld_src, const float *colscale, bf16_t *dst, int ld_dst, hipStream_t s);
int ch_transpose_bf16(const bf16_t *src, int R, int C, int ld_src, bf16_t *dst, int ld_dst, hipStream_t s);
// gradients of one adapter's parameters from G = dH^T g [D, bpad], cu = colsum(dH), T = dpre^T x_hat [bpad, D], cd = colsum(dpre);
// params / grads: one adapter's block of the arena ([ln_w D][ln_b D][down_w b*D][down_b b][up_w D*b][up_b D][scale 1])
int ch_adapter_grads(const float *G, const float *cu, const float *T, const float *cd, const float *params, int D, int b, int bpad,
                     float *grads, float *ws /* >= 256 floats per adapter */, hipStream_t s, int nad = 1, int64_t stride = 0);
// nad > 1: slot arrays (G / T stride D * bpad, cu D, cd bpad, params / grads `stride`, ws 256), one launch pair for all of them
// bf16 / LayerNorm-folded / transposed working copies of `nad` adapters from the parameter arena (slot arrays, one slot per adapter)
int ch_adapter_refresh(const float *params, int64_t stride, int nad, int D, int b, int bpad, bf16_t *down_wf, float *fold_c, float *fold_d,
                       bf16_t *up_w, bf16_t *up_wT, bf16_t *down_wgT, hipStream_t s);
int ch_sgd_step_launch(float *p, const float *g, float *buf, int64_t n, float lr, float momentum, float wd, float dampening, int nesterov,
                       int first, hipStream_t s);
int ch_concept_rows_sum(const float *dH, int B, int ntok, int Q, int D, float *out, hipStream_t s);
int ch_scatter_concept_rows(const float *dhf, int B, int ntok, int Q, int D, float *dH, bf16_t *dHb, hipStream_t s);
// compact head rows [B*(1+Q), D] (CLS, concept tokens) -> full token rows [B*ntok, D], zeros elsewhere; fp32 (is_f32) or bf16
int ch_expand_head_rows(const void *src, int is_f32, int B, int ntok, int Q, int D, void *dst, hipStream_t s);
int ch_small_ln_bwd(const float *dy, const float *x, const float *gamma, int rows, int D, float eps, float *dx, hipStream_t s);
int ch_gather_concept_rows(const float *H, int B, int ntok, int Q, int D, float *out, hipStream_t s);
// qkv [B*ntok, 3D] (q | k | v), dO [B*ntok, D] -> dqkv [B*ntok, 3D]; head_dim 64
// dpext (optional): [B, heads, ncon, ntok - ncon - 1] fp32 cotangent of the last `ncon` tokens' attention rows over tokens 1 .. ntok-ncon-1
int ch_attention_bwd(const bf16_t *qkv, const bf16_t *dO, int B, int ntok, int heads, bf16_t *dqkv, hipStream_t s,
                     const float *dpext = nullptr, int ncon = 0);
