// ch_trainer: the ConceptHash TRAINING step of the adapters -- forward with saved
// activations + backward through the frozen CLIP blocks, on the forward's GEMM / attention kernels plus train_kernels.hip
// and attention_bwd.hip.  Reference: trainers/coop.py:107-131 (train_one_batch: forward, criterion, loss.backward(),
// optimizer.step()), models/layers/adapter.py:46-60 (Adapter), :127-177 (CLIPEncoderLayerWithAdapter.forward),
// trainers/base.py:133-152 (what is trainable: the adapters + get_training_modules(); the backbone is frozen).
//
// Division of labour: this library owns everything that touches the [B*ntok, *] activations -- the 12
// encoder layers forward and backward, > 99.9 % of the step's FLOPs -- and the gradients of the 24 adapters.  The concept-token
// generator (4 tokens), the hashing head on [B, Q, D] and the loss are a few MFLOP; they stay on the host framework's autograd
// (concepthash_amd/training.py), which hands `concept_tokens` in and d(hash_features) back, and receives d(concept_tokens).
//
// Forward = the LN-fold chain of model.hip, one chain, with every layer's operands kept: bf16(H) + row statistics (the
// LayerNorm inputs), qkv, attention output, a / m (sub-block outputs) + statistics, adapter pre-activations and activations,
// fc1 pre-activation.  Pre-activations are saved instead of activations, so the activations are applied by a separate
// epilogue mode in training: the GEMM writes the pre-activation AND its activation (EPI_FOLD_ACT2_*); backward multiplies by the
// activation's derivative in the dgrad GEMM's epilogue (EPI_BIAS_DACT_*).
// Backward per layer, dH = gradient of the residual stream (fp32, bf16 copy dHb as the GEMM operand):
//   adapter:  G = dHb^T g, cu = colsum(dH)                                [weight-gradient products, up]
//             dpre = s (dHb W_up) o gelu'(pre)                            [dgrad GEMM + act_bwd]
//             d(branch input) = dH + LN_bwd(dpre (W_dn o gamma))          [dgrad GEMM with gamma folded into W^T + ln_bwd, which
//                                                                          also emits x_hat of the adapter input as bf16]
//             T = dpre^T x_hat, cd = colsum(dpre)                         [weight-gradient products, down]
//   MLP:      dF = (dM W_fc2) o act'(pre), dh = dF (W_fc1 o gamma2), dH += LN_bwd(dh)
//   attn:     dctx = dA W_o, dqkv = attention_bwd(qkv, dctx), dh = dqkv (W_qkv o gamma1), dH += LN_bwd(dh)
// Parameter arena (fp32, caller-owned, device): adapters in (layer, adapter) order, each
//   [ln_w D][ln_b D][down_w b*D][down_b b][up_w D*b][up_b D][scale 1];  the gradient arena has the same layout.
#include <algorithm>
#include <vector>

#include "model_internal.h"
#include "../../include/concepthash_hip_debug.h"

namespace {
struct AdWork {
    bf16_t *down_wf = nullptr, *up_w = nullptr, *up_wT = nullptr, *down_wgT = nullptr;
    float *fold_c = nullptr, *fold_d = nullptr;
};
struct LayerT {
    bf16_t *qkv_wgT = nullptr, *out_wT = nullptr, *fc1_wgT = nullptr, *fc2_wT = nullptr;
};
struct Saved {
    bf16_t *Xn1, *QKV, *AO, *A, *P1, *G1, *Xn2, *F1pre, *A2, *P2, *G2;
    float *st1, *stA, *st2, *stA2;
};
}  // namespace

constexpr int TR_CHAINS = 2;

struct ch_trainer {
    ch_model *m = nullptr;
    int max_batch = 0, B = 0;
    // Two micro-batch chains on two streams (as ch_encode does), model option "train_chains" = 2: chain 1 owns its own row
    // region of every activation buffer (so that padding rows never alias the other chain's data), its own weight-gradient
    // scratch and its own gradient arena; the two arenas are added once at the end (the assembly is linear in the weight-gradient
    // products).  Correct (tests run both), but MEASURED SLOWER than one chain for training -- 47.2 vs 42.5 ms at batch 256, 27.0
    // vs 23.6 ms at batch 128: the half-size GEMMs lose more than the overlap of the HBM-bound launches returns, and the
    // per-adapter reductions have fixed costs that double -- so one chain is the default.
    int nchains = 1, nc = 1, Bc[TR_CHAINS] = {0, 0};
    int64_t chain_min_rows = 12000;
    int64_t row_off[TR_CHAINS] = {0, 0}, prow_off[TR_CHAINS] = {0, 0}, region_rows[TR_CHAINS] = {0, 0}, region_prows[TR_CHAINS] = {0, 0};
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    float *params = nullptr, *grads = nullptr, *grads1 = nullptr;
    int64_t ad_numel = 0;
    std::vector<void *> allocs;
    size_t bytes = 0;
    std::vector<AdWork> ad;     // [L * 2]
    std::vector<LayerT> lt;     // [L]
    std::vector<Saved> sv;      // [L]
    float *H = nullptr, *dH = nullptr, *ctx = nullptr;
    // final-layer row pruning (as ch_encode): past the last attention only CLS + the Q concept tokens of
    // every image are carried, forward and backward -- the loss reads nothing else, so every other row's gradient is zero there
    bool prune_last = true;
    int64_t hc_rows = 0;
    float *Hc = nullptr, *dHc = nullptr;
    bf16_t *dHcb = nullptr;
    bf16_t *dHb = nullptr, *dMb = nullptr, *tD = nullptr, *tD2 = nullptr, *tB = nullptr, *tM = nullptr, *tQKV = nullptr, *F1act = nullptr,
           *PATCH = nullptr, *XnDummy = nullptr;
    float *stDummy = nullptr;
    float *ws_wgrad[TR_CHAINS] = {}, *ws_colsum[TR_CHAINS] = {}, *G[TR_CHAINS] = {}, *T[TR_CHAINS] = {}, *cu[TR_CHAINS] = {},
          *cd[TR_CHAINS] = {}, *dctx_sum[TR_CHAINS] = {};
    // batched gradient assembly (model option "train_batched_grads"): the four chunk-slab sets of an adapter live side by side and are
    // reduced by ONE launch into that adapter's slot of Gs / Ts / cus / cds; the gradients of all 2 L adapters are assembled by one
    // launch pair at the end of the backward chain
    bool batched_grads = true;
    float *ws_wgradT[TR_CHAINS] = {}, *ws_colsumD[TR_CHAINS] = {}, *Gs[TR_CHAINS] = {}, *Ts[TR_CHAINS] = {}, *cus[TR_CHAINS] = {},
          *cds[TR_CHAINS] = {}, *ws_ag[TR_CHAINS] = {};
    bool forward_done = false;
    bool attn_all_layers = false;   // layout of the concept-attention tap, latched by ch_train_forward for the matching ch_train_backward
};

namespace {

void *talloc(ch_trainer *t, size_t bytes, bool &ok) {
    void *p = nullptr;
    if (bytes == 0) bytes = 16;
    if (!ok) return nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess) {
        ch_set_error("trainer: hipMalloc/hipMemset failed for " + std::to_string(bytes) + " bytes");
        ok = false;
        return nullptr;
    }
    t->allocs.push_back(p);
    t->bytes += bytes;
    return p;
}

int64_t adapter_numel(const ch_model_config &c) {
    const int64_t D = c.dim, b = c.adapter_dim;
    return 2 * D + b * D + b + D * b + D + 1;
}

struct AdPtr {
    float *ln_w, *ln_b, *down_w, *down_b, *up_w, *up_b, *scale;
};
AdPtr ad_ptrs(float *base, const ch_model_config &c) {
    const int64_t D = c.dim, b = c.adapter_dim;
    AdPtr p;
    p.ln_w = base;
    p.ln_b = p.ln_w + D;
    p.down_w = p.ln_b + D;
    p.down_b = p.down_w + b * D;
    p.up_w = p.down_b + b;
    p.up_b = p.up_w + D * b;
    p.scale = p.up_b + D;
    return p;
}

struct GemmCall {
    int N, K;
    const bf16_t *X, *W;
    const float *bias;
    int epi;
    bf16_t *out = nullptr;
    int ldo = 0;
    float *resid = nullptr;
    const float *scale = nullptr;
    const bf16_t *addend = nullptr;
    const float *stats_in = nullptr, *fold_c = nullptr;
    float eps = 0.f;
    float *stats_out = nullptr;
    bf16_t *hb_out = nullptr;
    int ld_hb = 0;              // 0 = D
    const bf16_t *aux = nullptr;
};
int gemm(ch_trainer *t, int chain, int rows, const GemmCall &g, hipStream_t s, int64_t x_rows_alloc = 0) {
    const int D = t->m->cfg.dim;
    GemmParams p{};
    p.X = g.X; p.W = g.W; p.M = rows; p.N = g.N; p.K = g.K; p.X_rows_alloc = x_rows_alloc ? x_rows_alloc : t->region_rows[chain];
    p.bias = g.bias; p.out_bf16 = g.out; p.ldo = g.ldo; p.resid = g.resid; p.ldr = D; p.scale_ptr = g.scale; p.addend = g.addend;
    p.ld_addend = D; p.stats_in = g.stats_in; p.fold_c = g.fold_c; p.ln_eps = g.eps; p.stats_out = g.stats_out; p.hb_out = g.hb_out;
    p.ld_hb = g.ld_hb ? g.ld_hb : D; p.aux = g.aux; p.pp_min_k = t->m->pp_min_k;
    p.nt_resid_opt = t->m->resid_nt; p.nt_out_opt = t->m->nt_out; p.group_n_opt = t->m->group_n; p.wide_opt = t->m->wide_kernel;
    return ch_gemm_bf16(p, g.epi, s);
}

// the row region of chain `ch`: every activation buffer is addressed through these
struct Rows {
    int64_t r0;
    int D, M, bpad;
    template <typename T>
    T *d(T *p) const { return p + r0 * D; }
    template <typename T>
    T *d3(T *p) const { return p + r0 * 3 * D; }
    template <typename T>
    T *m(T *p) const { return p + r0 * M; }
    template <typename T>
    T *b(T *p) const { return p + r0 * bpad; }
    float *st(float *p) const { return p + r0 * (D / 64) * 2; }
};

int forward_chain(ch_trainer *t, int ch, const void *images_all, int image_dtype, int img0, int B, float *out_hf_all, float *out_cls_all,
                  float *out_cattn_all, hipStream_t s) {
    ch_model *m = t->m;
    const ch_model_config &c = m->cfg;
    const int D = c.dim, M = c.ffn, ntok = m->ntok, np = m->np, bpad = m->bpad, Q = c.ncontext, L = c.layers;
    const int rows = B * ntok;
    const Rows R{t->row_off[ch], D, M, bpad};
    float *H = R.d(t->H);   // the residual stream; re-pointed at the compact copy past the last layer's attention
    bf16_t *PATCH = t->PATCH + t->prow_off[ch] * m->Kp;
    const size_t img_elems = (size_t)3 * c.image_size * c.image_size;
    const void *images = (const char *)images_all + (size_t)img0 * img_elems * (image_dtype == 0 ? 4 : 2);
    // ---- embeddings (models/arch/coop.py:452-472): im2col + patch GEMM, CLS / position / concept tokens, pre-LN
    if (int e = ch_im2col(images, image_dtype, B, c.image_size, c.patch, m->Kp, PATCH, s)) return e;
    {
        GemmParams p{};
        p.X = PATCH; p.W = m->patch_w; p.M = B * np; p.N = D; p.K = m->Kp; p.X_rows_alloc = t->region_prows[ch];
        p.resid = H; p.ldr = D; p.pos = m->pos; p.tokens_per_img = ntok; p.patches_per_img = np; p.pp_min_k = m->pp_min_k;
        if (int e = ch_gemm_bf16(p, EPI_PATCH, s)) return e;
    }
    const LayerW &w0 = m->layers[0];
    if (int e = ch_assemble_preln(H, B, ntok, np, D, m->cls_pos0, t->ctx, m->pre_w, m->pre_b, w0.ln1_w, w0.ln1_b, c.ln_eps,
                                  R.d(t->XnDummy), s))
        return e;
    if (int e = ch_hb_stats(H, rows, D, R.d(t->sv[0].Xn1), R.st(t->sv[0].st1), s)) return e;
    const int nq = 1 + Q;
    float *Hc = t->Hc + (size_t)ch * t->hc_rows * D;
    int cur = rows;          // rows carried: all tokens, or B * (1 + Q) past the last layer's attention
    for (int l = 0; l < L; ++l) {
        const LayerW &w = m->layers[l];
        Saved &v = t->sv[l];
        GemmCall g;
        // attention block
        g = GemmCall{3 * D, D, R.d(v.Xn1), w.qkv_wf, w.qkv_d, EPI_FOLD_BIAS};
        g.out = R.d3(v.QKV); g.ldo = 3 * D; g.stats_in = R.st(v.st1); g.fold_c = w.qkv_c; g.eps = c.ln_eps;
        if (int e = gemm(t, ch, rows, g, s)) return e;
        const bool pruned = t->prune_last && l == L - 1;
        // last layer: optionally tap the concept tokens' attention rows over the patch tokens (attn_cache[-1][:, :, -Q:, 1:-Q])
        // (concept_attn_all_layers of the call: every layer's rows, [L, B, heads, Q, Np])
        float *cattn = !out_cattn_all ? nullptr
                       : t->attn_all_layers ? out_cattn_all + ((size_t)l * t->B + img0) * c.heads * Q * np
                       : l == L - 1 ? out_cattn_all + (size_t)img0 * c.heads * Q * np : nullptr;
        if (int e = ch_attention(R.d3(v.QKV), B, ntok, c.heads, R.d(v.AO), s, cattn, Q, pruned)) return e;
        if (pruned) {      // from here on every buffer of this layer holds B * (1 + Q) compact rows (CLS, then the concept tokens)
            cur = B * nq;
            if (int e = ch_gather_head_rows(H, B, ntok, Q, D, Hc, s)) return e;
            H = Hc;
        }
        g = GemmCall{D, D, R.d(v.AO), w.out_w, w.out_b, EPI_BIAS_STATS};
        g.out = R.d(v.A); g.ldo = D; g.stats_out = R.st(v.stA);
        if (int e = gemm(t, ch, cur, g, s)) return e;
        for (int a = 0; a < 2; ++a) {
            const AdWork &aw = t->ad[l * 2 + a];
            const AdPtr ap = ad_ptrs(t->params + (int64_t)(l * 2 + a) * t->ad_numel, c);
            const bf16_t *in = R.d(a == 0 ? v.A : v.A2);
            const float *stin = R.st(a == 0 ? v.stA : v.stA2);
            bf16_t *P = R.b(a == 0 ? v.P1 : v.P2), *G = R.b(a == 0 ? v.G1 : v.G2);
            // pre-activation kept for backward, nn.GELU() (models/layers/adapter.py:36) of it as the second output
            g = GemmCall{bpad, D, in, aw.down_wf, aw.fold_d, EPI_FOLD_ACT2_GELU};
            g.out = P; g.ldo = bpad; g.stats_in = stin; g.fold_c = aw.fold_c; g.eps = 1e-5f; g.hb_out = G; g.ld_hb = bpad;
            if (int e = gemm(t, ch, cur, g, s)) return e;
            g = GemmCall{D, bpad, G, aw.up_w, ap.up_b, EPI_SCALE_RESID_STATS};
            g.resid = H; g.scale = ap.scale; g.addend = in;
            if (a == 0) {
                g.stats_out = R.st(v.st2); g.hb_out = R.d(v.Xn2);
            } else {
                g.stats_out = R.st(l + 1 < L ? t->sv[l + 1].st1 : t->stDummy);
                g.hb_out = R.d(l + 1 < L ? t->sv[l + 1].Xn1 : t->XnDummy);
            }
            if (int e = gemm(t, ch, cur, g, s)) return e;
            if (a == 0) {  // MLP
                g = GemmCall{M, D, R.d(v.Xn2), w.fc1_wf, w.fc1_d, c.act == 0 ? EPI_FOLD_ACT2_QUICK : EPI_FOLD_ACT2_GELU};
                g.out = R.m(v.F1pre); g.ldo = M; g.stats_in = R.st(v.st2); g.fold_c = w.fc1_c; g.eps = c.ln_eps; g.hb_out = R.m(t->F1act); g.ld_hb = M;
                if (int e = gemm(t, ch, cur, g, s)) return e;
                g = GemmCall{D, M, R.m(t->F1act), w.fc2_w, w.fc2_b, EPI_BIAS_STATS};
                g.out = R.d(v.A2); g.ldo = D; g.stats_out = R.st(v.stA2);
                if (int e = gemm(t, ch, cur, g, s)) return e;
            }
        }
    }
    const int tok_out = cur == rows ? ntok : nq;   // row layout of the final residual: all tokens, or (CLS, concept tokens) per image
    if (int e = ch_gather_concept_rows(H, B, tok_out, Q, D, out_hf_all + (size_t)img0 * Q * D, s)) return e;
    if (out_cls_all) {  // CLS rows of the final residual (the pooled branch, models/arch/coop.py:484-499, is not part of the loss)
        CH_CHECK_HIP(hipMemcpy2DAsync(out_cls_all + (size_t)img0 * D, sizeof(float) * D, H, sizeof(float) * (size_t)tok_out * D, sizeof(float) * D, B,
                                      hipMemcpyDeviceToDevice, s));
    }
    return 0;
}

int backward_chain(ch_trainer *t, int ch, const float *dhf_all, const float *dcattn_all, int img0, int B, hipStream_t s) {
    ch_model *m = t->m;
    const ch_model_config &c = m->cfg;
    const int D = c.dim, M = c.ffn, ntok = m->ntok, bpad = m->bpad, Q = c.ncontext, L = c.layers, b = c.adapter_dim, nq = 1 + Q;
    const int rows = B * ntok;
    const Rows R{t->row_off[ch], D, M, bpad};
    const float *zero = m->zero_bias;
    float *grads = ch == 0 ? t->grads : t->grads1;
    bf16_t *dMb = R.d(t->dMb), *tD = R.d(t->tD), *tD2 = R.d(t->tD2), *tB = R.b(t->tB), *tM = R.m(t->tM), *tQKV = R.d3(t->tQKV);
    const int64_t ralloc = t->region_rows[ch];
    // gradient of the residual stream: fp32 + its bf16 copy.  Full token rows, or -- inside the pruned last layer -- the compact
    // (CLS, concept tokens) rows of every image
    float *dHfull = R.d(t->dH), *dHc = t->dHc + (size_t)ch * t->hc_rows * D;
    bf16_t *dHbfull = R.d(t->dHb), *dHcb = t->dHcb + (size_t)ch * t->hc_rows * D;
    const bool prune = t->prune_last;
    float *dH = prune ? dHc : dHfull;
    bf16_t *dHb = prune ? dHcb : dHbfull;
    int cur = prune ? B * nq : rows;
    // the loss reads only hash_features = H[:, -Q:, :] (models/arch/coop.py:484-486): dH is zero elsewhere
    if (int e = ch_scatter_concept_rows(dhf_all + (size_t)img0 * Q * D, B, prune ? nq : ntok, Q, D, dH, dHb, s)) return e;

    // gradient of the block output arrives in dH / dHb; leaves d(branch input) = dH + adapter path in dMb (bf16 only)
    auto adapter_bwd = [&](int l, int a) -> int {
        const AdWork &aw = t->ad[l * 2 + a];
        const Saved &v = t->sv[l];
        float *pbase = t->params + (int64_t)(l * 2 + a) * t->ad_numel;
        const AdPtr ap = ad_ptrs(pbase, c);
        const bf16_t *in = R.d(a == 0 ? v.A : v.A2), *P = R.b(a == 0 ? v.P1 : v.P2), *G = R.b(a == 0 ? v.G1 : v.G2);
        const float *stin = R.st(a == 0 ? v.stA : v.stA2);
        const int64_t dalloc = dHb == dHcb ? t->hc_rows : ralloc;
        // up projection: weight-gradient products (unscaled) and dgrad
        const bool bg = t->batched_grads;
        const int ai = l * 2 + a;
        int nchunk[4] = {0, 0, 0, 0};   // G, cu, T, cd
        if (int e = ch_wgrad_tn(dHb, D, G, bpad, cur, dalloc, D, bpad, t->G[ch], t->ws_wgrad[ch], s, bg ? &nchunk[0] : nullptr)) return e;
        // of the fp32 gradient: a bias gradient is a sum over rows that largely cancels, the bf16 copy costs 1e-1 relative there
        if (int e = ch_colsum(dH, 1, D, cur, D, t->cu[ch], t->ws_colsum[ch], s, bg ? &nchunk[1] : nullptr)) return e;
        GemmCall g{bpad, D, dHb, aw.up_wT, zero, EPI_BIAS_DACT_GELU};   // dpre = s (dH W_up) o gelu'(pre), in the epilogue
        g.out = tB; g.ldo = bpad; g.aux = P; g.scale = ap.scale;
        if (int e = gemm(t, ch, cur, g, s, dalloc)) return e;
        // down projection + adapter LayerNorm: the input-gradient GEMM and the LayerNorm backward first -- the latter emits the
        // normalised adapter input x_hat (bf16) on the way, which the weight-gradient product then consumes
        g = GemmCall{D, bpad, tB, aw.down_wgT, zero, EPI_BIAS};
        g.out = tD; g.ldo = D;
        if (int e = gemm(t, ch, cur, g, s)) return e;
        if (int e = ch_ln_bwd(tD, in, stin, cur, D, 1e-5f, dH, nullptr, dMb, s, tD2)) return e;
        if (int e = ch_wgrad_tn(tB, bpad, tD2, D, cur, ralloc, bpad, D, t->T[ch], bg ? t->ws_wgradT[ch] : t->ws_wgrad[ch], s, bg ? &nchunk[2] : nullptr))
            return e;
        if (int e = ch_colsum(tB, 0, bpad, cur, bpad, t->cd[ch], bg ? t->ws_colsumD[ch] : t->ws_colsum[ch], s, bg ? &nchunk[3] : nullptr)) return e;
        if (bg) {
            const ChReduceJob jobs[4] = {{t->ws_wgrad[ch], t->Gs[ch] + (size_t)ai * D * bpad, nchunk[0], D * bpad / 4},
                                         {t->ws_colsum[ch], t->cus[ch] + (size_t)ai * D, nchunk[1], D / 4},
                                         {t->ws_wgradT[ch], t->Ts[ch] + (size_t)ai * bpad * D, nchunk[2], bpad * D / 4},
                                         {t->ws_colsumD[ch], t->cds[ch] + (size_t)ai * bpad, nchunk[3], bpad / 4}};
            return ch_reduce_partials_multi(jobs, 4, s);
        }
        return ch_adapter_grads(t->G[ch], t->cu[ch], t->T[ch], t->cd[ch], pbase, D, b, bpad, grads + (int64_t)(l * 2 + a) * t->ad_numel,
                                t->ws_colsum[ch], s);
    };

    for (int l = L - 1; l >= 0; --l) {
        const Saved &v = t->sv[l];
        const LayerT &x = t->lt[l];
        // ---- x_out = x_mid + m + adapter_2(m),  m = fc2(act(fc1(LN2(x_mid))))
        if (int e = adapter_bwd(l, 1)) return e;
        GemmCall g{M, D, dMb, x.fc2_wT, zero, c.act == 0 ? EPI_BIAS_DACT_QUICK : EPI_BIAS_DACT_GELU};
        g.out = tM; g.ldo = M; g.aux = R.m(v.F1pre);
        if (int e = gemm(t, ch, cur, g, s)) return e;
        g = GemmCall{D, M, tM, x.fc1_wgT, zero, EPI_BIAS};
        g.out = tD; g.ldo = D;
        if (int e = gemm(t, ch, cur, g, s)) return e;
        if (int e = ch_ln_bwd(tD, R.d(v.Xn2), R.st(v.st2), cur, D, c.ln_eps, dH, dH, dHb, s)) return e;
        // ---- x_mid = x_in + a + adapter_1(a),  a = out_proj(attention(qkv(LN1(x_in))))
        if (int e = adapter_bwd(l, 0)) return e;
        g = GemmCall{D, D, dMb, x.out_wT, zero, EPI_BIAS};
        g.out = tD; g.ldo = D;
        if (int e = gemm(t, ch, cur, g, s)) return e;
        const bf16_t *dctx = tD;
        if (cur != rows) {
            // leaving the pruned part of the last layer: the compact gradients go back to their token rows (zeros elsewhere) --
            // attention's keys / values cover every token, so from here on all rows carry gradient
            if (int e = ch_expand_head_rows(tD, 0, B, ntok, Q, D, tD2, s)) return e;
            if (int e = ch_expand_head_rows(dH, 1, B, ntok, Q, D, dHfull, s)) return e;
            dctx = tD2;
            dH = dHfull;
            dHb = dHbfull;
            cur = rows;
        }
        const float *dpext = !dcattn_all ? nullptr
                             : t->attn_all_layers ? dcattn_all + ((size_t)l * t->B + img0) * c.heads * Q * (ntok - Q - 1)
                             : l == L - 1 ? dcattn_all + (size_t)img0 * c.heads * Q * (ntok - Q - 1) : nullptr;
        if (int e = ch_attention_bwd(R.d3(v.QKV), dctx, B, ntok, c.heads, tQKV, s, dpext, Q)) return e;
        g = GemmCall{D, 3 * D, tQKV, x.qkv_wgT, zero, EPI_BIAS};
        g.out = tD; g.ldo = D;
        if (int e = gemm(t, ch, cur, g, s)) return e;
        if (int e = ch_ln_bwd(tD, R.d(v.Xn1), R.st(v.st1), cur, D, c.ln_eps, dH, dH, dHb, s)) return e;
    }
    if (t->batched_grads)   // every adapter's reduced products are in their slots: one launch pair assembles all the gradients
        if (int e = ch_adapter_grads(t->Gs[ch], t->cus[ch], t->Ts[ch], t->cds[ch], t->params, D, b, bpad, grads, t->ws_ag[ch], s, 2 * L, t->ad_numel))
            return e;
    // ---- concept tokens: rows ntok-Q.. of every image are pre_layrnorm(ctx[q]) (models/arch/coop.py:470-472)
    return ch_concept_rows_sum(dH, B, ntok, Q, D, t->dctx_sum[ch], s);
}

}  // namespace

extern "C" int64_t ch_adapter_arena_numel(const ch_model *m) {
    if (!m || m->cfg.adapter_dim <= 0) return 0;
    return adapter_numel(m->cfg) * m->cfg.layers * 2;
}

extern "C" void ch_trainer_destroy(ch_trainer *t) {
    if (!t) return;
    for (void *p : t->allocs) (void)hipFree(p);
    if (t->aux) (void)hipStreamDestroy(t->aux);
    if (t->ev_fork) (void)hipEventDestroy(t->ev_fork);
    if (t->ev_join) (void)hipEventDestroy(t->ev_join);
    delete t;
}

extern "C" int ch_trainer_refresh(ch_trainer *t, void *stream) {
    CH_REQUIRE(t != nullptr, "null trainer");
    const ch_model_config &c = t->m->cfg;
    const AdWork &w = t->ad[0];   // slot 0 of the contiguous per-field arrays
    return ch_adapter_refresh(t->params, t->ad_numel, c.layers * 2, c.dim, c.adapter_dim, t->m->bpad, w.down_wf, w.fold_c, w.fold_d, w.up_w,
                              w.up_wT, w.down_wgT, (hipStream_t)stream);
}

extern "C" int ch_trainer_create(ch_model *m, int32_t max_batch, float *params, float *grads, ch_trainer **out) {
    CH_REQUIRE(out != nullptr, "null out pointer");
    *out = nullptr;
    CH_REQUIRE(m != nullptr && params != nullptr && grads != nullptr, "trainer: null model / parameter arena / gradient arena");
    const ch_model_config &c = m->cfg;
    CH_REQUIRE(c.adapter_dim > 0, "trainer: the model has no adapters (nothing to train in the encoder)");
    CH_REQUIRE(max_batch >= 1, "trainer: max_batch must be >= 1");
    CH_REQUIRE(m->layers[0].qkv_wf != nullptr, "trainer: the model was built without the LayerNorm-folded weights");
    ch_trainer *t = new ch_trainer();
    t->m = m;
    t->max_batch = max_batch;
    t->params = params;
    t->grads = grads;
    t->ad_numel = adapter_numel(c);
    // options of the model the trainer is created on (ch_model_set_option "train_chains" / "train_chain_min_rows" / "train_prune_last")
    t->nchains = std::max(1, std::min(m->train_chains, TR_CHAINS));
    t->chain_min_rows = std::max<int64_t>(0, m->train_chain_min_rows);
    t->prune_last = m->train_prune_last;
    t->batched_grads = m->train_batched_grads;
    const int D = c.dim, L = c.layers, M = c.ffn, bpad = m->bpad, Q = c.ncontext;
    // region 0 holds a whole batch (one chain) or the first half (two chains); region 1 the second half
    const int half = (max_batch + 1) / 2;
    t->region_rows[0] = round_up64((int64_t)max_batch * m->ntok, 256) + 256;
    t->region_prows[0] = round_up64((int64_t)max_batch * m->np, 256) + 256;
    t->region_rows[1] = t->nchains > 1 ? round_up64((int64_t)half * m->ntok, 256) + 256 : 0;
    t->region_prows[1] = t->nchains > 1 ? round_up64((int64_t)half * m->np, 256) + 256 : 0;
    t->row_off[1] = t->region_rows[0];
    t->prow_off[1] = t->region_prows[0];
    const int64_t rows = t->region_rows[0] + t->region_rows[1], prows = t->region_prows[0] + t->region_prows[1];
    bool ok = true;
    if (t->nchains > 1) {
        ok = hipStreamCreateWithFlags(&t->aux, hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&t->ev_fork, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&t->ev_join, hipEventDisableTiming) == hipSuccess;
        if (!ok) ch_set_error("trainer: cannot create the auxiliary stream / events");
    }
    auto bf = [&](int64_t cols) { return (bf16_t *)talloc(t, sizeof(bf16_t) * rows * cols, ok); };
    auto st = [&]() { return (float *)talloc(t, sizeof(float) * rows * (D / 64) * 2, ok); };
    t->ad.resize(L * 2);
    t->lt.resize(L);
    t->sv.resize(L);
    {   // one contiguous array per field, one slot per adapter (ch_adapter_refresh fills all slots in one launch per field);
        // zero-initialised: the padding rows / columns (bottleneck b -> bpad) are never written again
        const size_t nad = (size_t)L * 2, mat = (size_t)bpad * D;
        bf16_t *dwf = (bf16_t *)talloc(t, sizeof(bf16_t) * nad * mat, ok), *uw = (bf16_t *)talloc(t, sizeof(bf16_t) * nad * mat, ok),
               *uwT = (bf16_t *)talloc(t, sizeof(bf16_t) * nad * mat, ok), *dwT = (bf16_t *)talloc(t, sizeof(bf16_t) * nad * mat, ok);
        float *fc = (float *)talloc(t, sizeof(float) * nad * bpad, ok), *fd = (float *)talloc(t, sizeof(float) * nad * bpad, ok);
        for (size_t i = 0; i < nad && ok; ++i) {
            AdWork &w = t->ad[i];
            w.down_wf = dwf + i * mat; w.up_w = uw + i * mat; w.up_wT = uwT + i * mat; w.down_wgT = dwT + i * mat;
            w.fold_c = fc + i * bpad; w.fold_d = fd + i * bpad;
        }
    }
    hipStream_t s = nullptr;
    for (int l = 0; l < L && ok; ++l) {
        const LayerW &w = m->layers[l];
        LayerT &x = t->lt[l];
        x.qkv_wgT = (bf16_t *)talloc(t, sizeof(bf16_t) * (size_t)D * 3 * D, ok);
        x.out_wT = (bf16_t *)talloc(t, sizeof(bf16_t) * (size_t)D * D, ok);
        x.fc1_wgT = (bf16_t *)talloc(t, sizeof(bf16_t) * (size_t)D * M, ok);
        x.fc2_wT = (bf16_t *)talloc(t, sizeof(bf16_t) * (size_t)M * D, ok);
        if (!ok) break;
        // the folded weights already carry gamma: W' = bf16(W * gamma) [N, D] -> (W')^T [D, N]
        int e = ch_transpose_bf16(w.qkv_wf, 3 * D, D, D, x.qkv_wgT, 3 * D, s);
        e |= ch_transpose_bf16(w.out_w, D, D, D, x.out_wT, D, s);
        e |= ch_transpose_bf16(w.fc1_wf, M, D, D, x.fc1_wgT, M, s);
        e |= ch_transpose_bf16(w.fc2_w, D, M, M, x.fc2_wT, D, s);
        if (e) ok = false;
        Saved &v = t->sv[l];
        v.Xn1 = bf(D); v.QKV = bf(3 * D); v.AO = bf(D); v.A = bf(D); v.P1 = bf(bpad); v.G1 = bf(bpad); v.Xn2 = bf(D);
        v.F1pre = bf(M); v.A2 = bf(D); v.P2 = bf(bpad); v.G2 = bf(bpad);
        v.st1 = st(); v.stA = st(); v.st2 = st(); v.stA2 = st();
    }
    t->H = (float *)talloc(t, sizeof(float) * rows * D, ok);
    t->dH = (float *)talloc(t, sizeof(float) * rows * D, ok);
    t->dHb = bf(D); t->dMb = bf(D); t->tD = bf(D); t->tD2 = bf(D); t->tB = bf(bpad); t->tM = bf(M); t->tQKV = bf(3 * D); t->F1act = bf(M);
    t->XnDummy = bf(D);
    t->stDummy = st();
    t->PATCH = (bf16_t *)talloc(t, sizeof(bf16_t) * prows * m->Kp, ok);
    t->ctx = (float *)talloc(t, sizeof(float) * Q * D, ok);
    t->hc_rows = round_up64((int64_t)max_batch * (1 + Q), 256) + 256;
    t->Hc = (float *)talloc(t, sizeof(float) * t->hc_rows * TR_CHAINS * D, ok);
    t->dHc = (float *)talloc(t, sizeof(float) * t->hc_rows * TR_CHAINS * D, ok);
    t->dHcb = (bf16_t *)talloc(t, sizeof(bf16_t) * t->hc_rows * TR_CHAINS * D, ok);
    const int64_t max_rows = (int64_t)max_batch * m->ntok;
    for (int ch = 0; ch < t->nchains; ++ch) {
        t->ws_wgrad[ch] = (float *)talloc(t, sizeof(float) * std::max(ch_wgrad_ws_floats(max_rows, D, bpad), ch_wgrad_ws_floats(max_rows, bpad, D)), ok);
        t->ws_colsum[ch] = (float *)talloc(t, sizeof(float) * ch_colsum_ws_floats(std::max(D, bpad)), ok);
        if (t->batched_grads) {
            const size_t nad = (size_t)L * 2;
            t->ws_wgradT[ch] = (float *)talloc(t, sizeof(float) * std::max(ch_wgrad_ws_floats(max_rows, D, bpad), ch_wgrad_ws_floats(max_rows, bpad, D)), ok);
            t->ws_colsumD[ch] = (float *)talloc(t, sizeof(float) * ch_colsum_ws_floats(std::max(D, bpad)), ok);
            t->Gs[ch] = (float *)talloc(t, sizeof(float) * nad * D * bpad, ok);
            t->Ts[ch] = (float *)talloc(t, sizeof(float) * nad * bpad * D, ok);
            t->cus[ch] = (float *)talloc(t, sizeof(float) * nad * D, ok);
            t->cds[ch] = (float *)talloc(t, sizeof(float) * nad * bpad, ok);
            t->ws_ag[ch] = (float *)talloc(t, sizeof(float) * nad * 256, ok);
        }
        t->G[ch] = (float *)talloc(t, sizeof(float) * (size_t)D * bpad, ok);
        t->T[ch] = (float *)talloc(t, sizeof(float) * (size_t)bpad * D, ok);
        t->cu[ch] = (float *)talloc(t, sizeof(float) * D, ok);
        t->cd[ch] = (float *)talloc(t, sizeof(float) * bpad, ok);
        t->dctx_sum[ch] = (float *)talloc(t, sizeof(float) * Q * D, ok);
    }
    if (t->nchains > 1) t->grads1 = (float *)talloc(t, sizeof(float) * t->ad_numel * L * 2, ok);
    if (ok && ch_trainer_refresh(t, nullptr) != 0) ok = false;
    if (ok && hipDeviceSynchronize() != hipSuccess) {
        ch_set_error("trainer: device error while preparing the working copies");
        ok = false;
    }
    if (!ok) {
        ch_trainer_destroy(t);
        return 4;
    }
    *out = t;
    return 0;
}

extern "C" int64_t ch_trainer_bytes(const ch_trainer *t) { return t ? (int64_t)t->bytes : 0; }

extern "C" int ch_train_forward(ch_trainer *t, const void *images, int32_t image_dtype, int32_t B, const float *concept_tokens,
                                float *out_hash_features, float *out_cls, float *out_concept_attn, int32_t concept_attn_all_layers,
                                void *stream) {
    CH_REQUIRE(t != nullptr && images != nullptr && concept_tokens != nullptr && out_hash_features != nullptr, "train_forward: null argument");
    CH_REQUIRE(B >= 1 && B <= t->max_batch, "train_forward: batch outside [1, max_batch]");
    CH_REQUIRE(image_dtype == 0 || image_dtype == 1, "train_forward: image_dtype must be 0 (fp32) or 1 (bf16)");
    hipStream_t s = (hipStream_t)stream;
    const ch_model_config &c = t->m->cfg;
    t->B = B;
    t->forward_done = false;
    t->attn_all_layers = out_concept_attn != nullptr && concept_attn_all_layers != 0;
    CH_CHECK_HIP(hipMemcpyAsync(t->ctx, concept_tokens, sizeof(float) * c.ncontext * c.dim, hipMemcpyDeviceToDevice, s));
    {
        // Two micro-batch chains or one?  Option "train_chain_min_rows" > 0: two when each chain has at least that many token rows.  0 (default):
        // two when ONE chain's narrowest 256x256-tile GEMM (N = D: out_proj, fc2 and the input-gradient products) would need more than one
        // round of the chip's 256 CUs -- a second round that is nearly empty is what a single chain pays just above the boundary (batch 112
        // of ViT-B/16: 264 tiles, 18.4 ms against 16.5 with two chains; batch 96: 228 tiles, 14.3 against 15.5; profiles/r04_train_ab_two_chains.txt)
        const int64_t rows = (int64_t)B * t->m->ntok;
        const bool two = t->chain_min_rows > 0 ? (int64_t)(B / 2) * t->m->ntok >= t->chain_min_rows
                                               : ((rows + 255) / 256) * ((c.dim + 255) / 256) > 256;
        t->nc = (t->nchains > 1 && B >= 2 && two) ? 2 : 1;
    }
    if (t->nc == 1) {
        t->Bc[0] = B; t->Bc[1] = 0;
        if (int e = forward_chain(t, 0, images, image_dtype, 0, B, out_hash_features, out_cls, out_concept_attn, s)) return e;
    } else {
        t->Bc[0] = B - B / 2; t->Bc[1] = B / 2;      // chain 1 <= ceil(max_batch / 2) images: fits its region
        CH_CHECK_HIP(hipEventRecord(t->ev_fork, s));
        CH_CHECK_HIP(hipStreamWaitEvent(t->aux, t->ev_fork, 0));
        if (int e = forward_chain(t, 0, images, image_dtype, 0, t->Bc[0], out_hash_features, out_cls, out_concept_attn, s)) return e;
        if (int e = forward_chain(t, 1, images, image_dtype, t->Bc[0], t->Bc[1], out_hash_features, out_cls, out_concept_attn, t->aux)) return e;
        CH_CHECK_HIP(hipEventRecord(t->ev_join, t->aux));
        CH_CHECK_HIP(hipStreamWaitEvent(s, t->ev_join, 0));
    }
    t->forward_done = true;
    return 0;
}

extern "C" int ch_train_backward(ch_trainer *t, const float *d_hash_features, const float *d_concept_attn, float *d_concept_tokens,
                                 void *stream) {
    CH_REQUIRE(t != nullptr && d_hash_features != nullptr && d_concept_tokens != nullptr, "train_backward: null argument");
    CH_REQUIRE(t->forward_done, "train_backward: call ch_train_forward first (its saved activations are what backward reads)");
    hipStream_t s = (hipStream_t)stream;
    const ch_model_config &c = t->m->cfg;
    const int Q = c.ncontext, D = c.dim;
    if (t->nc == 1) {
        if (int e = backward_chain(t, 0, d_hash_features, d_concept_attn, 0, t->Bc[0], s)) return e;
    } else {
        CH_CHECK_HIP(hipEventRecord(t->ev_fork, s));
        CH_CHECK_HIP(hipStreamWaitEvent(t->aux, t->ev_fork, 0));
        if (int e = backward_chain(t, 0, d_hash_features, d_concept_attn, 0, t->Bc[0], s)) return e;
        if (int e = backward_chain(t, 1, d_hash_features, d_concept_attn, t->Bc[0], t->Bc[1], t->aux)) return e;
        CH_CHECK_HIP(hipEventRecord(t->ev_join, t->aux));
        CH_CHECK_HIP(hipStreamWaitEvent(s, t->ev_join, 0));
        // the two chains' contributions: parameter gradients and concept-token rows (both linear in the per-row products)
        if (int e = ch_small_add(t->grads, t->grads1, t->ad_numel * c.layers * 2, t->grads, s)) return e;
        if (int e = ch_small_add(t->dctx_sum[0], t->dctx_sum[1], (int64_t)Q * D, t->dctx_sum[0], s)) return e;
    }
    return ch_small_ln_bwd(t->dctx_sum[0], t->ctx, t->m->pre_w, Q, D, c.ln_eps, d_concept_tokens, s);
}

// torch.optim.SGD semantics (maximize False) over a flat fp32 array, e.g. the adapter arena: see include/concepthash_hip.h
extern "C" int ch_sgd_step(float *params, const float *grads, float *momentum_buf, int64_t n, float lr, float momentum, float weight_decay,
                           float dampening, int32_t nesterov, int32_t first_step, void *stream) {
    CH_REQUIRE(params && grads && (momentum == 0.f || momentum_buf), "sgd_step: null argument");
    CH_REQUIRE(n > 0, "sgd_step: empty array");
    return ch_sgd_step_launch(params, grads, momentum_buf, n, lr, momentum, weight_decay, dampening, nesterov, first_step, (hipStream_t)stream);
}

// ---- kernel taps for the tests -------------------------------------------------------------------------------------------------
extern "C" int ch_debug_attention_bwd(const void *qkv, const void *dO, int32_t B, int32_t ntok, int32_t heads, void *dqkv, const float *dpext,
                                      int32_t ncon, void *stream) {
    CH_REQUIRE(qkv && dO && dqkv, "debug_attention_bwd: null argument");
    return ch_attention_bwd((const bf16_t *)qkv, (const bf16_t *)dO, B, ntok, heads, (bf16_t *)dqkv, (hipStream_t)stream, dpext, ncon);
}
extern "C" int ch_debug_wgrad(const void *A, int32_t lda, const void *Bm, int32_t ldb, int64_t rows, int64_t rows_alloc, int32_t N,
                              int32_t K, float *out, void *stream) {
    CH_REQUIRE(A && Bm && out, "debug_wgrad: null argument");
    float *ws = nullptr;
    CH_CHECK_HIP(hipMalloc((void **)&ws, sizeof(float) * ch_wgrad_ws_floats(rows, N, K)));
    const int e = ch_wgrad_tn((bf16_t *)A, lda, (const bf16_t *)Bm, ldb, rows, rows_alloc, N, K, out, ws, (hipStream_t)stream);
    (void)hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(ws);
    return e;
}
// row statistics of x are computed here (hb_stats on an fp32 copy is what the chain does; the tap takes bf16 x and derives the
// partials from it through an fp32 round trip), then ln_bwd; xhat_out (optional) receives normalize(x)
extern "C" int ch_debug_ln_bwd(const void *dyg, const void *x, int64_t rows, int32_t D, float eps, const float *dres_in, float *dres_out,
                               void *out_b, void *xhat_out, void *stream) {
    CH_REQUIRE(dyg && x && dres_in, "debug_ln_bwd: null argument");
    hipStream_t s = (hipStream_t)stream;
    float *st = nullptr, *xf = nullptr;
    bf16_t *hb = nullptr;
    CH_CHECK_HIP(hipMalloc((void **)&st, sizeof(float) * rows * (D / 64) * 2));
    CH_CHECK_HIP(hipMalloc((void **)&xf, sizeof(float) * rows * D));
    CH_CHECK_HIP(hipMalloc((void **)&hb, sizeof(bf16_t) * rows * D));
    int e = 0;
    {   // bf16 -> fp32 (exact) by a strided 2-byte copy into the high halves
        CH_CHECK_HIP(hipMemsetAsync(xf, 0, sizeof(float) * rows * D, s));
        CH_CHECK_HIP(hipMemcpy2DAsync((char *)xf + 2, 4, x, 2, 2, (size_t)rows * D, hipMemcpyDeviceToDevice, s));
    }
    e = ch_hb_stats(xf, rows, D, hb, st, s);
    if (!e) e = ch_ln_bwd((const bf16_t *)dyg, (const bf16_t *)x, st, rows, D, eps, dres_in, dres_out, (bf16_t *)out_b, s);
    if (!e && xhat_out) e = ch_normalize_bf16((const bf16_t *)x, st, rows, D, eps, (bf16_t *)xhat_out, s);
    (void)hipStreamSynchronize(s);
    (void)hipFree(st);
    (void)hipFree(xf);
    (void)hipFree(hb);
    return e;
}
extern "C" int ch_debug_act(const void *g, const void *pre, int64_t n, int32_t act, const float *scale_ptr, int32_t backward, void *out,
                            void *stream) {
    CH_REQUIRE(pre && out, "debug_act: null argument");
    if (backward) return ch_act_bwd((const bf16_t *)g, (const bf16_t *)pre, n, act, scale_ptr, (bf16_t *)out, (hipStream_t)stream);
    return ch_act_fwd((const bf16_t *)pre, n, act, (bf16_t *)out, (hipStream_t)stream);
}
