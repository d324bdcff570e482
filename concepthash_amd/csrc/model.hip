// ch_model: weights + workspace + the per-batch launch sequence of the ConceptHash encoder (C-ABI in
// include/concepthash_hip.h).  Restates LGHWithFixedPrompt.forward (models/arch/coop.py:524-598) as a fixed chain of
// HIP launches on one stream; see DESIGN.md for the kernel list and data layout.
#include <algorithm>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "../../include/concepthash_hip.h"
#include "../../include/concepthash_hip_debug.h"
#include "ch_common.h"
#include "kernels.h"

thread_local ch_prof_pair g_ch_prof_pair;   // (the last-error string lives in errors.cpp)
extern "C" int ch_abi_version(void) { return CH_ABI_VERSION; }

#include "model_internal.h"

namespace {

struct Builder {
    ch_model *m;
    std::map<std::string, const ch_tensor *> tab;
    hipStream_t s = nullptr;
    bool ok = true;

    void *alloc(size_t bytes) {
        void *p = nullptr;
        if (bytes == 0) bytes = 16;
        if (hipMalloc(&p, bytes) != hipSuccess) {
            ch_set_error("hipMalloc failed for " + std::to_string(bytes) + " bytes");
            ok = false;
            return nullptr;
        }
        m->allocs.push_back(p);
        m->bytes += bytes;
        return p;
    }
    const ch_tensor *find(const std::string &name, int64_t numel) {
        auto it = tab.find(name);
        if (it == tab.end()) {
            ch_set_error("missing tensor '" + name + "'");
            ok = false;
            return nullptr;
        }
        if (it->second->numel != numel) {
            ch_set_error("tensor '" + name + "' has " + std::to_string(it->second->numel) + " elements, expected " +
                         std::to_string(numel));
            ok = false;
            return nullptr;
        }
        return it->second;
    }
    bool has(const std::string &name) { return tab.count(name) != 0; }
    // fp32 host -> fp32 device
    float *f32(const std::string &name, int64_t numel) {
        const ch_tensor *t = find(name, numel);
        if (!t) return nullptr;
        float *d = (float *)alloc(sizeof(float) * numel);
        if (!d) return nullptr;
        if (hipMemcpy(d, t->data, sizeof(float) * numel, hipMemcpyDefault) != hipSuccess) {
            ch_set_error("hipMemcpy failed for '" + name + "'");
            ok = false;
        }
        return d;
    }
    float *f32_zeros(int64_t numel) {
        float *d = (float *)alloc(sizeof(float) * numel);
        if (d && hipMemset(d, 0, sizeof(float) * numel) != hipSuccess) ok = false;
        return d;
    }
    // fp32 host [rows, cols] -> bf16 device [rows, cols_pad] written at row offset into dst (or fresh alloc)
    bf16_t *bf16(const std::string &name, int64_t rows, int cols, int cols_pad, bf16_t *dst = nullptr) {
        const ch_tensor *t = find(name, rows * cols);
        if (!t) return nullptr;
        float *tmp = nullptr;
        if (hipMalloc((void **)&tmp, sizeof(float) * rows * cols) != hipSuccess) {
            ch_set_error("hipMalloc failed (staging)");
            ok = false;
            return nullptr;
        }
        if (!dst) dst = (bf16_t *)alloc(sizeof(bf16_t) * rows * cols_pad);
        if (dst) {
            if (hipMemcpy(tmp, t->data, sizeof(float) * rows * cols, hipMemcpyDefault) != hipSuccess) ok = false;
            if (ch_convert_bf16(tmp, rows, cols, cols_pad, dst, s) != 0) ok = false;
            if (hipStreamSynchronize(s) != hipSuccess) ok = false;
        }
        (void)hipFree(tmp);
        return dst;
    }
    // LayerNorm fold of a Linear made of `nparts` row blocks ([rows_each, D] weights + [rows_each] biases, host fp32):
    // wf = bf16(W * gamma) [n_pad, D] (rows past the true ones zero), c[n] = sum_k wf[n][k], d[n] = bias[n] + sum_k W[n][k] beta[k]
    bool fold(const std::string *wnames, const std::string *bnames, int nparts, int rows_each, int n_pad, int D,
              const float *gamma, const float *beta, const bf16_t **wf, const float **cc, const float **dd) {
        const int n_true = nparts * rows_each;
        float *w32 = nullptr, *b32 = nullptr;
        if (hipMalloc((void **)&w32, sizeof(float) * (size_t)n_true * D) != hipSuccess ||
            hipMalloc((void **)&b32, sizeof(float) * n_true) != hipSuccess) {
            ch_set_error("hipMalloc failed (LayerNorm fold staging)");
            ok = false;
            (void)hipFree(w32);
            return false;
        }
        for (int j = 0; j < nparts && ok; ++j) {
            const ch_tensor *wt = find(wnames[j], (int64_t)rows_each * D), *bt = find(bnames[j], rows_each);
            if (!wt || !bt) break;
            if (hipMemcpy(w32 + (size_t)j * rows_each * D, wt->data, sizeof(float) * (size_t)rows_each * D,
                          hipMemcpyDefault) != hipSuccess ||
                hipMemcpy(b32 + (size_t)j * rows_each, bt->data, sizeof(float) * rows_each, hipMemcpyDefault) != hipSuccess)
                ok = false;
        }
        bf16_t *w = (bf16_t *)alloc(sizeof(bf16_t) * (size_t)n_pad * D);
        float *c = (float *)alloc(sizeof(float) * n_pad), *d = (float *)alloc(sizeof(float) * n_pad);
        if (ok && ch_fold_ln(w32, b32, gamma, beta, n_true, n_pad, D, w, c, d, s) != 0) ok = false;
        if (hipStreamSynchronize(s) != hipSuccess) ok = false;
        (void)hipFree(w32);
        (void)hipFree(b32);
        *wf = w;
        *cc = c;
        *dd = d;
        return ok;
    }
};

int build_model(ch_model *m, const ch_tensor *tensors, int ntensors) {
    const ch_model_config &c = m->cfg;
    Builder B;
    B.m = m;
    for (int i = 0; i < ntensors; ++i) {
        CH_REQUIRE(tensors[i].name && tensors[i].data, "tensor entry with null name/data");
        B.tab[tensors[i].name] = &tensors[i];
    }
    const int D = c.dim, L = c.layers, M = c.ffn, b = c.adapter_dim, Q = c.ncontext, P = c.proj_dim, C = c.nclass;
    const int np = m->np, K = 3 * c.patch * c.patch, Kp = m->Kp;
    const std::string VM = "backbone.vision_model.";
    hipStream_t s = nullptr;

    // ---- embeddings
    m->patch_w = B.bf16(VM + "embeddings.patch_embedding.weight", D, K, Kp);
    m->pos = B.f32(VM + "embeddings.position_embedding.weight", (int64_t)(np + 1) * D);
    const float *cls = B.f32(VM + "embeddings.class_embedding", D);
    m->pre_w = B.f32(VM + "pre_layrnorm.weight", D);
    m->pre_b = B.f32(VM + "pre_layrnorm.bias", D);
    m->zero_bias = B.f32_zeros(std::max(std::max(3 * D, M), 1024));
    float *cls_pos0 = (float *)B.alloc(sizeof(float) * D);
    if (!B.ok) return 4;
    if (ch_small_add(cls, m->pos, D, cls_pos0, s)) return 4;
    m->cls_pos0 = cls_pos0;

    // ---- encoder layers
    m->layers.resize(L);
    for (int i = 0; i < L && B.ok; ++i) {
        const std::string pre = VM + "encoder.layers." + std::to_string(i) + ".";
        LayerW &w = m->layers[i];
        w.ln1_w = B.f32(pre + "layer_norm1.weight", D);
        w.ln1_b = B.f32(pre + "layer_norm1.bias", D);
        w.ln2_w = B.f32(pre + "layer_norm2.weight", D);
        w.ln2_b = B.f32(pre + "layer_norm2.bias", D);
        bf16_t *qkvw = (bf16_t *)B.alloc(sizeof(bf16_t) * 3 * D * D);
        float *qkvb = (float *)B.alloc(sizeof(float) * 3 * D);
        if (!B.ok) break;
        const char *names[3] = {"q_proj", "k_proj", "v_proj"};
        for (int j = 0; j < 3 && B.ok; ++j) {
            B.bf16(pre + "self_attn." + names[j] + ".weight", D, D, D, qkvw + (size_t)j * D * D);
            const ch_tensor *bt = B.find(pre + "self_attn." + names[j] + ".bias", D);
            if (bt && hipMemcpy(qkvb + (size_t)j * D, bt->data, sizeof(float) * D, hipMemcpyDefault) != hipSuccess)
                B.ok = false;
        }
        w.qkv_w = qkvw;
        w.qkv_b = qkvb;
        w.out_w = B.bf16(pre + "self_attn.out_proj.weight", D, D, D);
        w.out_b = B.f32(pre + "self_attn.out_proj.bias", D);
        w.fc1_w = B.bf16(pre + "mlp.fc1.weight", M, D, D);
        w.fc1_b = B.f32(pre + "mlp.fc1.bias", M);
        w.fc2_w = B.bf16(pre + "mlp.fc2.weight", D, M, M);
        w.fc2_b = B.f32(pre + "mlp.fc2.bias", D);
        if (b > 0 && B.ok) {  // LN-fold chain
            const std::string qw[3] = {pre + "self_attn.q_proj.weight", pre + "self_attn.k_proj.weight", pre + "self_attn.v_proj.weight"};
            const std::string qb[3] = {pre + "self_attn.q_proj.bias", pre + "self_attn.k_proj.bias", pre + "self_attn.v_proj.bias"};
            B.fold(qw, qb, 3, D, 3 * D, D, w.ln1_w, w.ln1_b, &w.qkv_wf, &w.qkv_c, &w.qkv_d);
            const std::string fw = pre + "mlp.fc1.weight", fb = pre + "mlp.fc1.bias";
            B.fold(&fw, &fb, 1, M, M, D, w.ln2_w, w.ln2_b, &w.fc1_wf, &w.fc1_c, &w.fc1_d);
        }
        for (int a = 0; a < 2 && B.ok && b > 0; ++a) {
            const std::string ap = pre + "adapt_mlp_" + std::to_string(a + 1) + ".";
            AdapterW &aw = w.ad[a];
            aw.ln_w = B.f32(ap + "adapter_layer_norm.weight", D);
            aw.ln_b = B.f32(ap + "adapter_layer_norm.bias", D);
            aw.scale = B.f32(ap + "scale", 1);
            // bottleneck padded to the GEMM N/K granularity (128): padded rows/cols are zero, GELU(0) = 0
            bf16_t *dw = (bf16_t *)B.alloc(sizeof(bf16_t) * (size_t)m->bpad * D);
            float *db = B.f32_zeros(m->bpad);
            bf16_t *uw = (bf16_t *)B.alloc(sizeof(bf16_t) * (size_t)D * m->bpad);
            if (!B.ok) break;
            if (hipMemset(dw, 0, sizeof(bf16_t) * (size_t)m->bpad * D) != hipSuccess) B.ok = false;
            B.bf16(ap + "down_proj.weight", b, D, D, dw);
            const ch_tensor *bt = B.find(ap + "down_proj.bias", b);
            if (bt && hipMemcpy(db, bt->data, sizeof(float) * b, hipMemcpyDefault) != hipSuccess) B.ok = false;
            B.bf16(ap + "up_proj.weight", D, b, m->bpad, uw);
            aw.down_w = dw;
            aw.down_b = db;
            aw.up_w = uw;
            aw.up_b = B.f32(ap + "up_proj.bias", D);
            if (B.ok) {
                const std::string dwn = ap + "down_proj.weight", dbn = ap + "down_proj.bias";
                B.fold(&dwn, &dbn, 1, b, m->bpad, D, aw.ln_w, aw.ln_b, &aw.down_wf, &aw.fold_c, &aw.fold_d);
            }
        }
    }
    if (!B.ok) return 4;

    // ---- concept tokens: forward_hash_query (coop.py:413-427), folded here once
    {
        const float *hq = B.f32("hash_queries", (int64_t)Q * P);
        const float *inw = B.f32("hash_attention.sa.in_proj_weight", (int64_t)3 * P * P);
        const float *inb = B.f32("hash_attention.sa.in_proj_bias", 3 * P);
        const float *ow = B.f32("hash_attention.sa.out_proj.weight", (int64_t)P * P);
        const float *ob = B.f32("hash_attention.sa.out_proj.bias", P);
        const float *f0w = B.f32("hash_attention.ffn.0.weight", (int64_t)P * P);
        const float *f0b = B.f32("hash_attention.ffn.0.bias", P);
        const float *f3w = B.f32("hash_attention.ffn.3.weight", (int64_t)P * P);
        const float *f3b = B.f32("hash_attention.ffn.3.bias", P);
        const float *n1w = B.f32("hash_attention.norm1.weight", P);
        const float *n1b = B.f32("hash_attention.norm1.bias", P);
        const float *n2w = B.f32("hash_attention.norm2.weight", P);
        const float *n2b = B.f32("hash_attention.norm2.bias", P);
        const float *f2w = B.f32("hash_attention.ffn2.weight", (int64_t)D * P);
        const float *f2b = B.f32("hash_attention.ffn2.bias", D);
        float *qkv = (float *)B.alloc(sizeof(float) * Q * 3 * P);
        float *t0 = (float *)B.alloc(sizeof(float) * Q * P);
        float *t1 = (float *)B.alloc(sizeof(float) * Q * P);
        float *t2 = (float *)B.alloc(sizeof(float) * Q * P);
        float *ctx = (float *)B.alloc(sizeof(float) * Q * D);
        if (!B.ok) return 4;
        int e = 0;
        e |= ch_small_linear(hq, Q, P, inw, inb, 3 * P, 0, qkv, s);
        e |= ch_small_mha(qkv, Q, P, c.upt_heads, t0, s);
        e |= ch_small_linear(t0, Q, P, ow, ob, P, 0, t1, s);         // sa(x)
        e |= ch_small_layernorm(hq, Q, P, n1w, n1b, 1e-5f, t0, s);   // norm1(x)
        e |= ch_small_add(t0, t1, (int64_t)Q * P, t2, s);            // x1 = norm1(x) + sa(x)
        e |= ch_small_linear(t2, Q, P, f0w, f0b, P, 1, t0, s);       // relu(ffn.0(x1))
        e |= ch_small_linear(t0, Q, P, f3w, f3b, P, 0, t1, s);       // ffn(x1)
        e |= ch_small_layernorm(t2, Q, P, n2w, n2b, 1e-5f, t0, s);   // norm2(x1)
        e |= ch_small_add(t0, t1, (int64_t)Q * P, t2, s);            // x2
        e |= ch_small_linear(t2, Q, P, f2w, f2b, D, 0, ctx, s);
        if (e) return 4;
        m->ctx = ctx;
    }

    // ---- hash head
    m->hash_pe = B.has("hash_pe") ? B.f32("hash_pe", (int64_t)Q * D) : B.f32_zeros((int64_t)Q * D);
    m->hash_fc = B.f32("hash_fc.weight", (int64_t)(c.nbit / Q) * D);
    {
        float *sc = (float *)B.alloc(sizeof(float) * c.nbit), *sh = (float *)B.alloc(sizeof(float) * c.nbit);
        if (!B.ok) return 4;
        if (B.has("hash_bn.running_mean")) {
            const float *w = B.f32("hash_bn.weight", c.nbit), *bb = B.f32("hash_bn.bias", c.nbit);
            const float *mu = B.f32("hash_bn.running_mean", c.nbit), *var = B.f32("hash_bn.running_var", c.nbit);
            if (!B.ok) return 4;
            if (ch_bn_fold(w, bb, mu, var, c.nbit, c.bn_eps, sc, sh, s)) return 4;
        } else {
            std::vector<float> ones(c.nbit, 1.0f);
            if (hipMemcpy(sc, ones.data(), sizeof(float) * c.nbit, hipMemcpyHostToDevice) != hipSuccess) return 4;
            if (hipMemset(sh, 0, sizeof(float) * c.nbit) != hipSuccess) return 4;
        }
        m->bn_scale = sc;
        m->bn_shift = sh;
    }
    // ---- centres: get_center (coop.py:624-625) + l2 / sign variants (coop.py:573-580)
    {
        const int cd = c.center_dim;
        const float *cen = B.f32("center", (int64_t)C * cd);
        float *proj = (float *)B.alloc(sizeof(float) * C * c.nbit);
        if (B.has("text_projection.0.weight")) {
            const float *w0 = B.f32("text_projection.0.weight", (int64_t)cd * cd), *b0 = B.f32("text_projection.0.bias", cd);
            const float *w2 = B.f32("text_projection.2.weight", (int64_t)c.nbit * cd),
                        *b2 = B.f32("text_projection.2.bias", c.nbit);
            float *t = (float *)B.alloc(sizeof(float) * C * cd);
            if (!B.ok) return 4;
            if (ch_small_linear(cen, C, cd, w0, b0, cd, 1, t, s)) return 4;
            if (ch_small_linear(t, C, cd, w2, b2, c.nbit, 0, proj, s)) return 4;
        } else {
            const float *w = B.f32("text_projection.weight", (int64_t)c.nbit * cd), *bb = B.f32("text_projection.bias", c.nbit);
            if (!B.ok) return 4;
            if (ch_small_linear(cen, C, cd, w, bb, c.nbit, 0, proj, s)) return 4;
        }
        float *l2 = (float *)B.alloc(sizeof(float) * C * c.nbit), *bin = (float *)B.alloc(sizeof(float) * C * c.nbit);
        if (!B.ok) return 4;
        if (ch_small_l2norm(proj, C, c.nbit, l2, bin, s)) return 4;
        m->center_l2 = l2;
        m->center_bin = bin;
    }
    if (B.has("concept_ce.centroids")) {
        m->concept_pe = B.f32("concept_pe", (int64_t)Q * D);
        const float *cc = B.f32("concept_ce.centroids", (int64_t)C * D);
        float *l2 = (float *)B.alloc(sizeof(float) * C * D);
        if (!B.ok) return 4;
        if (ch_small_l2norm(cc, C, D, l2, nullptr, s)) return 4;
        m->concept_cent_l2 = l2;
    }
    if (B.has(VM + "post_layernorm.weight") && B.has("backbone.visual_projection.weight")) {
        m->post_w = B.f32(VM + "post_layernorm.weight", D);
        m->post_b = B.f32(VM + "post_layernorm.bias", D);
        m->vis_proj = B.f32("backbone.visual_projection.weight", (int64_t)P * D);
    }
    if (!B.ok) return 4;

    // ---- workspace (rows padded to the GEMM block tile; padding rows are zero and never read back)
    const int64_t rows = round_up64((int64_t)c.max_batch * m->ntok, 256) + 256;   // +256: a micro-batch's last tile over-reads
    const int64_t prows = round_up64((int64_t)c.max_batch * np, 256) + 256;
    m->rows_alloc = rows;
    m->prow_alloc = prows;
    m->H = (float *)B.alloc(sizeof(float) * rows * D);
    m->Xn = (bf16_t *)B.alloc(sizeof(bf16_t) * rows * D);
    // (the split-K tail's slabs are allocated when ch_model_set_option switches "splitk" on)
    m->Hc = (float *)B.alloc(sizeof(float) * ((size_t)c.max_batch * (1 + c.ncontext) + 512) * D);
    m->head_xn = (float *)B.alloc(sizeof(float) * ((size_t)c.max_batch * c.ncontext + 16) * D);
    m->head_cls = (float *)B.alloc(sizeof(float) * ((size_t)c.max_batch + 16) * D);
    m->statsA = (float *)B.alloc(sizeof(float) * rows * (D / 64) * 2);
    m->statsH = (float *)B.alloc(sizeof(float) * rows * (D / 64) * 2);
    m->QKV = (bf16_t *)B.alloc(sizeof(bf16_t) * rows * 3 * D);
    m->AO = (bf16_t *)B.alloc(sizeof(bf16_t) * rows * D);
    m->A = (bf16_t *)B.alloc(sizeof(bf16_t) * rows * D);
    m->AD = (bf16_t *)B.alloc(sizeof(bf16_t) * rows * std::max(m->bpad, 128));
    m->F1 = (bf16_t *)B.alloc(sizeof(bf16_t) * rows * M);
    m->PATCH = (bf16_t *)B.alloc(sizeof(bf16_t) * prows * Kp);
    if (!B.ok) return 4;
    CH_CHECK_HIP(hipMemset(m->H, 0, sizeof(float) * rows * D));
    CH_CHECK_HIP(hipMemset(m->Xn, 0, sizeof(bf16_t) * rows * D));
    CH_CHECK_HIP(hipMemset(m->Hc, 0, sizeof(float) * ((size_t)c.max_batch * (1 + c.ncontext) + 512) * D));
    CH_CHECK_HIP(hipMemset(m->statsA, 0, sizeof(float) * rows * (D / 64) * 2));
    CH_CHECK_HIP(hipMemset(m->statsH, 0, sizeof(float) * rows * (D / 64) * 2));
    CH_CHECK_HIP(hipMemset(m->QKV, 0, sizeof(bf16_t) * rows * 3 * D));
    CH_CHECK_HIP(hipMemset(m->AO, 0, sizeof(bf16_t) * rows * D));
    CH_CHECK_HIP(hipMemset(m->A, 0, sizeof(bf16_t) * rows * D));
    CH_CHECK_HIP(hipMemset(m->AD, 0, sizeof(bf16_t) * rows * std::max(m->bpad, 128)));
    CH_CHECK_HIP(hipMemset(m->F1, 0, sizeof(bf16_t) * rows * M));
    CH_CHECK_HIP(hipMemset(m->PATCH, 0, sizeof(bf16_t) * prows * Kp));
    CH_CHECK_HIP(hipDeviceSynchronize());
    return 0;
}

// profiler mark: called immediately before a launch of category `cat` doing `flops` algorithmic FLOPs
inline void mark(ch_model *m, int pi, int cat, double flops, hipStream_t s) {
    ch_model::Prof &P = m->prof[pi];
    g_ch_prof_pair = ch_prof_pair();
    if (!m->prof_on || P.n + 1 >= P.ev.size()) return;
    (void)hipEventRecord(P.ev[P.n], s);
    P.cat[P.n] = cat;
    P.flops[P.n] = flops;
    P.exact[P.n] = 0;
    if (cat != CH_CAT_END) {   // the launch that follows may take the pair (CH_LAUNCH): then its time is the dispatch's own
        g_ch_prof_pair.start = P.kstart[P.n];
        g_ch_prof_pair.stop = P.kstop[P.n];
        g_ch_prof_pair.used = (bool *)&P.exact[P.n];
    }
    P.n++;
}

// one launch chain: images [img0, img0 + B) through `nlayers` layers on stream s; rows of every activation buffer are
// independent, so a micro-batch simply works on its own row range of the shared workspace
// prune: (ch_encode) in the final layer only the rows the hashing head reads -- CLS and the Q concept tokens of every image --
// are carried past the attention (whose keys / values still cover all tokens); their residual lives in mm->Hc afterwards
int run_chain(ch_model *mm, int pi, const void *images_all, int image_dtype, int img0, int B, int nlayers, hipStream_t s,
              float *concept_attn, bool attn_all_layers, bool prune, int Btot) {
    const ch_model_config &c = mm->cfg;
    const int D = c.dim, M = c.ffn, ntok = mm->ntok, np = mm->np;
    const int rows = B * ntok;
    const int act_epi = c.act == 0 ? EPI_BIAS_QUICKGELU : EPI_BIAS_GELU;
    const size_t row0 = (size_t)img0 * ntok, prow0 = (size_t)img0 * np;
    // view of the workspace for this micro-batch
    struct View {
        float *H, *statsA, *statsH;
        bf16_t *Xn, *QKV, *AO, *A, *AD, *F1, *PATCH;
        int64_t rows_alloc, prow_alloc;
    } v{mm->H + row0 * D, mm->statsA + row0 * (D / 64) * 2, mm->statsH + row0 * (D / 64) * 2, mm->Xn + row0 * D, mm->QKV + row0 * 3 * D, mm->AO + row0 * D, mm->A + row0 * D,
        mm->AD + row0 * std::max(mm->bpad, 128), mm->F1 + row0 * M, mm->PATCH + prow0 * mm->Kp,
        mm->rows_alloc - (int64_t)row0, mm->prow_alloc - (int64_t)prow0};
    View *m = &v;
    const size_t img_elems = (size_t)3 * c.image_size * c.image_size;
    const void *images = (const char *)images_all + (size_t)img0 * img_elems * (image_dtype == 0 ? 4 : 2);
    if (concept_attn) concept_attn += (size_t)img0 * c.heads * c.ncontext * np;

    mark(mm, pi, CH_CAT_IM2COL, 0.0, s);
    if (int e = ch_im2col(images, image_dtype, B, c.image_size, c.patch, mm->Kp, m->PATCH, s)) return e;
    mark(mm, pi, CH_CAT_GEMM_PATCH, 2.0 * B * np * (double)D * 3.0 * c.patch * c.patch, s);
    {
        GemmParams p{};
        p.X = m->PATCH; p.W = mm->patch_w; p.M = B * np; p.N = D; p.K = mm->Kp; p.X_rows_alloc = m->prow_alloc;
        p.bias = nullptr; p.resid = m->H; p.ldr = D; p.pos = mm->pos; p.tokens_per_img = ntok; p.patches_per_img = np;
        p.pp_min_k = mm->pp_min_k; p.pp_sched = mm->pp_sched; p.small_kernel = mm->small_kernel; p.group_n_opt = mm->group_n;
        if (int e = ch_gemm_bf16(p, EPI_PATCH, s)) return e;
    }
    const LayerW &w0 = mm->layers[0];
    mark(mm, pi, CH_CAT_ROWOPS, 0.0, s);
    if (int e = ch_assemble_preln(m->H, B, ntok, np, D, mm->cls_pos0, mm->ctx, mm->pre_w, mm->pre_b, w0.ln1_w, w0.ln1_b,
                                  c.ln_eps, m->Xn, s))
        return e;

    // cat / n_true / k_true: profiler category and the un-padded (algorithmic) GEMM extents
    // LayerNorm-fold plumbing of one GEMM: statistics it consumes (fold_c != nullptr) and/or produces (stats_out != nullptr)
    struct Fold {
        const float *stats_in = nullptr, *fold_c = nullptr;
        float eps = 0.f;
        float *stats_out = nullptr;
        bf16_t *hb_out = nullptr;
    };
    const bool fold = mm->ln_fold && c.adapter_dim > 0 && !mm->use_fused_adapter;
    int dir = 0;              // serpentine: flipped before every row-streaming launch of the chain
    auto next_dir = [&]() { return mm->serpentine ? (dir ^= 1) : 0; };
    int cur_rows = rows;      // rows the GEMMs work on: all token rows, or the compact head rows in the pruned final layer
    float *cur_H = m->H;      // ... and their residual stream
    auto pruned_cat = [](int cat) {   // profiler category of the same launch on the compact head rows (final layer)
        switch (cat) {
            case CH_CAT_GEMM_OUT: return (int)CH_CAT_GEMM_OUT_PRUNED;
            case CH_CAT_GEMM_DOWN: return (int)CH_CAT_GEMM_DOWN_PRUNED;
            case CH_CAT_GEMM_UP: return (int)CH_CAT_GEMM_UP_PRUNED;
            case CH_CAT_GEMM_FC1: return (int)CH_CAT_GEMM_FC1_PRUNED;
            case CH_CAT_GEMM_FC2: return (int)CH_CAT_GEMM_FC2_PRUNED;
        }
        return cat;
    };
    auto gemm = [&](int cat, int n_true, int k_true, const bf16_t *X, const bf16_t *W, int N, int K, const float *bias,
                    int epi, bf16_t *out, int ldo, const float *scale, const bf16_t *addend = nullptr, const Fold &f = Fold()) {
        mark(mm, pi, cur_rows != rows ? pruned_cat(cat) : cat, 2.0 * cur_rows * (double)n_true * k_true, s);
        GemmParams p{};
        p.tag = cat == CH_CAT_GEMM_FC2 ? 1 : 0;   // fc2 runs out_proj's kernel instance: second symbol name for per-kernel profiles
        p.splitk_ws = mm->splitk_ws[pi]; p.splitk_cnt = mm->splitk_cnt[pi]; p.pp_min_k = mm->pp_min_k; p.pp_sched = mm->pp_sched; p.small_kernel = mm->small_kernel; p.rev = next_dir();
        p.nt_resid_opt = mm->resid_nt; p.nt_out_opt = mm->nt_out; p.group_n_opt = mm->group_n; p.splitk_opt = mm->splitk; p.rows_opt = mm->gemm_rows; p.wide_opt = mm->wide_kernel;
        p.footprint_rows = (int64_t)cur_rows * Btot / B;   // all concurrent chains of this call: what the cache-policy choice is sized on
        p.stats_in = f.stats_in; p.fold_c = f.fold_c; p.ln_eps = f.eps; p.stats_out = f.stats_out; p.hb_out = f.hb_out; p.ld_hb = D;
        p.addend = addend; p.ld_addend = D;
        p.X = X; p.W = W; p.M = cur_rows; p.N = N; p.K = K; p.X_rows_alloc = m->rows_alloc; p.bias = bias;
        p.out_bf16 = out; p.ldo = ldo; p.resid = cur_H; p.ldr = D; p.scale_ptr = scale;
        return ch_gemm_bf16(p, epi, s);
    };
    // emit_h: (LN-fold chain) the up-projection also writes bf16(H) to Xn + its row statistics for the next folded GEMM
    auto adapter = [&](const AdapterW &aw, bool emit_h) -> int {
        if (!aw.down_w) return 0;
        if (fold) {
            // adapter LayerNorm folded into the down projection, which reads the sub-block output `a` (m->A) directly
            Fold fd;
            fd.stats_in = m->statsA; fd.fold_c = aw.fold_c; fd.eps = 1e-5f;
            if (int e = gemm(CH_CAT_GEMM_DOWN, c.adapter_dim, D, m->A, aw.down_wf, mm->bpad, D, aw.fold_d, EPI_FOLD_GELU, m->AD,
                             mm->bpad, nullptr, nullptr, fd))
                return e;
            Fold fu;
            if (emit_h) { fu.stats_out = m->statsH; fu.hb_out = m->Xn; }
            return gemm(CH_CAT_GEMM_UP, D, c.adapter_dim, m->AD, aw.up_w, D, mm->bpad, aw.up_b,
                        emit_h ? EPI_SCALE_RESID_STATS : EPI_SCALE_RESID, nullptr, 0, aw.scale, m->A, fu);
        }
        if (aw.down_wf && mm->use_fused_adapter && ch_adapter_fused_supported(D, mm->bpad)) {
            // LN + down + GELU + up + residual in one launch (adapter_fused.hip); a = m->A (bf16), H updated in place
            mark(mm, pi, CH_CAT_ADAPTER, 4.0 * rows * (double)D * c.adapter_dim, s);
            AdapterParams ap{};
            ap.A = m->A; ap.H = m->H; ap.M = rows; ap.D = D; ap.bpad = mm->bpad; ap.Wd = aw.down_wf; ap.c = aw.fold_c;
            ap.d = aw.fold_d; ap.Wu = aw.up_w; ap.bu = aw.up_b; ap.scale = aw.scale; ap.eps = 1e-5f;
            return ch_adapter_fused(ap, s);
        }
        // Adapter (models/layers/adapter.py:46-60) on the bf16 copy of the sub-block output held in m->A
        mark(mm, pi, CH_CAT_ROWOPS, 0.0, s);
        if (int e = ch_layernorm_bf16(m->A, rows, D, aw.ln_w, aw.ln_b, 1e-5f, m->Xn, s)) return e;
        if (int e = gemm(CH_CAT_GEMM_DOWN, c.adapter_dim, D, m->Xn, aw.down_w, mm->bpad, D, aw.down_b, EPI_BIAS_GELU, m->AD,
                         mm->bpad, nullptr))
            return e;
        // one fp32 read-modify-write of the residual per sub-block: H += a + scale * (up(...) + bias), `a` read back as bf16
        return gemm(CH_CAT_GEMM_UP, D, c.adapter_dim, m->AD, aw.up_w, D, mm->bpad, aw.up_b, EPI_SCALE_RESID, nullptr, 0,
                    aw.scale, m->A);
    };

    Fold stats_a;  // out_proj / fc2 of the LN-fold chain: row statistics of `a` for the adapter's folded LayerNorm
    if (fold) stats_a.stats_out = m->statsA;
    for (int i = 0; i < nlayers; ++i) {
        const LayerW &w = mm->layers[i];
        if (i > 0 && fold) {
            // layer_norm1 folded: Xn holds bf16(H) and statsH its row statistics (previous layer's second up-projection)
            Fold fq;
            fq.stats_in = m->statsH; fq.fold_c = w.qkv_c; fq.eps = c.ln_eps;
            if (int e = gemm(CH_CAT_GEMM_QKV, 3 * D, D, m->Xn, w.qkv_wf, 3 * D, D, w.qkv_d, EPI_FOLD_BIAS, m->QKV, 3 * D, nullptr,
                             nullptr, fq))
                return e;
        } else {
            if (i > 0) {
                mark(mm, pi, CH_CAT_ROWOPS, 0.0, s);
                if (int e = ch_layernorm_f32(m->H, rows, D, w.ln1_w, w.ln1_b, c.ln_eps, m->Xn, s)) return e;
            }
            if (int e = gemm(CH_CAT_GEMM_QKV, 3 * D, D, m->Xn, w.qkv_w, 3 * D, D, w.qkv_b, EPI_BIAS, m->QKV, 3 * D, nullptr)) return e;
        }
        const bool pruned = prune && fold && mm->prune_last && i == nlayers - 1 && nlayers == c.layers;
        const int nq = 1 + c.ncontext;
        mark(mm, pi, pruned ? CH_CAT_ATTENTION_PRUNED : CH_CAT_ATTENTION, 4.0 * B * (double)(pruned ? nq : ntok) * ntok * D, s);
        // concept-token attention tap: the last layer's rows, or (the call's concept_attn_all_layers) every layer's, [L, Btot, heads, Q, Np]
        float *cattn = !concept_attn ? nullptr
                       : attn_all_layers ? concept_attn + (size_t)i * Btot * c.heads * c.ncontext * np
                       : i == nlayers - 1 ? concept_attn : nullptr;
        if (int e = ch_attention(m->QKV, B, ntok, c.heads, m->AO, s, cattn, c.ncontext, pruned, next_dir() != 0))
            return e;
        if (pruned) {
            // from here on every buffer holds B * (1 + Q) compact rows (image-major: CLS, then the concept tokens)
            cur_rows = B * nq;
            cur_H = mm->Hc + (size_t)img0 * nq * D;
            mark(mm, pi, CH_CAT_ROWOPS, 0.0, s);
            if (int e = ch_gather_head_rows(m->H, B, ntok, c.ncontext, D, cur_H, s)) return e;
        }
        // h = r + a  (+ adapter_1(a) below);  a kept as bf16 in m->A for the adapter branch
        // with adapters the residual add of `a` is deferred to the adapter's up-projection epilogue (see adapter())
        const int sub_epi = fold ? EPI_BIAS_STATS : (w.ad[0].down_w ? EPI_BIAS : EPI_BIAS_RESID);
        if (int e = gemm(CH_CAT_GEMM_OUT, D, D, m->AO, w.out_w, D, D, w.out_b, sub_epi, m->A, D, nullptr, nullptr, stats_a)) return e;
        if (int e = adapter(w.ad[0], true)) return e;
        if (fold) {
            Fold f1;
            f1.stats_in = m->statsH; f1.fold_c = w.fc1_c; f1.eps = c.ln_eps;
            if (int e = gemm(CH_CAT_GEMM_FC1, M, D, m->Xn, w.fc1_wf, M, D, w.fc1_d,
                             c.act == 0 ? EPI_FOLD_QUICKGELU : EPI_FOLD_GELU, m->F1, M, nullptr, nullptr, f1))
                return e;
        } else {
            mark(mm, pi, CH_CAT_ROWOPS, 0.0, s);
            if (int e = ch_layernorm_f32(m->H, rows, D, w.ln2_w, w.ln2_b, c.ln_eps, m->Xn, s)) return e;
            if (int e = gemm(CH_CAT_GEMM_FC1, M, D, m->Xn, w.fc1_w, M, D, w.fc1_b, act_epi, m->F1, M, nullptr)) return e;
        }
        if (int e = gemm(CH_CAT_GEMM_FC2, D, M, m->F1, w.fc2_w, D, M, w.fc2_b, sub_epi, m->A, D, nullptr, nullptr, stats_a)) return e;
        if (int e = adapter(w.ad[1], i + 1 < nlayers)) return e;
    }
    return 0;
}

// encoder up to `nlayers` layers; leaves the residual stream in m->H.  With two streams the batch is split into two
// micro-batches whose chains run concurrently (fork/join by events on the caller's stream).
int run_encoder(ch_model *m, const void *images, int image_dtype, int B, int nlayers, hipStream_t s,
                float *concept_attn = nullptr, bool attn_all_layers = false, bool prune = false) {
    int ns = std::min(m->nstreams, B);
    if (ns == 2 && m->chain_auto) {
        // Two chains are the default because they win from batch 32 up (+2 % there, +8 % at batch 256, +22 % at batch 112, where one
        // chain's N = D GEMMs are 264 tiles = one round and a sliver).  Below ~5,600 token rows every launch is one tile's latency whatever
        // its size, and splitting only doubles the launches: batch 4 / 8 / 12 / 16 / 24 of ViT-B/16 measure 2-5 % faster as ONE chain
        // (1.93 vs 2.03 ms at batch 8) -- profiles/r04_encode_streams_by_batch.txt.  (A second rule, one chain when its N = D GEMMs fill
        // >= 85 % of exactly one round of tiles, measured +2.6 % on one box and -1.3 % on another at batch 96: not kept.)
        const int64_t rows = (int64_t)B * m->ntok;
        if (rows < 5600) ns = 1;
    }
    m->last_chains = ns;
    if (ns < 2) return run_chain(m, 0, images, image_dtype, 0, B, nlayers, s, concept_attn, attn_all_layers, prune, B);
    // micro-batch i = images [i*B/ns, (i+1)*B/ns); chain 0 on the caller's stream, the others fork from / join it
    CH_CHECK_HIP(hipEventRecord(m->ev_fork, s));
    for (int i = 0; i < ns; ++i) {
        const int b0 = (int)((int64_t)B * i / ns), b1 = (int)((int64_t)B * (i + 1) / ns);
        hipStream_t si = i == 0 ? s : m->aux_stream[i - 1];
        if (i > 0) CH_CHECK_HIP(hipStreamWaitEvent(si, m->ev_fork, 0));
        if (int e = run_chain(m, i, images, image_dtype, b0, b1 - b0, nlayers, si, concept_attn, attn_all_layers, prune, B)) return e;
        if (i > 0) {
            CH_CHECK_HIP(hipEventRecord(m->ev_join[i - 1], si));
            CH_CHECK_HIP(hipStreamWaitEvent(s, m->ev_join[i - 1], 0));
        }
    }
    return 0;
}

void drop_graphs(ch_model *m) {
    // a replay may still be executing: destroy an executable graph only when the device has drained (opt-in path, rare call)
    if (!m->graphs.empty()) (void)hipDeviceSynchronize();
    for (auto &kv : m->graphs) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    m->graphs.clear();
}

int validate(const ch_model_config *c) {
    CH_REQUIRE(c != nullptr, "null config");
    CH_REQUIRE(c->dim > 0 && c->dim % 128 == 0 && c->dim <= 1280, "dim must be a multiple of 128 and <= 1280");
    CH_REQUIRE(c->heads > 0 && c->dim == c->heads * 64, "head_dim must be 64 (dim == heads * 64)");
    CH_REQUIRE(c->ffn > 0 && c->ffn % 128 == 0, "ffn must be a multiple of 128");
    CH_REQUIRE(c->layers >= 1, "layers must be >= 1");
    CH_REQUIRE(c->patch > 0 && c->image_size > 0 && c->image_size % c->patch == 0, "image_size must be divisible by patch");
    CH_REQUIRE(c->ncontext >= 1 && c->ncontext <= 64, "ncontext must be in [1, 64]");
    CH_REQUIRE(c->nbit > 0 && c->nbit % c->ncontext == 0, "nbit must be divisible by ncontext");
    CH_REQUIRE(c->nclass > 0 && c->proj_dim > 0 && c->center_dim > 0, "nclass / proj_dim / center_dim must be positive");
    CH_REQUIRE(c->upt_heads > 0 && c->proj_dim % c->upt_heads == 0, "proj_dim must be divisible by upt_heads");
    CH_REQUIRE(c->adapter_dim >= 0, "adapter_dim must be >= 0");
    CH_REQUIRE(c->max_batch >= 1, "max_batch must be >= 1");
    CH_REQUIRE(c->act == 0 || c->act == 1, "act must be 0 (quick_gelu) or 1 (gelu)");
    const int grid = c->image_size / c->patch;
    CH_REQUIRE(1 + grid * grid + c->ncontext <= 288, "more than 288 tokens per image is not supported");
    return 0;
}

}  // namespace

extern "C" int ch_model_create(const ch_model_config *cfg, const ch_tensor *tensors, int32_t ntensors, ch_model **out) {
    CH_REQUIRE(out != nullptr, "null out pointer");
    *out = nullptr;
    if (int e = validate(cfg)) return e;
    CH_REQUIRE(tensors != nullptr && ntensors > 0, "no tensors");
    ch_model *m = new ch_model();
    m->cfg = *cfg;
    if (m->cfg.ln_eps <= 0.f) m->cfg.ln_eps = 1e-5f;
    if (m->cfg.bn_eps <= 0.f) m->cfg.bn_eps = 1e-5f;
    const int grid = cfg->image_size / cfg->patch;
    m->np = grid * grid;
    m->ntok = 1 + m->np + cfg->ncontext;
    m->Kp = (int)round_up64(3 * cfg->patch * cfg->patch, 64);
    bool aux_ok = hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; aux_ok && i < CH_MAX_STREAMS - 1; ++i)
        aux_ok = hipStreamCreateWithFlags(&m->aux_stream[i], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&m->ev_join[i], hipEventDisableTiming) == hipSuccess;
    if (!aux_ok) {
        ch_set_error("cannot create the auxiliary streams / events");
        ch_model_destroy(m);
        return 4;
    }
    m->bpad = (int)round_up64(cfg->adapter_dim, 128);
    int e = build_model(m, tensors, ntensors);
    if (e == 0 && hipDeviceSynchronize() != hipSuccess) {
        ch_set_error("device error while folding weights");
        e = 4;
    }
    if (e) {
        ch_model_destroy(m);
        return e;
    }
    *out = m;
    return 0;
}

extern "C" void ch_model_destroy(ch_model *m) {
    if (!m) return;
    drop_graphs(m);
    if (m->cap_stream) (void)hipStreamDestroy(m->cap_stream);
    for (auto &P : m->prof) {
        for (hipEvent_t e : P.ev) (void)hipEventDestroy(e);
        for (hipEvent_t e : P.kstart) (void)hipEventDestroy(e);
        for (hipEvent_t e : P.kstop) (void)hipEventDestroy(e);
    }
    for (hipStream_t a : m->aux_stream)
        if (a) (void)hipStreamDestroy(a);
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    for (hipEvent_t e : m->ev_join)
        if (e) (void)hipEventDestroy(e);
    for (void *p : m->allocs)
        if (p) (void)hipFree(p);
    delete m;
}

extern "C" size_t ch_model_device_bytes(const ch_model *m) { return m ? m->bytes : 0; }

// ---- per-handle options (include/concepthash_hip.h: ch_model_set_option).  The library reads no environment variable: every knob of the
// launch chain is state of ONE opaque handle, set through this entry point (the Python wrapper maps its debug environment overrides onto it).
namespace {
struct OptionRef {
    const char *key;
    int kind;   // 0 int field, 1 bool field, 2 int64 field
    void *(*field)(ch_model *);
    int64_t lo, hi;
};
#define CH_OPT_FIELD(name) [](ch_model *m) -> void * { return &m->name; }
const OptionRef g_options[] = {
    {"streams", 0, CH_OPT_FIELD(nstreams), 1, CH_MAX_STREAMS},
    {"chain_auto", 1, CH_OPT_FIELD(chain_auto), 0, 1},
    {"ln_fold", 1, CH_OPT_FIELD(ln_fold), 0, 1},
    {"prune_last", 1, CH_OPT_FIELD(prune_last), 0, 1},
    {"pp_min_k", 0, CH_OPT_FIELD(pp_min_k), 0, 1 << 20},
    {"small_kernel", 0, CH_OPT_FIELD(small_kernel), 0, 2},
    {"serpentine", 1, CH_OPT_FIELD(serpentine), 0, 1},
    {"pp_sched", 0, CH_OPT_FIELD(pp_sched), 0, 2},
    {"fused_adapter", 1, CH_OPT_FIELD(use_fused_adapter), 0, 1},
    {"resid_nt", 0, CH_OPT_FIELD(resid_nt), -1, 1},
    {"nt_out", 0, CH_OPT_FIELD(nt_out), -1, 1},
    {"group_n", 0, CH_OPT_FIELD(group_n), 0, 64},
    {"splitk", 0, CH_OPT_FIELD(splitk), 0, 1},
    {"gemm_rows", 0, CH_OPT_FIELD(gemm_rows), 0, 1},
    {"wide_kernel", 0, CH_OPT_FIELD(wide_kernel), 0, 1},
    {"graph_max_batch", 0, CH_OPT_FIELD(graph_max_batch), 0, 1 << 16},
    {"train_chains", 0, CH_OPT_FIELD(train_chains), 1, 2},
    {"train_chain_min_rows", 2, CH_OPT_FIELD(train_chain_min_rows), 0, (int64_t)1 << 40},
    {"train_prune_last", 1, CH_OPT_FIELD(train_prune_last), 0, 1},
    {"train_batched_grads", 1, CH_OPT_FIELD(train_batched_grads), 0, 1},
};
#undef CH_OPT_FIELD
const OptionRef *find_option(const char *key) {
    if (!key) return nullptr;
    for (const OptionRef &o : g_options)
        if (std::string(o.key) == key) return &o;
    return nullptr;
}
}  // namespace

extern "C" int ch_model_set_option(ch_model *m, const char *key, int64_t value) {
    CH_REQUIRE(m != nullptr, "set_option: null model");
    const OptionRef *o = find_option(key);
    if (!o) {
        ch_set_error(std::string("invalid argument: set_option: unknown key '") + (key ? key : "(null)") + "'");
        return 2;
    }
    if (value < o->lo || value > o->hi) {
        ch_set_error(std::string("invalid argument: set_option: '") + key + "' must be in [" + std::to_string(o->lo) + ", " +
                     std::to_string(o->hi) + "], got " + std::to_string(value));
        return 2;
    }
#ifndef CH_EXPERIMENTS
    const std::string k = key;
    if (value != 0 && (k == "fused_adapter" || k == "pp_sched" || k == "gemm_rows" || k == "wide_kernel")) {
        ch_set_error("set_option: '" + k + "' selects an experiment kernel that is not part of this build (CH_BUILD_EXPERIMENTS=1)");
        return 2;
    }
#endif
    if (std::string(key) == "splitk" && value != 0) {   // the tail's slabs + tickets, one set per chain: allocated on first use
        for (int i = 0; i < CH_MAX_STREAMS; ++i) {
            if (m->splitk_ws[i]) continue;
            CH_CHECK_HIP(hipMalloc((void **)&m->splitk_ws[i], CH_SPLITK_WS_BYTES));
            m->allocs.push_back(m->splitk_ws[i]);
            CH_CHECK_HIP(hipMalloc((void **)&m->splitk_cnt[i], CH_SPLITK_CNT_BYTES));
            m->allocs.push_back(m->splitk_cnt[i]);
            CH_CHECK_HIP(hipMemset(m->splitk_cnt[i], 0, CH_SPLITK_CNT_BYTES));
            m->bytes += CH_SPLITK_WS_BYTES + CH_SPLITK_CNT_BYTES;
        }
    }
    void *f = o->field(m);
    if (o->kind == 0) *(int *)f = (int)value;
    else if (o->kind == 1) *(bool *)f = value != 0;
    else *(int64_t *)f = value;
    if (std::string(key) == "graph_max_batch") m->graph_max_batch = std::min(m->graph_max_batch, m->cfg.max_batch);
    drop_graphs(m);   // a captured chain has the old setting baked in
    return 0;
}

extern "C" int ch_model_get_option(ch_model *m, const char *key, int64_t *value) {
    CH_REQUIRE(m != nullptr && value != nullptr, "get_option: null model / value");
    if (key && std::string(key) == "graph_replays") { *value = m->graph_replays; return 0; }      // read-only counters (tests, bench)
    if (key && std::string(key) == "last_chains") { *value = m->last_chains; return 0; }          // chains of the latest ch_encode launch sequence
    if (key && std::string(key) == "graph_captures") { *value = m->graph_captures; return 0; }
    const OptionRef *o = find_option(key);
    if (!o) {
        ch_set_error(std::string("invalid argument: get_option: unknown key '") + (key ? key : "(null)") + "'");
        return 2;
    }
    void *f = o->field(m);
    *value = o->kind == 0 ? (int64_t) * (int *)f : o->kind == 1 ? (int64_t) * (bool *)f : *(int64_t *)f;
    return 0;
}

extern "C" double ch_model_flops_per_image(const ch_model *m) {
    if (!m) return 0.0;
    const ch_model_config &c = m->cfg;
    const double N = m->ntok, D = c.dim, M = c.ffn, b = c.adapter_dim, L = c.layers;
    const double patch = 2.0 * m->np * D * 3.0 * c.patch * c.patch;
    const double layer = 8.0 * N * D * D + 4.0 * N * D * M + 4.0 * N * N * D + 8.0 * N * D * b;
    double total = patch + L * layer + 2.0 * D * c.nbit;
    if (m->prune_last && m->ln_fold && c.adapter_dim > 0 && !m->use_fused_adapter) {
        // final layer: qkv on all rows, everything after it on the 1 + Q rows the head reads
        const double nq = 1 + c.ncontext;
        total += -layer + 6.0 * N * D * D + 4.0 * nq * N * D + 2.0 * nq * D * D + 4.0 * nq * D * M + 8.0 * nq * D * b;
    }
    return total;
}

namespace {

// the launch chain of one ch_encode call on stream s (kernels + the fork / join events of the micro-batch chains: capturable)
int encode_impl(ch_model *m, const void *images, int image_dtype, int B, float *out_codes, uint64_t *out_packed, float *out_logits_cont,
                float *out_logits_bin, float *out_logits_concept, float *out_hash_features, float *out_image_features,
                float *out_concept_attn, bool all_layers, hipStream_t s) {
    if (int e = run_encoder(m, images, image_dtype, B, m->cfg.layers, s, out_concept_attn, all_layers, true)) return e;
    const ch_model_config &c = m->cfg;
    const bool pruned = m->prune_last && m->ln_fold && c.adapter_dim > 0 && !m->use_fused_adapter;  // as run_chain decides
    HeadParams p{};
    p.H = pruned ? m->Hc : m->H; p.B = B; p.ntok = pruned ? 1 + c.ncontext : m->ntok; p.D = c.dim; p.Q = c.ncontext; p.nbit = c.nbit; p.C = c.nclass; p.P = c.proj_dim;
    p.hash_pe = m->hash_pe; p.hash_fc = m->hash_fc; p.bn_scale = m->bn_scale; p.bn_shift = m->bn_shift;
    p.center_l2 = m->center_l2; p.center_bin = m->center_bin; p.concept_pe = m->concept_pe;
    p.concept_cent_l2 = m->concept_cent_l2; p.post_w = m->post_w; p.post_b = m->post_b; p.vis_proj = m->vis_proj;
    p.ln_eps = c.ln_eps;
    p.out_codes = out_codes; p.out_packed = out_packed; p.out_logits_cont = out_logits_cont;
    p.out_logits_bin = out_logits_bin; p.out_logits_concept = out_logits_concept;
    p.out_hash_features = out_hash_features; p.out_image_features = out_image_features;
    p.ws_xn = m->head_xn; p.ws_cls = m->head_cls;
    mark(m, 0, CH_CAT_HEAD, 2.0 * B * (double)c.dim * c.nbit, s);
    if (int e = ch_head(p, s)) return e;
    mark(m, 0, CH_CAT_END, 0.0, s);
    return 0;
}

int ensure_bytes(ch_model *m, void **buf, size_t *have, size_t need) {
    if (*have >= need) return 0;
    drop_graphs(m);                      // cached graphs hold the old pointer
    if (*buf) {
        CH_CHECK_HIP(hipDeviceSynchronize());
        for (void *&p : m->allocs)
            if (p == *buf) p = nullptr;
        CH_CHECK_HIP(hipFree(*buf));
        m->bytes -= *have;
    }
    CH_CHECK_HIP(hipMalloc(buf, need));
    m->allocs.push_back(*buf);
    m->bytes += need;
    *have = need;
    return 0;
}

// B <= graph_max_batch: images -> staging, one graph replay, requested outputs <- staging
int encode_graph(ch_model *m, const void *images, int image_dtype, int B, void *const outs[8], bool all_layers, hipStream_t s) {
    const ch_model_config &c = m->cfg;
    const size_t img_bytes = (size_t)B * 3 * c.image_size * c.image_size * (image_dtype == 0 ? 4 : 2);
    const size_t np = (size_t)m->np;
    const size_t out_bytes[8] = {sizeof(float) * B * c.nbit, sizeof(uint64_t) * B * ((c.nbit + 63) / 64), sizeof(float) * B * c.nclass,
                                 sizeof(float) * B * c.nclass, sizeof(float) * c.ncontext * B * c.nclass, sizeof(float) * B * c.ncontext * c.dim,
                                 sizeof(float) * B * c.proj_dim,
                                 sizeof(float) * (all_layers ? c.layers : 1) * B * c.heads * c.ncontext * np};
    int mask = 0;
    for (int i = 0; i < 8; ++i)
        if (outs[i]) mask |= 1 << i;
    const size_t cap = (size_t)m->graph_max_batch;
    if (int e = ensure_bytes(m, &m->g_in, &m->g_in_bytes, cap * 3 * c.image_size * c.image_size * 4)) return e;
    for (int i = 0; i < 8; ++i)
        if (outs[i])
            if (int e = ensure_bytes(m, &m->g_out[i], &m->g_out_bytes[i], out_bytes[i] / B * cap)) return e;
    CH_CHECK_HIP(hipMemcpyAsync(m->g_in, images, img_bytes, hipMemcpyDeviceToDevice, s));
    auto staged = [&](int i) { return outs[i] ? m->g_out[i] : nullptr; };
    auto chain = [&](hipStream_t st) {
        return encode_impl(m, m->g_in, image_dtype, B, (float *)staged(0), (uint64_t *)staged(1), (float *)staged(2), (float *)staged(3),
                           (float *)staged(4), (float *)staged(5), (float *)staged(6), (float *)staged(7), all_layers, st);
    };
    const auto key = std::make_tuple(B, image_dtype, mask, all_layers ? 1 : 0);
    auto it = m->graphs.find(key);
    if (it == m->graphs.end()) {
        // first call of this shape: run it eagerly (this call's result; also performs every once-per-device function attribute
        // setting outside the capture), then capture the same chain on the library's own stream and keep the instantiated graph
        if (int e = chain(s)) return e;
        if (!m->cap_stream) CH_CHECK_HIP(hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking));
        ch_model::GraphEntry ge;
        CH_CHECK_HIP(hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeThreadLocal));
        const int e = chain(m->cap_stream);
        const hipError_t he = hipStreamEndCapture(m->cap_stream, &ge.graph);
        if (e) {
            if (ge.graph) (void)hipGraphDestroy(ge.graph);
            return e;
        }
        CH_CHECK_HIP(he);
        CH_CHECK_HIP(hipGraphInstantiate(&ge.exec, ge.graph, nullptr, nullptr, 0));
        m->graphs[key] = ge;
        m->graph_captures++;
    } else {
        CH_CHECK_HIP(hipGraphLaunch(it->second.exec, s));
        m->graph_replays++;
    }
    for (int i = 0; i < 8; ++i)
        if (outs[i]) CH_CHECK_HIP(hipMemcpyAsync(outs[i], m->g_out[i], out_bytes[i], hipMemcpyDeviceToDevice, s));
    return 0;
}

}  // namespace

extern "C" int ch_encode(ch_model *m, const void *images, int32_t image_dtype, int32_t B, float *out_codes,
                         uint64_t *out_packed, float *out_logits_cont, float *out_logits_bin, float *out_logits_concept,
                         float *out_hash_features, float *out_image_features, float *out_concept_attn,
                         int32_t concept_attn_all_layers, void *stream) {
    CH_REQUIRE(m != nullptr && images != nullptr && out_codes != nullptr, "null model / images / out_codes");
    CH_REQUIRE(image_dtype == 0 || image_dtype == 1, "image_dtype must be 0 (fp32) or 1 (bf16)");
    CH_REQUIRE(B >= 1 && B <= m->cfg.max_batch, "batch outside [1, max_batch]");
    CH_REQUIRE(!out_logits_concept || m->concept_cent_l2, "model has no concept classifier (concept_ce.centroids)");
    CH_REQUIRE(!out_image_features || m->vis_proj, "model has no post_layernorm / visual_projection");
    hipStream_t s = (hipStream_t)stream;
    if (m->graph_max_batch > 0 && B <= m->graph_max_batch && !m->prof_on) {
        void *const outs[8] = {out_codes, out_packed, out_logits_cont, out_logits_bin, out_logits_concept, out_hash_features,
                               out_image_features, out_concept_attn};
        return encode_graph(m, images, image_dtype, B, outs, concept_attn_all_layers != 0, s);
    }
    return encode_impl(m, images, image_dtype, B, out_codes, out_packed, out_logits_cont, out_logits_bin, out_logits_concept,
                       out_hash_features, out_image_features, out_concept_attn, concept_attn_all_layers != 0, s);
}

extern "C" int ch_encode_hidden(ch_model *m, const void *images, int32_t image_dtype, int32_t B, int32_t layer,
                                float *out_hidden, void *stream) {
    CH_REQUIRE(m != nullptr && images != nullptr && out_hidden != nullptr, "null model / images / out_hidden");
    CH_REQUIRE(B >= 1 && B <= m->cfg.max_batch, "batch outside [1, max_batch]");
    CH_REQUIRE(layer >= 0 && layer <= m->cfg.layers, "layer outside [0, layers]");
    hipStream_t s = (hipStream_t)stream;
    if (int e = run_encoder(m, images, image_dtype, B, layer, s)) return e;
    CH_CHECK_HIP(hipMemcpyAsync(out_hidden, m->H, sizeof(float) * (size_t)B * m->ntok * m->cfg.dim,
                                hipMemcpyDeviceToDevice, s));
    return 0;
}

// test tap: copy the first nbytes of one workspace buffer (as the last ch_encode / ch_encode_hidden left it) to `out`
extern "C" int ch_debug_copy_buffer(ch_model *m, int32_t which, void *out, int64_t nbytes, void *stream) {
    CH_REQUIRE(m != nullptr && out != nullptr && nbytes >= 0, "debug_copy_buffer: null pointer");
    const int64_t rows = m->rows_alloc, D = m->cfg.dim;
    const void *src = nullptr;
    int64_t size = 0;
    switch (which) {
        case 0: src = m->H; size = rows * D * 4; break;
        case 1: src = m->Xn; size = rows * D * 2; break;
        case 2: src = m->QKV; size = rows * 3 * D * 2; break;
        case 3: src = m->AO; size = rows * D * 2; break;
        case 4: src = m->A; size = rows * D * 2; break;
        case 5: src = m->AD; size = rows * std::max(m->bpad, 128) * 2; break;
        case 6: src = m->F1; size = rows * (int64_t)m->cfg.ffn * 2; break;
        default: CH_REQUIRE(false, "debug_copy_buffer: which must be 0..6 (H, Xn, QKV, AO, A, AD, F1)");
    }
    CH_REQUIRE(nbytes <= size, "debug_copy_buffer: more bytes requested than the buffer holds");
    CH_CHECK_HIP(hipMemcpyAsync(out, src, (size_t)nbytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

extern "C" int ch_pack_sign(const float *codes, int64_t rows, int32_t nbit, float threshold, uint64_t *out_packed,
                            void *stream) {
    CH_REQUIRE(rows >= 0 && nbit > 0, "pack_sign: rows must be >= 0 and nbit > 0");
    CH_REQUIRE(rows == 0 || (codes && out_packed), "pack_sign: null pointer");
    return ch_pack_sign_launch(codes, rows, nbit, threshold, out_packed, (hipStream_t)stream);
}

extern "C" int ch_model_profile_begin(ch_model *m, int32_t max_launches) {
    CH_REQUIRE(m != nullptr && max_launches > 0, "profile_begin: null model or non-positive capacity");
    for (auto &P : m->prof) {
        while (P.ev.size() < (size_t)max_launches + 1) {
            hipEvent_t e, a, b;
            CH_CHECK_HIP(hipEventCreate(&e));
            CH_CHECK_HIP(hipEventCreate(&a));
            CH_CHECK_HIP(hipEventCreate(&b));
            P.ev.push_back(e);
            P.kstart.push_back(a);
            P.kstop.push_back(b);
        }
        P.cat.assign(P.ev.size(), CH_CAT_END);
        P.flops.assign(P.ev.size(), 0.0);
        P.exact.assign(P.ev.size(), 0);
        P.n = 0;
    }
    m->prof_on = true;
    return 0;
}

extern "C" int ch_model_profile_end(ch_model *m, int32_t ncat, double *ms_per_cat, int64_t *launches_per_cat, double *flops_per_cat) {
    CH_REQUIRE(m != nullptr && ms_per_cat && launches_per_cat && flops_per_cat, "profile_end: null pointer");
    CH_REQUIRE(ncat >= CH_NCAT, "profile_end: the caller's arrays hold fewer than CH_NCAT entries (built against an older header?)");
    m->prof_on = false;
    for (int i = 0; i < CH_NCAT; ++i) {
        ms_per_cat[i] = 0.0;
        launches_per_cat[i] = 0;
        flops_per_cat[i] = 0.0;
    }
    for (auto &P : m->prof) {
        if (P.n < 2) {
            P.n = 0;
            continue;
        }
        CH_CHECK_HIP(hipEventSynchronize(P.ev[P.n - 1]));
        for (size_t j = 0; j + 1 < P.n; ++j) {
            const int cat = P.cat[j];
            if (cat == CH_CAT_END) continue;  // gap between two ch_encode calls
            float ms = 0.f;
            if (P.exact[j])   // begin -> end of the dispatch itself; otherwise launch-to-launch interval (small LDS-free kernels)
                CH_CHECK_HIP(hipEventElapsedTime(&ms, P.kstart[j], P.kstop[j]));
            else
                CH_CHECK_HIP(hipEventElapsedTime(&ms, P.ev[j], P.ev[j + 1]));
            ms_per_cat[cat] += ms;
            launches_per_cat[cat] += 1;
            flops_per_cat[cat] += P.flops[j];
        }
        P.n = 0;
    }
    return 0;
}

// split-K workspace of the debug taps (off by default so that the 256x256 kernel stays bit-identical to the 128x128 one)
static bool g_debug_splitk = false;
static float *g_debug_ws = nullptr;
static unsigned *g_debug_cnt = nullptr;
static int debug_attach_splitk(GemmParams &p) {
    if (!g_debug_splitk) return 0;
    if (!g_debug_ws) {
        CH_CHECK_HIP(hipMalloc((void **)&g_debug_ws, CH_SPLITK_WS_BYTES));
        CH_CHECK_HIP(hipMalloc((void **)&g_debug_cnt, CH_SPLITK_CNT_BYTES));
        CH_CHECK_HIP(hipMemset(g_debug_cnt, 0, CH_SPLITK_CNT_BYTES));
    }
    p.splitk_ws = g_debug_ws;
    p.splitk_cnt = g_debug_cnt;
    p.force_split = 1;
    return 0;
}
extern "C" void ch_debug_set_gemm_splitk(int32_t on) { g_debug_splitk = on != 0; }

// ---- test / bench tap: one GEMM launch on caller buffers (tests/test_gemm_gpu.py, tools/gemm_bench.py) -----------------
extern "C" int ch_debug_gemm(int32_t variant, const void *X, int64_t X_rows_alloc, const void *W, const float *bias,
                             int32_t M, int32_t N, int32_t K, int32_t epi, void *out_bf16, int32_t ldo, float *resid,
                             int32_t ldr, const float *scale_ptr, const void *addend, void *stream) {
    CH_REQUIRE(X && W, "debug_gemm: null operand");
    CH_REQUIRE(epi >= EPI_BIAS && epi <= EPI_SCALE_RESID, "debug_gemm: epilogue must be one of the non-patch modes");
    GemmParams p{};
    p.X = (const bf16_t *)X; p.W = (const bf16_t *)W; p.M = M; p.N = N; p.K = K; p.X_rows_alloc = X_rows_alloc;
    p.bias = bias; p.out_bf16 = (bf16_t *)out_bf16; p.ldo = ldo; p.resid = resid; p.ldr = ldr; p.scale_ptr = scale_ptr; p.addend = (const bf16_t *)addend; p.ld_addend = N;
    hipStream_t s = (hipStream_t)stream;
    if (int e = debug_attach_splitk(p)) return e;
    if (variant == 1) return ch_gemm_bf16_v1(p, epi, s);
    if (variant == 2) return ch_gemm_bf16_pp(p, epi, s);
    if (variant == 4 || variant == 8) {
        p.pp_sched = variant == 4 ? 1 : 2;
        return ch_gemm_bf16_pp(p, epi, s);
    }
    if (variant == 3) return ch_gemm_bf16_dp(p, epi, s);
    if (variant == 10) return ch_gemm_bf16_wide(p, epi, s);
    if (variant == 9) return ch_gemm_bf16_rows(p, epi, s);
    if (variant == 5) return ch_gemm_bf16_ppp(p, epi, s);
    if (variant == 6) return ch_gemm_bf16_pq(p, epi, s);
    if (variant == 7) return ch_gemm_bf16_r4(p, epi, s);
    if (variant >= 21 && variant <= 29) return ch_gemm_bf16_pp_dbg(p, variant - 20, s);  // timing-only / stamped builds
    if (variant >= 41 && variant <= 47) return ch_gemm_bf16_wide_dbg(p, variant - 40, s);  // timing-only builds of the 256x384 kernel
    return ch_gemm_bf16(p, epi, s);
}
extern "C" int ch_debug_gemm_ln(int32_t variant, const void *X, int64_t X_rows_alloc, const void *W, const float *bias,
                                int32_t M, int32_t N, int32_t K, int32_t epi, void *out_bf16, int32_t ldo, float *resid,
                                int32_t ldr, const float *scale_ptr, const void *addend, const float *stats_in,
                                const float *fold_c, float ln_eps, float *stats_out, void *hb_out, void *stream) {
    CH_REQUIRE(X && W, "debug_gemm_ln: null operand");
    CH_REQUIRE(epi >= EPI_BIAS_STATS && epi <= EPI_FOLD_GELU, "debug_gemm_ln: epilogue must be one of the LayerNorm-fold modes");
    GemmParams p{};
    p.X = (const bf16_t *)X; p.W = (const bf16_t *)W; p.M = M; p.N = N; p.K = K; p.X_rows_alloc = X_rows_alloc;
    p.bias = bias; p.out_bf16 = (bf16_t *)out_bf16; p.ldo = ldo; p.resid = resid; p.ldr = ldr; p.scale_ptr = scale_ptr;
    p.addend = (const bf16_t *)addend; p.ld_addend = N;
    p.stats_in = stats_in; p.fold_c = fold_c; p.ln_eps = ln_eps; p.stats_out = stats_out; p.hb_out = (bf16_t *)hb_out; p.ld_hb = N;
    hipStream_t s = (hipStream_t)stream;
    if (int e = debug_attach_splitk(p)) return e;
    if (variant == 5) return ch_gemm_bf16_ppp(p, epi, s);
    if (variant == 6) return ch_gemm_bf16_pq(p, epi, s);
    if (variant == 7) return ch_gemm_bf16_r4(p, epi, s);
    if (variant == 1 || variant == 2 || variant == 4 || variant == 8 || variant == 9 || variant == 10) ch_gemm_set_variant(variant);
    const int rc = ch_gemm_bf16(p, epi, s);
    if (variant == 1 || variant == 2 || variant == 4 || variant == 8 || variant == 9 || variant == 10) ch_gemm_set_variant(0);
    return rc;
}
extern "C" void ch_debug_set_gemm_variant(int32_t v) { ch_gemm_set_variant(v); }
extern "C" int32_t ch_debug_experiments_built(void) {
#ifdef CH_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

extern "C" int ch_debug_attention(const void *qkv, int32_t B, int32_t ntok, int32_t heads, void *out, void *stream) {
    CH_REQUIRE(qkv && out, "debug_attention: null pointer");
    return ch_attention((const bf16_t *)qkv, B, ntok, heads, (bf16_t *)out, (hipStream_t)stream);
}

extern "C" int ch_debug_adapter(const void *A, float *H, int32_t M, int32_t D, int32_t b, const float *Wd, const float *bd,
                                const float *gamma, const float *beta, const void *Wu_bf16_padded, const float *bu,
                                const float *scale, void *work_wdf, float *work_c, float *work_d, int32_t dbg, void *stream) {
    // Wd [b, D] fp32, Wu [D, bpad] bf16 (already padded); work_*: caller scratch for the folded weights ([bpad, D] bf16, [bpad] x2)
    const int bpad = (int)round_up64(b, 128);
    hipStream_t s = (hipStream_t)stream;
    if (int e = ch_fold_ln(Wd, bd, gamma, beta, b, bpad, D, (bf16_t *)work_wdf, work_c, work_d, s)) return e;
    AdapterParams p{};
    p.A = (const bf16_t *)A; p.H = H; p.M = M; p.D = D; p.bpad = bpad; p.Wd = (const bf16_t *)work_wdf; p.c = work_c; p.d = work_d;
    p.Wu = (const bf16_t *)Wu_bf16_padded; p.bu = bu; p.scale = scale; p.eps = 1e-5f; p.dbg = dbg;
    return ch_adapter_fused(p, s);
}
