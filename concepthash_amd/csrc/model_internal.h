// Internal layout of ch_model (weights, workspace, launch profiler): shared by model.hip (the encode chain) and train.hip
// (the training step, which reuses the frozen backbone weights of an existing model).  Not part of the C-ABI.
#pragma once
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/concepthash_hip.h"
#include "ch_common.h"
#include "kernels.h"

constexpr int CH_MAX_STREAMS = 4;  // micro-batch chains that may run concurrently (option "streams")

struct AdapterW {
    const float *ln_w = nullptr, *ln_b = nullptr, *down_b = nullptr, *up_b = nullptr, *scale = nullptr;
    const bf16_t *down_w = nullptr, *up_w = nullptr;
    // adapter LayerNorm folded into the down projection (LN-fold chain and adapter_fused.hip)
    const bf16_t *down_wf = nullptr;
    const float *fold_c = nullptr, *fold_d = nullptr;
};
struct LayerW {
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *qkv_b, *out_b, *fc1_b, *fc2_b;
    const bf16_t *qkv_w, *out_w, *fc1_w, *fc2_w;
    // layer_norm1 / layer_norm2 folded into qkv / fc1: W' = bf16(W * gamma), c = row sums of W', d = bias + W beta
    const bf16_t *qkv_wf = nullptr, *fc1_wf = nullptr;
    const float *qkv_c = nullptr, *qkv_d = nullptr, *fc1_c = nullptr, *fc1_d = nullptr;
    AdapterW ad[2];
};

struct ch_model {
    ch_model_config cfg;
    int np = 0, ntok = 0, Kp = 0, bpad = 0;
    std::vector<void *> allocs;
    size_t bytes = 0;
    // weights
    const bf16_t *patch_w = nullptr;
    const float *pos = nullptr, *cls_pos0 = nullptr, *ctx = nullptr, *pre_w = nullptr, *pre_b = nullptr;
    const float *zero_bias = nullptr;
    std::vector<LayerW> layers;
    const float *hash_pe = nullptr, *hash_fc = nullptr, *bn_scale = nullptr, *bn_shift = nullptr;
    const float *center_l2 = nullptr, *center_bin = nullptr, *concept_pe = nullptr, *concept_cent_l2 = nullptr;
    const float *post_w = nullptr, *post_b = nullptr, *vis_proj = nullptr;
    // launch profiler (bench.py): one hipEvent before every launch + one after the last; elapsed(e[j], e[j+1]) is
    // attributed to launch j's category
    // adapter_fused.hip is correct (parity-tested) but measured slower than the three-launch chain on MI355X (200 vs 179 us
    // per call at B=256: its HBM phases and MFMA phases do not overlap) -> opt-in only
    bool use_fused_adapter = false;
    // LayerNorm folded into the consumer GEMMs: no LayerNorm launches inside the layer loop.
    // Needs adapters (their up-projection epilogue is where the bf16 copy of the residual and its row statistics are made).
    bool ln_fold = true;
    bool prof_on = false;
    struct Prof {
        std::vector<hipEvent_t> ev;          // interval form: one event in front of every launch (+ one behind the last)
        std::vector<hipEvent_t> kstart, kstop;  // exact form: the dispatch's own begin / end (CH_LAUNCH, ch_common.h)
        std::vector<char> exact;             // launch j took the pair (the GEMM and attention launch sites do)
        std::vector<int> cat;
        std::vector<double> flops;
        size_t n = 0;
    } prof[CH_MAX_STREAMS];  // one per micro-batch stream
    // option "streams" micro-batches (default 2) on as many HIP streams: memory-bound launches of one chain (adapter up-projection,
    // attention, epilogue-heavy GEMM tails) co-run with MFMA-bound launches of the other, and partly filled last rounds of
    // tiles get filled.  Rows are independent, so the outputs are bit-identical for every value.  Measured at B = 256:
    // 1 -> 2 streams +8 % images/s, 3 and 4 no better.  streams = 1 is what a per-kernel profile wants:
    // per-launch durations stop describing single kernels once launches overlap.
    int nstreams = 2;
    hipStream_t aux_stream[CH_MAX_STREAMS - 1] = {};
    hipEvent_t ev_fork = nullptr, ev_join[CH_MAX_STREAMS - 1] = {};
    // workspace
    int64_t rows_alloc = 0, prow_alloc = 0;
    float *H = nullptr;
    float *splitk_ws[CH_MAX_STREAMS] = {};      // split-K tail slabs + tickets of the 256x256 GEMM, one set per chain
    unsigned *splitk_cnt[CH_MAX_STREAMS] = {};
    // final-layer row pruning: compact fp32 copy of the residual rows the head reads, [max_batch * (1 + Q) (+pad), D]
    bool prune_last = true;
    // Serpentine launch order (option "serpentine"): every row-streaming launch of a chain walks its row
    // tiles in the direction opposite to its predecessor's, so that it starts on the rows the predecessor wrote LAST -- the
    // ones that should still be in the 256 MB Infinity Cache.  Measured: no gain (12.44 vs 12.39 ms per step) -> off.
    bool serpentine = false;
    int pp_sched = 0;  // option "pp_sched": schedule of the 256x256 GEMM (gemm_pp.hip)
    int small_kernel = 0;  // option "small_kernel": 0 = dispatcher (ring up to CH_RING_MAX_ROWS rows), 1 = 128x128x64 two-phase always, 2 = ring always
    int pp_min_k = 0;  // option "pp_min_k" (tests: sends small-K GEMMs of a small fixture to the 256x256 kernel)
    // ---- per-handle tuning / test options (ch_model_set_option; the library reads no environment variable)
    int resid_nt = 0, nt_out = 0;   // cache policy of the fp32 residual read-modify-write / of large bf16 outputs: 0 = by tensor size, 1 = on, -1 = off
    int group_n = 0;                // n-tiles per L2-resident weight group of the GEMM tile order (0 = host heuristic)
    int splitk = 0;                 // split-K tail of the 256x256 GEMM (opt-in, measured slower: DESIGN.md section 3.8)
    int gemm_rows = 0;              // whole-row kernel for N = 384 (experiments build)
    int wide_kernel = 0;            // 256x384 GEMM for N % 384 == 0 (adapter bottleneck; experiments build): 1 = wherever supported
    // ---- hipGraph replay of small batches (option "graph_max_batch", 0 = off): at batch 8 .. 64 the ~230 launches of a two-chain
    // ch_encode cost more host time than GPU time, so a call with B <= graph_max_batch stages its images into `g_in`, replays the chain
    // as ONE captured graph (keyed by batch, image dtype, requested outputs, attention-tap layout) and copies the requested outputs out
    // of the staging buffers.  Any ch_model_set_option drops the cached graphs.  Off in the C-ABI default (a caller may be capturing
    // ch_encode into a graph of its own); the Python wrapper switches it on.
    int graph_max_batch = 0;
    hipStream_t cap_stream = nullptr;
    struct GraphEntry {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
    };
    std::map<std::tuple<int, int, int, int>, GraphEntry> graphs;   // (B, image dtype, output mask, all-layers tap)
    void *g_in = nullptr;
    size_t g_in_bytes = 0;
    void *g_out[8] = {};          // codes, packed, logits_cont, logits_bin, logits_concept, hash_features, image_features, concept_attn
    size_t g_out_bytes[8] = {};
    int64_t graph_replays = 0, graph_captures = 0;
    // read by ch_trainer_create from the model it is created on
    int train_chains = 2;           // two micro-batch chains when each has >= train_chain_min_rows rows (round 4: -4.6 % at batch 256, -5.3 % at 128)
    int last_chains = 0;                // (read-only option "last_chains")
    bool chain_auto = true;             // ch_encode with streams = 2: fall back to one chain for the two measured exceptions (model.hip run_encoder)
    int64_t train_chain_min_rows = 0;   // 0 = by tile rounds (train.hip: ch_train_forward); > 0 = explicit rows-per-chain threshold
    bool train_prune_last = true;
    bool train_batched_grads = true;  // one reduction launch per adapter, one gradient-assembly launch pair per step (train.hip)
    float *Hc = nullptr;
    float *head_xn = nullptr, *head_cls = nullptr;  // head.hip: left operands of the two dense optional outputs
    float *statsA = nullptr, *statsH = nullptr;  // [rows, D/64, 2] partial (sum, sumsq) of the rows of A / of bf16(H) in Xn
    bf16_t *Xn = nullptr, *QKV = nullptr, *AO = nullptr, *A = nullptr, *AD = nullptr, *F1 = nullptr, *PATCH = nullptr;
};

