/*
 * concepthash_hip.h -- C-ABI of libconcepthash_hip.so (MI355X / gfx950).
 *
 * The reference (kamwoh/concepthash) is pure Python/PyTorch and exposes NO FFI/operator interface (SURVEY.md F1,
 * section 8b); its boundary is Python import paths.  This header is therefore the boundary *defined by this
 * project*: each entry point names the reference function(s) it replaces (paths relative to the reference root) and
 * is what a maintainer's ctypes stub binds.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; ch_last_error() gives the message (thread local);
 *     no C++ exception crosses the boundary.
 *   - all data pointers are DEVICE pointers owned by the caller unless a parameter says "host".
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls only enqueue work and never
 *     synchronise, allocate or free device memory (graph-capture safe) -- except ch_model_create/destroy.
 *   - integers are bit-exact by contract; floating point tolerances are stated in DESIGN.md / tests.
 *   - no hidden global state: the library reads NO environment variable; every tuning / test knob is state of one opaque handle
 *     (ch_model_set_option), the only process-wide switches are the test taps of concepthash_hip_debug.h.
 */
#ifndef CONCEPTHASH_HIP_H
#define CONCEPTHASH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): ch_encode / ch_train_forward take the layout of the concept-attention tap as an argument (ch_model_set_concept_attn_layers
 * is gone), ch_model_profile_end takes the capacity of the caller's arrays, ch_model_set_option / ch_model_get_option replace every
 * environment variable the library used to read.
 * 3 (late round 4): ch_image_desc gained `stride` and `flip` (56 bytes: the training transforms through ch_preprocess), ch_preprocess takes
 * `max_taps` (kernel selection); ch_tensor.data may be a device pointer; new entry points ch_jpeg_* (decode split), ch_io_file_sizes / ch_io_read_files (batch file reads). */
#define CH_ABI_VERSION 3

typedef struct ch_model ch_model; /* opaque: weights (bf16/fp32, device) + activation workspace */

typedef struct ch_model_config {
    int32_t image_size;  /* H == W of the input, e.g. 224 (pretrain resolution; no pos-embed interpolation) */
    int32_t patch;       /* 14 / 16 / 32 */
    int32_t dim;         /* D: ViT width (multiple of 64) */
    int32_t layers;      /* L */
    int32_t heads;       /* attention heads; dim / heads must be 64 */
    int32_t ffn;         /* MLP hidden width */
    int32_t adapter_dim; /* adapter bottleneck b (0 = no adapters) */
    int32_t ncontext;    /* Q: number of concept tokens */
    int32_t nbit;        /* total hash bits = Q * bits-per-concept */
    int32_t nclass;      /* C */
    int32_t proj_dim;    /* P: CLIP projection dim (concept-token generator width) */
    int32_t center_dim;  /* width of the `center` buffer (512 for CLIP text features) */
    int32_t upt_heads;   /* heads of the concept-token generator MHA (config upt_config.num_heads, 8) */
    int32_t act;         /* 0 = quick_gelu (OpenAI CLIP), 1 = exact gelu */
    int32_t max_batch;   /* largest B accepted by ch_encode (workspace is sized for it) */
    float ln_eps;        /* LayerNorm eps (1e-5) */
    float bn_eps;        /* BatchNorm1d eps (1e-5) */
} ch_model_config;

/* One named fp32 tensor, in host OR device memory (the library copies with hipMemcpyDefault: a model that already sits on the GPU
 * is ingested without a round trip through the host).  `name` is the reference state_dict key, e.g.
 * "backbone.vision_model.encoder.layers.0.self_attn.q_proj.weight", "hash_fc.weight", "hash_bn.running_var". */
typedef struct ch_tensor {
    const char *name;
    const float *data; /* host or device, fp32, contiguous, PyTorch layout */
    int64_t numel;
} ch_tensor;

int ch_abi_version(void);
const char *ch_last_error(void);

/* ---------------------------------------------------------------------------------------------------------------
 * Encode
 * ------------------------------------------------------------------------------------------------------------- */

/* Replaces: model construction + BaseTrainer.load_model_state (trainers/base.py:195-197) + .to(device)
 * (trainers/base.py:57-71) for models.arch.coop.LGHWithFixedPrompt.  Copies/convert weights to the device (GEMM
 * operands -> bf16, everything else fp32), and folds the input-independent parts of the forward once, on the GPU:
 * forward_hash_query (models/arch/coop.py:413-427) -> concept tokens; get_center (coop.py:624-625) -> projected,
 * l2-normalised centres; BatchNorm1d eval (coop.py:559) -> per-bit affine.  Unknown/missing names are an error;
 * aliases (`adapter_params.*`, `trainable_params.*`) are ignored. */
int ch_model_create(const ch_model_config *cfg, const ch_tensor *tensors, int32_t ntensors, ch_model **out);
void ch_model_destroy(ch_model *m);
/* bytes of device memory held by the model (weights + workspace) */
size_t ch_model_device_bytes(const ch_model *m);

/* Tuning / test options of ONE model handle ("no hidden global state except an opaque handle").  Every call
 * made on the handle afterwards uses the new value; outputs are bit-identical for every setting except "ln_fold" (rounding points
 * move) and "splitk" (summation order).  Unknown keys and out-of-range values are errors.
 *   "streams"       1..4   micro-batch launch chains of ch_encode on as many HIP streams (default 2; 1 = what a per-kernel profile wants)
 *   "ln_fold"       0/1    LayerNorm folded into the consumer GEMMs (default 1)
 *   "prune_last"    0/1    final layer past its attention on the rows the hashing head reads only (default 1)
 *   "pp_min_k"      >= 0   smallest K that goes to the 256x256 ping-pong GEMM (0 = dispatcher default: 512 and >= 128 tiles)
 *   "resid_nt"      -1/0/1 non-temporal read-modify-write of the fp32 residual: off / by tensor size (default) / on
 *   "nt_out"        -1/0/1 non-temporal stores of large bf16 GEMM outputs: off / by tensor size (default) / on
 *   "group_n"       >= 0   n-tiles per L2-resident weight group of the GEMM tile order (0 = host heuristic)
 *   "graph_max_batch" >= 0 ch_encode calls with B <= this value replay the launch chain as ONE captured hipGraph (images staged into a library
 *                          buffer, requested outputs copied out of staging buffers; first call per (B, dtype, output set) runs eagerly and
 *                          captures).  0 (default) = off: ch_encode then only enqueues kernels and is itself capturable by the caller.
 *                          Bit-identical outputs; measured NEUTRAL on MI355X (batch 8: 2.07 vs 2.09 ms -- small batches are bound by the
 *                          latency of one GEMM tile per launch on the GPU, not by host launch time), hence opt-in.
 *                          (read-only keys of ch_model_get_option: "graph_replays", "graph_captures", "last_chains")
 *   "splitk"        0/1    split-K tail of the 256x256 GEMM (default 0; allocates 64 MiB per chain on first use)
 *   "serpentine"    0/1    alternate the row direction of consecutive launches (default 0)
 *   "chain_auto"    0/1    with "streams" = 2: use ONE chain for batches of fewer than 5,600 token rows (batch <= 27 of ViT-B/16), where it
 *                          measures 2-5 % faster.  Default 1; 0 = always split.  Outputs are bit-identical.
 *   "small_kernel"  0..2   GEMMs of the 128x128 path: 0 = the four-stage ring kernel up to 8,192 rows (batch <= 40 of ViT-B/16: -7 % per
 *                          step at batch 8), the two-phase kernel above; 1 = two-phase always; 2 = ring always.  Outputs are bit-identical.
 *   "pp_sched", "fused_adapter", "gemm_rows", "wide_kernel"  experiment kernels: non-zero values need the experiments build
 *   "train_chains" 1..2, "train_chain_min_rows", "train_prune_last" 0/1, "train_batched_grads" 0/1   read by ch_trainer_create from the model it is created on */
int ch_model_set_option(ch_model *m, const char *key, int64_t value);
int ch_model_get_option(ch_model *m, const char *key, int64_t *value);

/* Replaces: LGHWithFixedPrompt.forward (models/arch/coop.py:524-598) as driven by
 * COOPTrainer.compute_features_one_batch (trainers/coop.py:59-71), eval mode.
 *   images        [B,3,H,W] NCHW, fp32 (image_dtype 0) or bf16 (image_dtype 1)
 *   out_codes     [B,nbit] fp32 pre-sign codes ("codes", coop.py:559)                      (required)
 *   out_packed    [B,W] uint64, W = ceil(nbit/64); bit i = codes[i] > 0 (little endian)   (optional, NULL)
 *   out_logits_cont / out_logits_bin [B,C] fp32 (coop.py:573-580)                          (optional)
 *   out_logits_concept [Q,B,C] fp32 (coop.py:269-276)                                      (optional)
 *   out_hash_features  [B,Q,D] fp32 raw last-layer concept-token states (coop.py:503-509) (optional)
 *   out_image_features [B,P] fp32 pooled CLS -> post-LN -> visual_projection (coop.py:498-501) (optional)
 *   out_concept_attn   [B,heads,Q,Np] fp32 last-layer attention of the Q concept tokens over the Np patch tokens
 *                      = attn_cache[-1][:, :, -Q:, 1:-Q] (coop.py:481-482; consumer models/loss/coop.py:164-176)  (optional)
 *   concept_attn_all_layers  layout of out_concept_attn, part of THIS call (no sticky state on the handle): 0 = the last layer only,
 *                      [B,heads,Q,Np]; 1 = EVERY layer, [L,B,heads,Q,Np], layer l = attn_cache[l][:, :, -Q:, 1:-Q] -- what the
 *                      `avg_attn` form of the attention-diversity loss averages (models/loss/coop.py:164-167,
 *                      `torch.stack(outputs['attn_cache']).mean(0)`) and the per-layer visualisations read
 *                      (models/arch/coop.py:481-482), without the (B, heads, N, N) maps ever being written.
 * B must be in [1, max_batch]. */
int ch_encode(ch_model *m, const void *images, int32_t image_dtype, int32_t B, float *out_codes,
              uint64_t *out_packed, float *out_logits_cont, float *out_logits_bin, float *out_logits_concept,
              float *out_hash_features, float *out_image_features, float *out_concept_attn, int32_t concept_attn_all_layers,
              void *stream);

/* Parity tap (tests only): run the encoder for `layer` layers (0 = embeddings + concept tokens + pre-LN) and copy the
 * fp32 residual stream [B*N, D], N = 1 + patches + Q, to out_hidden.  Mirrors `image_hidden_states[layer]`
 * (models/arch/coop.py:474-486). */
int ch_encode_hidden(ch_model *m, const void *images, int32_t image_dtype, int32_t B, int32_t layer,
                     float *out_hidden, void *stream);

/* ---- Training step of the adapters -----------------------------------------------------------
 * Replaces, for everything that touches the [B*N, *] activations: the forward and `loss.backward()` of
 * COOPTrainer.train_one_batch (trainers/coop.py:107-131) through CLIPEncoderLayerWithAdapter.forward
 * (models/layers/adapter.py:127-177) and Adapter.forward (:46-60), with the backbone frozen and the adapters trainable
 * (trainers/base.py:133-152, configs/model/concept_hash_final_v1_nosa_apt.yaml `backbone_lr_scale: 0`).  The concept-token
 * generator (4 tokens), the hashing head on [B,Q,D] and the loss stay with the caller's autograd, which passes
 * `concept_tokens` in, takes `hash_features` out, and exchanges their gradients.
 *
 * Adapter parameter arena (fp32, device, CALLER-owned; the optimizer updates it in place): adapters in (layer, adapter 1, adapter
 * 2) order, each  [adapter_layer_norm.weight D][.bias D][down_proj.weight b*D][down_proj.bias b][up_proj.weight D*b]
 * [up_proj.bias D][scale 1];  the gradient arena has the same layout and is OVERWRITTEN by ch_train_backward. */
typedef struct ch_trainer ch_trainer;
/* total floats of the adapter arena of this model (0 without adapters) */
int64_t ch_adapter_arena_numel(const ch_model *m);
/* Shares the frozen weights of `m` (which must outlive the trainer); allocates the per-layer saved activations for max_batch.
 * Reads the model's "train_*" options (ch_model_set_option) once, here. */
int ch_trainer_create(ch_model *m, int32_t max_batch, float *params, float *grads, ch_trainer **out);
void ch_trainer_destroy(ch_trainer *t);
int64_t ch_trainer_bytes(const ch_trainer *t);
/* Re-derive the bf16 / LayerNorm-folded / transposed working copies from the parameter arena: call after every optimizer step. */
int ch_trainer_refresh(ch_trainer *t, void *stream);
/* Forward in training mode (adapter dropout 0, as the reference configs have it).
 *   concept_tokens    [Q,D] fp32 device: forward_hash_query() output (models/arch/coop.py:413-427), before pre_layrnorm
 *   out_hash_features [B,Q,D] fp32: last-layer states of the concept tokens (coop.py:503-509)
 *   out_cls           [B,D] fp32 last-layer CLS states (optional, NULL)
 *   out_concept_attn  [B,heads,Q,Np] fp32 last-layer attention of the concept tokens over the patch tokens
 *                     = attn_cache[-1][:, :, -Q:, 1:-Q] (coop.py:481-482), what the attention-diversity term of the loss reads
 *                     (models/loss/coop.py:164-189)                                                    (optional, NULL)
 *   concept_attn_all_layers  as in ch_encode: 1 = out_concept_attn is [L,B,heads,Q,Np]; the trainer remembers the layout of its last
 *                     forward, and the matching ch_train_backward takes d_concept_attn in the same one */
int ch_train_forward(ch_trainer *t, const void *images, int32_t image_dtype, int32_t B, const float *concept_tokens,
                     float *out_hash_features, float *out_cls, float *out_concept_attn, int32_t concept_attn_all_layers, void *stream);
/* Backward of the last ch_train_forward: d_hash_features [B,Q,D] fp32 and (optional, NULL) d_concept_attn fp32 in the layout that
 * forward was called with ([B,heads,Q,Np] or [L,B,heads,Q,Np]; ignored when that forward had no out_concept_attn) in;
 * adapter gradients into the gradient arena, d_concept_tokens [Q,D] fp32 out. */
int ch_train_backward(ch_trainer *t, const float *d_hash_features, const float *d_concept_attn, float *d_concept_tokens, void *stream);
/* torch.optim.SGD.step (configs/optim/sgd.yaml; maximize False) over ONE flat fp32 array -- the adapter arena: d = g + weight_decay * p;
 * buf = d on the first step, else momentum * buf + (1 - dampening) * d; p -= lr * (nesterov ? d + momentum * buf : buf).
 * n must be a multiple of 4 (ch_adapter_arena_numel is padded by the caller if not). */
int ch_sgd_step(float *params, const float *grads, float *momentum_buf, int64_t n, float lr, float momentum, float weight_decay,
                float dampening, int32_t nesterov, int32_t first_step, void *stream);
/* Launch profiler for bench.py's roofline: between begin and end every kernel launch of ch_encode is bracketed by
 * HIP events on the caller's stream (capacity max_launches events; launches beyond it are not recorded).
 * ch_model_profile_end waits for the last recorded event and returns, per category, the summed launch durations
 * (ms), the number of launches and their algorithmic FLOPs; `ncat` is the capacity of the caller's three arrays and must be
 * >= CH_NCAT (a caller built against a header with fewer categories is refused instead of overrun). */
/* The *_PRUNED categories are the launches of the final layer behind its attention, which run on the B * (1 + Q) rows the hashing
 * head reads instead of all B * N token rows: kept apart so that every category is ONE problem shape. */
enum {
    CH_CAT_IM2COL = 0, CH_CAT_GEMM_PATCH, CH_CAT_ROWOPS, CH_CAT_GEMM_QKV, CH_CAT_ATTENTION, CH_CAT_GEMM_OUT,
    CH_CAT_GEMM_DOWN, CH_CAT_GEMM_UP, CH_CAT_GEMM_FC1, CH_CAT_GEMM_FC2, CH_CAT_HEAD, CH_CAT_ADAPTER,
    CH_CAT_ATTENTION_PRUNED, CH_CAT_GEMM_OUT_PRUNED, CH_CAT_GEMM_DOWN_PRUNED, CH_CAT_GEMM_UP_PRUNED, CH_CAT_GEMM_FC1_PRUNED,
    CH_CAT_GEMM_FC2_PRUNED, CH_CAT_END, CH_NCAT
};
int ch_model_profile_begin(ch_model *m, int32_t max_launches);
int ch_model_profile_end(ch_model *m, int32_t ncat, double *ms_per_cat, int64_t *launches_per_cat, double *flops_per_cat);

/* Algorithmic FLOPs of one image through ch_encode (SURVEY.md section 8d formula). */
double ch_model_flops_per_image(const ch_model *m);

/* ---------------------------------------------------------------------------------------------------------------
 * Pre-process  (replaces the CPU-worker transform chain of configs/dataset/cub200.yaml:31-47 -- torchvision
 * Resize(size, bicubic) -> CenterCrop(crop) -> ToTensor -> normalize -- as run by engine.dataloader, engine.py:41-54)
 * ------------------------------------------------------------------------------------------------------------- */

/* One decoded image inside the concatenated uint8 HWC (RGB, 3 bytes per pixel) buffer.  The derived integers follow
 * Python / torchvision rounding and are computed by the host (concepthash_amd/preprocess.py):
 *   nw, nh      size after Resize(size): shorter side = size, the other int(size * long / short);
 *   left, top   CenterCrop origin int(round((n - crop) / 2.0)) (round-half-even);
 *   row0, nrows source rows the vertical pass needs for the crop's rows; tmp_offset: byte offset of this image's
 *               [nrows, crop, 3] intermediate in the workspace.  nrows = 0: the kernels leave this image's output slot
 *               untouched (the host wrapper uses it for images whose down-scaling needs more than ch_preprocess_max_taps()
 *               filter taps and fills the slot itself).
 * The training transforms of the same configs (configs/dataset/cub200.yaml:13-23: RandomResizedCrop(crop, bicubic) ->
 * RandomHorizontalFlip -> ToTensor -> normalize) use the same kernels: (h, w) is then the SIZE OF THE BOX the host drew
 * (torchvision get_params; PIL crops the box, then resizes it as an image of its own), src_offset points at the box's first
 * pixel, stride = the full image's width, nh = nw = crop, top = left = 0, and flip mirrors the output columns
 * (PIL FLIP_LEFT_RIGHT after the resize). */
typedef struct ch_image_desc {
    int64_t src_offset; /* bytes from `pixels` to the first pixel that is read (the image's, or the crop box's) */
    int64_t tmp_offset;
    int32_t h, w, nh, nw, top, left, row0, nrows;
    int32_t stride;     /* pixels per source row; 0 = w */
    int32_t flip;       /* 1: output column x is written to crop - 1 - x */
} ch_image_desc;

/* pixels: device uint8; desc_device: device array of B descriptors; max_rows = max nrows over the batch;
 * max_taps: an upper bound of the horizontal pass's filter taps per output column over the batch's images with nrows > 0 --
 * Pillow's ksize = 2 * ceil(2 * max(w / nw, 1)) + 1 (w / nw in double precision) -- or 0 = not known.  It only selects kernels:
 * up to 16 taps (down-scaling up to 3.5x) with crop % 4 == 0, 4-byte-aligned `pixels` and workspace and a 16-byte-aligned `out`
 * run the dword forms of the two passes (3x faster on MI355X), anything else the byte forms; the results are the same bits;
 * mean3_host / std3_host: HOST float[3]; out: device NCHW [B,3,crop,crop], out_dtype 0 = fp32, 1 = bf16;
 * workspace: device bytes, >= sum of nrows * crop * 3.  Resampling is Pillow's two-pass 8-bit bicubic (22-bit fixed-point
 * coefficients, uint8 intermediate): the uint8 pixels are bit-equal to PIL's, the output to the CPU chain's. */
int ch_preprocess(const uint8_t *pixels, const ch_image_desc *desc_device, int32_t B, int32_t max_rows, int32_t max_taps,
                  int32_t crop, const float *mean3_host, const float *std3_host, void *out, int32_t out_dtype,
                  uint8_t *workspace, void *stream);
/* largest number of filter taps per output pixel the kernels hold (down-scaling factor up to ~15.5) */
int32_t ch_preprocess_max_taps(void);

/* ---------------------------------------------------------------------------------------------------------------
 * JPEG decode split  (replaces the decode half of the loader workers: `Image.open(path).convert("RGB")` in the dataset classes
 * that engine.dataloader drives, engine.py:41-54 -- PIL = libjpeg-turbo with its default settings: islow IDCT, fancy upsampling)
 *
 * Host threads do what is serial (marker parsing + Huffman entropy decode -> int16 coefficient blocks in pinned memory), the GPU does
 * what is data-parallel (dequantise + 8x8 inverse DCT + chroma upsampling + YCbCr -> RGB), writing decoded RGB bytes in the layout
 * ch_preprocess reads.  Integer arithmetic restated from libjpeg-turbo (jidctint.c, jdsample.c, jdcolor.c): the bytes are bit-equal
 * to Pillow's.  Files outside the supported subset get status != 0 and are decoded by the caller's host decoder (PIL, the
 * reference's own path); see csrc/jpeg_host.cpp for the subset (plain C++: the host half; csrc/jpeg.hip holds the kernels).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct ch_jpeg_desc {
    int64_t coef_offset;  /* int16 elements from the batch coefficient buffer to this image's first block */
    int64_t pix_offset;   /* bytes from `pixels` to this image's [height, width, 3] RGB output */
    int64_t plane_offset; /* bytes from the plane workspace to this image's component planes */
    int32_t width, height;
    int32_t ncomp;        /* 1 (grey, replicated to RGB as PIL's convert("RGB") does) or 3 (YCbCr) */
    int32_t hs, vs;       /* luma sampling factors: 1x1 (4:4:4), 2x1 (4:2:2), 2x2 (4:2:0); chroma is 1x1 */
    int32_t mcu_w, mcu_h; /* MCUs per row / column */
    int32_t status;       /* 0 = decoded by this path (baseline, extended-sequential and progressive Huffman files); 1 not a JPEG,
                             2 truncated (inside the headers, or -- set by ch_jpeg_entropy_decode -- inside the entropy-coded data), 3 lossless / arithmetic coding, or a progressive file whose scans leave DC or the first AC
                             coefficients above bit 0 (libjpeg smooths those; set by ch_jpeg_entropy_decode), 4 not 8 bit, 5 component
                             count, 6 sampling factors, 7 multi-scan sequential, 8 colour space, 9 tables, 10 smaller than 16x16,
                             11 corrupt entropy data (set by ch_jpeg_entropy_decode), 12 more pixels than Pillow accepts (2 x MAX_IMAGE_PIXELS) */
    int32_t nblocks;      /* 8x8 blocks of all components = coefficient elements / 64 */
    int32_t reserved;
    uint16_t quant[3][64]; /* quantisation tables per component, natural (row-major) order */
} ch_jpeg_desc;

/* HOST: parse the headers of n files (files[i], lens[i]: host pointers / byte counts).  Fills desc[i] (sizes, sampling, tables,
 * status) and default back-to-back offsets; totals: int16 coefficient elements, output bytes, plane-workspace bytes.  The caller may
 * rewrite the three offsets (e.g. after sizing the slots of the files it decodes itself). */
int ch_jpeg_plan(const uint8_t *const *files, const int64_t *lens, int32_t n, ch_jpeg_desc *desc, int64_t *total_coef, int64_t *total_pix,
                 int64_t *total_plane);
/* HOST: Huffman-decode the entropy-coded segments of the files with status 0 on `nthreads` threads into coef_host (host memory,
 * ideally pinned; every block fully written, DC prediction undone, natural order; a progressive file's scans -- spectral selection and
 * successive approximation, ITU-T T.81 annex G -- are accumulated into the same blocks).  A corrupt stream sets that descriptor's status. */
int ch_jpeg_entropy_decode(const uint8_t *const *files, const int64_t *lens, int32_t n, ch_jpeg_desc *desc, int16_t *coef_host,
                           int32_t nthreads);
/* HOST: the same two calls for a batch whose files sit back to back in ONE buffer (what a `gpu_decode` loader worker hands over):
 * file i = data[offsets[i], offsets[i + 1]), offsets has n + 1 entries. */
int ch_jpeg_plan_packed(const uint8_t *data, const int64_t *offsets, int32_t n, ch_jpeg_desc *desc, int64_t *total_coef, int64_t *total_pix,
                        int64_t *total_plane);
int ch_jpeg_entropy_decode_packed(const uint8_t *data, const int64_t *offsets, int32_t n, ch_jpeg_desc *desc, int16_t *coef_host,
                                  int32_t nthreads);
/* HOST: the file reads of a batch, without Python in the loop and without worker processes (replaces what remains of the reference's
 * loader workers, engine.py:41-54, once decoding has left the CPU).  ch_io_file_sizes: stat every path; ch_io_read_files: file i ->
 * dst[offsets[i], offsets[i] + sizes[i]) on `nthreads` threads.  Status 3 + ch_last_error() names the file that failed. */
int ch_io_file_sizes(const char *const *paths, int32_t n, int64_t *sizes);
int ch_io_read_files(const char *const *paths, int32_t n, const int64_t *offsets, const int64_t *sizes, uint8_t *dst, int32_t nthreads);
/* GPU: coefficient blocks -> RGB bytes.  coef_dev: the coefficient buffer on the device; desc_dev / desc_host: the same n descriptors
 * on the device and on the host (the host copy sizes the launch); planes_ws: device workspace of total_plane bytes; pixels: device
 * output (total_pix bytes); images with status != 0 are skipped (their slots are the caller's to fill). */
int ch_jpeg_reconstruct(const int16_t *coef_dev, const ch_jpeg_desc *desc_dev, const ch_jpeg_desc *desc_host, int32_t n,
                        uint8_t *planes_ws, uint8_t *pixels, void *stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Retrieve  (replaces the un-vendored utils.hashing.{calculate_mAP, calculate_pr_curve, get_hamm_dist}; call sites
 *            experiments/test_hashing.py:106-119,153-162, trainers/orthohash.py:362; definition: SURVEY.md 8c)
 * ------------------------------------------------------------------------------------------------------------- */

/* codes [rows,nbit] fp32 -> packed [rows,W] uint64; bit i = (codes[i] - threshold) > 0.  Replaces the sign() /
 * threshold step of calculate_mAP (ternary_threshold = 0 at every reference call site). */
int ch_pack_sign(const float *codes, int64_t rows, int32_t nbit, float threshold, uint64_t *out_packed, void *stream);

/* Full distance matrix (small problems / get_hamm_dist semantics, trainers/orthohash.py:362; in-repo twin
 * get_hd, trainers/orthohash.py:263-264): out[Qn,G] int32 = popcount(q xor g). */
int ch_hamming_dist(const uint64_t *q, int64_t Qn, const uint64_t *g, int64_t G, int32_t W, int32_t *out,
                    void *stream);

/* Top-k by ascending (distance, gallery index).  g_index_base is added to the local gallery row number (gallery
 * shards).  out_idx [Qn,k] int64 (global index, -1 when k > G), out_dist [Qn,k] int32 (-1 when k > G).
 * 1 <= k <= 128, 1 <= W <= 4.  workspace: ch_hamming_topk_workspace() bytes. */
size_t ch_hamming_topk_workspace(int64_t Qn, int64_t G, int32_t W, int32_t k);
int ch_hamming_topk(const uint64_t *q, int64_t Qn, const uint64_t *g, int64_t G, int32_t W, int32_t k,
                    int64_t g_index_base, int64_t *out_idx, int32_t *out_dist, void *workspace,
                    size_t workspace_bytes, void *stream);

/* Merge nlists per-shard top-k lists ([nlists,Qn,k] each sorted ascending by (dist, idx); -1 entries = absent)
 * into the global top-k [Qn,k]. */
int ch_topk_merge(const int64_t *idx_lists, const int32_t *dist_lists, int32_t nlists, int64_t Qn, int32_t k,
                  int64_t *out_idx, int32_t *out_dist, void *stream);

/* Labels: single-label mode (LW == 0): int32 class id per row.  Multi-label mode (LW > 0): uint64[rows,LW] bitmasks;
 * relevant <=> masks intersect. */

/* mAP pass 1: per (gallery segment, query, distance bucket) counts.  The gallery is cut into nseg contiguous segments
 * of seg_rows rows (last one shorter); out_hist [nseg, Qn, nb, 2] uint32 with nb = 64*W+1, [..,0] = #rows at that
 * distance, [..,1] = #relevant rows.  seg_rows <= 65535. */
int ch_hamming_hist(const uint64_t *q, int64_t Qn, const uint64_t *g, int64_t G, int32_t W, const void *q_labels,
                    const void *g_labels, int32_t LW, int32_t seg_rows, uint32_t *out_hist, void *stream);

/* mAP pass 2: AP numerators in 2^-32 fixed point.  base [nseg,Qn,nb,2] uint32: for each (segment, query, bucket) the
 * number of rows / relevant rows ranked before the segment's first row of that bucket (global prefix over lower
 * buckets and over earlier segments -- and earlier gallery shards -- of the same bucket; built by
 * ch_hamming_hist_prefix or by the multi-GPU host code).  rank_limit = R (rows ranked after R do not count; <= 0: no
 * limit).  first_rel: NULL, or int32[Qn] = relevance (0/1) of each query's rank-1 row, which is then dropped before
 * ranking (remove_first_retrieved, experiments/test_hashing.py:105-112).  Accumulates (atomic add) into out_S[Qn]
 * uint64 and out_nrel[Qn] uint32 -- zero them first.  AP[q] = S[q] / (nrel[q] * 2^32). */
int ch_hamming_ap(const uint64_t *q, int64_t Qn, const uint64_t *g, int64_t G, int32_t W, const void *q_labels,
                  const void *g_labels, int32_t LW, int32_t seg_rows, const uint32_t *base, int64_t rank_limit,
                  const int32_t *first_rel, unsigned long long *out_S, uint32_t *out_nrel, void *stream);

/* The same pass for up to 16 rank limits at once (one gallery scan instead of one per limit): rank_limits[nlimits] is a
 * HOST array, ascending, entries <= 0 meaning "no limit" (they sort last); out_S [nlimits, Qn] uint64 and out_nrel
 * [nlimits, Qn] uint32 are accumulated row r under limit r.  With limits = the list R of calculate_mAP, the depths of
 * calculate_pr_curve (experiments/test_hashing.py:124-128,153-167) or the k of P@k / R@k -- out_nrel under limit k IS
 * the number of relevant rows in the top k -- every depth-dependent statistic of the evaluator comes out of one pass. */
int ch_hamming_ap_multi(const uint64_t *q, int64_t Qn, const uint64_t *g, int64_t G, int32_t W, const void *q_labels,
                        const void *g_labels, int32_t LW, int32_t seg_rows, const uint32_t *base,
                        const int64_t *rank_limits, int32_t nlimits, const int32_t *first_rel, unsigned long long *out_S,
                        uint32_t *out_nrel, void *stream);

/* Record form of the two passes: ONE distance scan.  ch_hamming_hist_rec is ch_hamming_hist that additionally leaves, for every
 * relevant (query, gallery row) pair, a record (distance, position inside its (segment, query, distance) bucket) in a per-lane list:
 * rec = uint2[ch_hamming_rec_workgroups() * rec_cap * ch_hamming_rec_block(W)], rec_cnt = uint32[nseg, Qn] (list lengths),
 * wg_flags = uint32[ch_hamming_rec_workgroups()], ZEROED by the caller, set to 1 for a (query tile, segment) workgroup in which
 * some list needed more than rec_cap entries.  ch_hamming_ap_rec then produces exactly what ch_hamming_ap_multi produces (same
 * arguments + the record buffers; out_S / out_nrel zeroed by the caller): it walks the lists -- G / C records per query instead of
 * G rows -- and redoes the flagged workgroups by the two-scan kernel, so the result does not depend on rec_cap.  Replaces the second
 * scan of calculate_mAP's ranking (experiments/test_hashing.py:114-119) where the label distribution lets the lists stay short. */
size_t ch_hamming_rec_workgroups(int64_t Qn, int64_t G, int32_t W, int32_t seg_rows);
int32_t ch_hamming_rec_block(int32_t W);
int ch_hamming_hist_rec(const uint64_t *q, int64_t Qn, const uint64_t *g, int64_t G, int32_t W, const void *q_labels,
                        const void *g_labels, int32_t LW, int32_t seg_rows, uint32_t *out_hist, void *rec, int32_t rec_cap,
                        uint32_t *rec_cnt, uint32_t *wg_flags, void *stream);
int ch_hamming_ap_rec(const uint64_t *q, int64_t Qn, const uint64_t *g, int64_t G, int32_t W, const void *q_labels,
                      const void *g_labels, int32_t LW, int32_t seg_rows, const uint32_t *base, const void *rec, int32_t rec_cap,
                      const uint32_t *rec_cnt, const uint32_t *wg_flags, const int64_t *rank_limits, int32_t nlimits,
                      const int32_t *first_rel, unsigned long long *out_S, uint32_t *out_nrel, void *stream);

/* Single-GPU helper: hist [nseg,Qn,nb,2] -> base (same shape) + totals[Qn,2] (rows, relevant rows overall). */
int ch_hamming_hist_prefix(const uint32_t *hist, int32_t nseg, int64_t Qn, int32_t nb, uint32_t *out_base,
                           uint32_t *out_totals, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CONCEPTHASH_HIP_H */
