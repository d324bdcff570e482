// ConceptHash hashing head (fp32): everything after the last encoder layer.
//
// Reference restated (models/arch/coop.py):
//   :503-509  hash_features = last_hidden_state[:, -Q:, :]              (raw residual stream, no post-LN)
//   :544-559  codes = BN_eval(concat_c((hash_features_c + hash_pe_c) @ hash_fc.W^T))   -- BN folded to scale/shift
//   :573-580  logits_cont = l2(codes) . l2(center')^T ; logits_bin = l2(codes) . (sign(l2(center')) / sqrt(nbit))^T
//   :269-276  logits_concept[c] = l2(hash_features_c + concept_pe_c) . l2(centroids)^T   (CosSim, models/layers/cossim.py:37-82)
//   :498-501  image_features = visual_projection(post_layernorm(h[:, 0]))
// plus sign-packing of the codes to uint64 (bit i = codes[i] > 0, little endian), the input of the Hamming kernels.
// One workgroup per image; a wave per dot product; < 0.001 % of the encoder's FLOPs.
// The two WIDE optional outputs -- logits_concept (Q x C dot products of length D per image) and image_features (P of them) -- are
// small dense products over the whole batch: head_kernel only leaves their left operands (the l2-normalised concept rows, the
// post-LN CLS rows) in a workspace and head_dense_kernel multiplies them by the centroid / projection matrices with the fp32 MFMA
// (v_mfma_f32_16x16x4_f32: an exact fp32 fmaf chain), one 16 x 16 output tile per wave.  As per-image wave dot products these two
// outputs re-read 3.9 MB of weights per image from L2 with one workgroup per CU: 1.3 ms per 256 images, 10 % of an encode step,
// in the evaluator loop that always asks for them (trainers/coop.py:59-71); as tiles 16 images share every weight row: ~30 us.
#include "ch_common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float wave_dot(const float *a, const float *b, int n, int lane) {
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += a[i] * b[i];
    return wave_sum(s);
}

__global__ __launch_bounds__(256) void head_kernel(HeadParams p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *x = sm;                    // [Q*D]  hash_features (+ pe variants are formed on the fly)
    float *codes = x + p.Q * p.D;     // [nbit]
    float *red = codes + p.nbit;      // [8] scratch

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int sub = p.nbit / p.Q;
    const float *hrow = p.H + ((size_t)b * p.ntok + (p.ntok - p.Q)) * p.D;
    for (int i = tid; i < p.Q * p.D; i += 256) {
        const float v = hrow[i];
        x[i] = v;
        if (p.out_hash_features) p.out_hash_features[(size_t)b * p.Q * p.D + i] = v;
    }
    __syncthreads();

    // codes
    for (int o = wid; o < p.nbit; o += 4) {
        const int c = o / sub, s = o - c * sub;
        const float *xc = x + c * p.D, *pe = p.hash_pe + c * p.D, *w = p.hash_fc + (size_t)s * p.D;
        float acc = 0.f;
        for (int i = lane; i < p.D; i += 64) acc += (xc[i] + pe[i]) * w[i];
        acc = wave_sum(acc);
        if (lane == 0) {
            const float v = acc * p.bn_scale[o] + p.bn_shift[o];
            codes[o] = v;
            p.out_codes[(size_t)b * p.nbit + o] = v;
        }
    }
    __syncthreads();

    if (p.out_packed) {
        const int W = (p.nbit + 63) / 64;
        if (tid < W) {
            uint64_t word = 0;
            for (int i = 0; i < 64 && tid * 64 + i < p.nbit; ++i)
                if (codes[tid * 64 + i] > 0.0f) word |= (uint64_t)1 << i;
            p.out_packed[(size_t)b * W + tid] = word;
        }
    }

    if (p.out_logits_cont || p.out_logits_bin) {
        if (wid == 0) {
            float s = 0.f;
            for (int i = lane; i < p.nbit; i += 64) s += codes[i] * codes[i];
            s = wave_sum(s);
            if (lane == 0) red[0] = 1.0f / fmaxf(sqrtf(s), 1e-12f);  // F.normalize eps
        }
        __syncthreads();
        const float inv = red[0];
        for (int c = tid; c < p.C; c += 256) {
            const float *cl = p.center_l2 + (size_t)c * p.nbit, *cb = p.center_bin + (size_t)c * p.nbit;
            float a = 0.f, bb = 0.f;
            for (int i = 0; i < p.nbit; ++i) {
                a += codes[i] * cl[i];
                bb += codes[i] * cb[i];
            }
            if (p.out_logits_cont) p.out_logits_cont[(size_t)b * p.C + c] = a * inv;
            if (p.out_logits_bin) p.out_logits_bin[(size_t)b * p.C + c] = bb * inv;
        }
    }

    if (p.out_logits_concept) {
        __syncthreads();
        // x <- l2(x + concept_pe) per concept, left in the workspace for head_dense_kernel
        for (int c = 0; c < p.Q; ++c) {
            float s = 0.f;
            for (int i = tid; i < p.D; i += 256) {
                const float v = x[c * p.D + i] + p.concept_pe[c * p.D + i];
                x[c * p.D + i] = v;
                s += v * v;
            }
            s = wave_sum(s);
            if (lane == 0) red[4 + wid] = s;
            __syncthreads();
            const float inv = 1.0f / fmaxf(sqrtf(red[4] + red[5] + red[6] + red[7]), 1e-12f);
            for (int i = tid; i < p.D; i += 256) p.ws_xn[((size_t)b * p.Q + c) * p.D + i] = x[c * p.D + i] * inv;
            __syncthreads();
        }
    }

    if (p.out_image_features) {
        __syncthreads();
        const float *h0 = p.H + (size_t)b * p.ntok * p.D;
        float s = 0.f;
        for (int i = tid; i < p.D; i += 256) s += h0[i];
        s = wave_sum(s);
        if (lane == 0) red[4 + wid] = s;
        __syncthreads();
        const float mean = (red[4] + red[5] + red[6] + red[7]) / (float)p.D;
        __syncthreads();
        float q = 0.f;
        for (int i = tid; i < p.D; i += 256) {
            const float d = h0[i] - mean;
            q += d * d;
        }
        q = wave_sum(q);
        if (lane == 0) red[4 + wid] = q;
        __syncthreads();
        const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)p.D + p.ln_eps);
        for (int i = tid; i < p.D; i += 256) p.ws_cls[(size_t)b * p.D + i] = (h0[i] - mean) * rstd * p.post_w[i] + p.post_b[i];
    }
}

// out[row][col] = sum_k X[row][k] * W[col][k]  (fp32, v_mfma_f32_16x16x4_f32), one 16 x 16 tile per wave.  Lane (i = lane & 15,
// q = lane >> 4) loads X[row i][k0 + 4q .. +3] and W[col i][k0 + 4q .. +3] as float4 and feeds component s to MFMA step s: over the
// four steps every k of the 16-wide slab is used exactly once (the A / B operand maps are lane -> [i][k = q]).
// q_rows > 0: rows are (image b, concept c) pairs, row = b * q_rows + c, and the output is laid out [c][b][col] (logits_concept).
__global__ __launch_bounds__(256) void head_dense_kernel(const float *__restrict__ X, int rows, int D, const float *__restrict__ W, int N,
                                                         float *__restrict__ out, int q_rows, int nimg) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int tiles_n = (N + 15) / 16, tiles = ((rows + 15) / 16) * tiles_n;
    if (wave >= tiles) return;   // wave-uniform
    const int tr = wave / tiles_n, tc = wave - tr * tiles_n;
    const int i = lane & 15, q = lane >> 4;
    const float *xp = X + (size_t)min(tr * 16 + i, rows - 1) * D + 4 * q;
    const float *wp = W + (size_t)min(tc * 16 + i, N - 1) * D + 4 * q;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k0 = 0; k0 < D; k0 += 16) {
        const f32x4 a = *(const f32x4 *)(xp + k0), w = *(const f32x4 *)(wp + k0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], w[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], w[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], w[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], w[3], acc, 0, 0, 0);
    }
    const int col = tc * 16 + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = tr * 16 + q * 4 + r;     // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
        if (row < rows && col < N) {
            if (q_rows > 0) {
                const int b = row / q_rows, c = row - b * q_rows;
                out[((size_t)c * nimg + b) * N + col] = acc[r];
            } else {
                out[(size_t)row * N + col] = acc[r];
            }
        }
    }
}

__global__ void pack_sign_kernel(const float *codes, int64_t rows, int nbit, float thr, uint64_t *out) {
    const int W = (nbit + 63) / 64;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one wave-lane per (row, word)? -> one thread per bit
    // thread = one bit; a 64-lane wave = one output word (ballot)
    const int64_t word = gid >> 6;
    const int bit = (int)(gid & 63);
    if (word >= rows * W) return;
    const int64_t r = word / W;
    const int w = (int)(word - r * W);
    const int i = w * 64 + bit;
    const bool on = i < nbit && (codes[r * nbit + i] - thr) > 0.0f;
    const unsigned long long m = __ballot(on);
    if (bit == 0) out[word] = m;
}

}  // namespace

int ch_head(const HeadParams &p, hipStream_t s) {
    CH_REQUIRE(p.nbit % p.Q == 0, "head: nbit must be divisible by the number of concept tokens");
    const size_t lds = sizeof(float) * ((size_t)p.Q * p.D + p.nbit + 8);
    CH_REQUIRE(lds <= 64 * 1024, "head: Q*D too large");
    CH_REQUIRE(!(p.out_logits_concept && !p.ws_xn) && !(p.out_image_features && !p.ws_cls), "head: workspace for the dense outputs is missing");
    CH_REQUIRE(p.D % 16 == 0, "head: D must be a multiple of 16");
    hipLaunchKernelGGL(head_kernel, dim3(p.B), dim3(256), lds, s, p);
    CH_LAUNCH_CHECK();
    if (p.out_logits_concept) {
        const int rows = p.B * p.Q, tiles = ((rows + 15) / 16) * ((p.C + 15) / 16);
        hipLaunchKernelGGL(head_dense_kernel, dim3((tiles + 3) / 4), dim3(256), 0, s, p.ws_xn, rows, p.D, p.concept_cent_l2, p.C,
                           p.out_logits_concept, p.Q, p.B);
        CH_LAUNCH_CHECK();
    }
    if (p.out_image_features) {
        const int tiles = ((p.B + 15) / 16) * ((p.P + 15) / 16);
        hipLaunchKernelGGL(head_dense_kernel, dim3((tiles + 3) / 4), dim3(256), 0, s, p.ws_cls, p.B, p.D, p.vis_proj, p.P,
                           p.out_image_features, 0, p.B);
        CH_LAUNCH_CHECK();
    }
    return 0;
}

int ch_pack_sign_launch(const float *codes, int64_t rows, int nbit, float thr, uint64_t *out, hipStream_t s) {
    if (rows == 0) return 0;
    const int W = (nbit + 63) / 64;
    const int64_t threads = rows * W * 64;
    hipLaunchKernelGGL(pack_sign_kernel, dim3((unsigned)ceil_div64(threads, 256)), dim3(256), 0, s, codes, rows, nbit, thr,
                       out);
    CH_LAUNCH_CHECK();
    return 0;
}
