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
    float *cls = red + 8;             // [D]  post-LN CLS token (optional branch)

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
        // x <- l2(x + concept_pe) per concept
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
            for (int i = tid; i < p.D; i += 256) x[c * p.D + i] *= inv;
            __syncthreads();
        }
        for (int o = wid; o < p.Q * p.C; o += 4) {
            const int c = o / p.C, k = o - c * p.C;
            const float v = wave_dot(x + c * p.D, p.concept_cent_l2 + (size_t)k * p.D, p.D, lane);
            if (lane == 0) p.out_logits_concept[((size_t)c * p.B + b) * p.C + k] = v;
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
        for (int i = tid; i < p.D; i += 256) cls[i] = (h0[i] - mean) * rstd * p.post_w[i] + p.post_b[i];
        __syncthreads();
        for (int o = wid; o < p.P; o += 4) {
            const float v = wave_dot(cls, p.vis_proj + (size_t)o * p.D, p.D, lane);
            if (lane == 0) p.out_image_features[(size_t)b * p.P + o] = v;
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
    const size_t lds = sizeof(float) * ((size_t)p.Q * p.D + p.nbit + 8 + p.D);
    CH_REQUIRE(lds <= 64 * 1024, "head: Q*D too large");
    hipLaunchKernelGGL(head_kernel, dim3(p.B), dim3(256), lds, s, p);
    CH_LAUNCH_CHECK();
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
