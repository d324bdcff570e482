// fp32 helper kernels used ONCE at model load to fold the input-independent parts of the reference forward on the GPU:
//   forward_hash_query (models/arch/coop.py:413-427)  -> concept tokens (LN + MHA + FFN + Linear on `hash_queries`)
//   get_center          (models/arch/coop.py:624-625)  -> text_projection(center), then l2 / sign variants (:573-580)
//   CosSim centroids    (models/layers/cossim.py:74-76) -> l2-normalised rows
//   BatchNorm1d eval    (models/arch/coop.py:559)      -> per-bit scale/shift
//   weight conversion fp32 -> bf16 (GEMM operands), optional K padding (patch-embed for patch 14)
// Plain one-wave-per-output dot products; none of this is on the per-image hot path.
#include "ch_common.h"
#include "kernels.h"

namespace {

__global__ void small_linear_kernel(const float *x, int rows, int in_f, const float *W, const float *b, int out_f, int act,
                                    float *y) {
    const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (o >= (int64_t)rows * out_f) return;
    const int r = (int)(o / out_f), c = (int)(o - (int64_t)r * out_f);
    float s = 0.f;
    for (int i = lane; i < in_f; i += 64) s += x[(size_t)r * in_f + i] * W[(size_t)c * in_f + i];
    s = wave_sum(s);
    if (lane == 0) {
        if (b) s += b[c];
        if (act == 1) s = fmaxf(s, 0.f);
        y[o] = s;
    }
}

__global__ void small_layernorm_kernel(const float *x, int rows, int D, const float *w, const float *b, float eps, float *y) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float *xr = x + (size_t)r * D;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) s += xr[i];
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
    for (int i = lane; i < D; i += 64) {
        const float d = xr[i] - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    for (int i = lane; i < D; i += 64) y[(size_t)r * D + i] = (xr[i] - mean) * rstd * w[i] + b[i];
}

__global__ void small_add_kernel(const float *a, const float *b, int64_t n, float *y) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a[i] + b[i];
}

// one wave per (head, query token); T <= 64 keys
__global__ void small_mha_kernel(const float *qkv, int T, int P, int heads, float *out) {
    const int hd = P / heads;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (o >= heads * T) return;
    const int h = o / T, t = o - h * T;
    const float *q = qkv + (size_t)t * 3 * P + h * hd;
    float sc = -1e30f;
    if (lane < T) {
        const float *k = qkv + (size_t)lane * 3 * P + P + h * hd;
        float s = 0.f;
        for (int d = 0; d < hd; ++d) s += q[d] * k[d];
        sc = s * rsqrtf((float)hd);
    }
    const float mx = wave_max(sc);
    const float e = lane < T ? __expf(sc - mx) : 0.f;
    const float prob = e / wave_sum(e);
    for (int d = 0; d < hd; ++d) {
        float v = lane < T ? prob * qkv[(size_t)lane * 3 * P + 2 * P + h * hd + d] : 0.f;
        v = wave_sum(v);
        if (lane == 0) out[(size_t)t * P + h * hd + d] = v;
    }
}

__global__ void small_l2norm_kernel(const float *x, int rows, int cols, float *out_l2, float *out_bin) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    float s = 0.f;
    for (int i = lane; i < cols; i += 64) s += x[(size_t)r * cols + i] * x[(size_t)r * cols + i];
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    const float bs = rsqrtf((float)cols);
    for (int i = lane; i < cols; i += 64) {
        const float v = x[(size_t)r * cols + i] * inv;
        out_l2[(size_t)r * cols + i] = v;
        if (out_bin) out_bin[(size_t)r * cols + i] = v > 0.f ? bs : (v < 0.f ? -bs : 0.f);
    }
}

__global__ void convert_bf16_kernel(const float *x, int64_t rows, int cols, int cols_pad, bf16_t *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols_pad) return;
    const int64_t r = i / cols_pad;
    const int c = (int)(i - r * cols_pad);
    out[i] = c < cols ? f2bf(x[r * cols + c]) : (bf16_t)0;
}

__global__ void bn_fold_kernel(const float *w, const float *b, const float *mean, const float *var, int n, float eps,
                               float *scale, float *shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float sc = w[i] / sqrtf(var[i] + eps);
    scale[i] = sc;
    shift[i] = b[i] - mean[i] * sc;
}

// LayerNorm fold of a Linear, done once at model build:
// Wd' = bf16(Wd * gamma) (rows >= b zero), c[n] = sum_k float(Wd'[n][k]), d[n] = sum_k beta[k] * Wd[n][k] + bd[n]
__global__ void fold_ln_kernel(const float *Wd, const float *bd, const float *gamma, const float *beta, int b, int bpad, int D,
                               bf16_t *Wdf, float *c, float *d) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= bpad) return;
    float cs = 0.f, ds = 0.f;
    for (int k = lane; k < D; k += 64) {
        const float w = n < b ? Wd[(size_t)n * D + k] : 0.f;
        const bf16_t wb = f2bf(w * gamma[k]);
        Wdf[(size_t)n * D + k] = wb;
        cs += bf2f(wb);
        ds += beta[k] * w;
    }
    cs = wave_sum(cs);
    ds = wave_sum(ds);
    if (lane == 0) {
        c[n] = cs;
        d[n] = ds + (n < b ? bd[n] : 0.f);
    }
}

}  // namespace

int ch_small_linear(const float *x, int rows, int in_f, const float *W, const float *b, int out_f, int act, float *y,
                    hipStream_t s) {
    const int64_t outs = (int64_t)rows * out_f;
    hipLaunchKernelGGL(small_linear_kernel, dim3((unsigned)ceil_div64(outs, 4)), dim3(256), 0, s, x, rows, in_f, W, b,
                       out_f, act, y);
    CH_LAUNCH_CHECK();
    return 0;
}
int ch_small_layernorm(const float *x, int rows, int D, const float *w, const float *b, float eps, float *y, hipStream_t s) {
    hipLaunchKernelGGL(small_layernorm_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, s, x, rows, D, w, b, eps, y);
    CH_LAUNCH_CHECK();
    return 0;
}
int ch_small_add(const float *a, const float *b, int64_t n, float *y, hipStream_t s) {
    hipLaunchKernelGGL(small_add_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, s, a, b, n, y);
    CH_LAUNCH_CHECK();
    return 0;
}
int ch_small_mha(const float *qkv, int T, int P, int heads, float *out, hipStream_t s) {
    CH_REQUIRE(T <= 64, "concept-token generator: more than 64 concept tokens is not supported");
    CH_REQUIRE(P % heads == 0, "concept-token generator: width not divisible by heads");
    hipLaunchKernelGGL(small_mha_kernel, dim3((unsigned)ceil_div64((int64_t)heads * T, 4)), dim3(256), 0, s, qkv, T, P, heads,
                       out);
    CH_LAUNCH_CHECK();
    return 0;
}
int ch_small_l2norm(const float *x, int rows, int cols, float *out_l2, float *out_bin, hipStream_t s) {
    hipLaunchKernelGGL(small_l2norm_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, s, x, rows, cols, out_l2,
                       out_bin);
    CH_LAUNCH_CHECK();
    return 0;
}
int ch_convert_bf16(const float *x, int64_t rows, int cols, int cols_pad, bf16_t *out, hipStream_t s) {
    hipLaunchKernelGGL(convert_bf16_kernel, dim3((unsigned)ceil_div64(rows * cols_pad, 256)), dim3(256), 0, s, x, rows, cols,
                       cols_pad, out);
    CH_LAUNCH_CHECK();
    return 0;
}
int ch_bn_fold(const float *w, const float *b, const float *mean, const float *var, int n, float eps, float *scale,
               float *shift, hipStream_t s) {
    hipLaunchKernelGGL(bn_fold_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, s, w, b, mean, var, n, eps, scale,
                       shift);
    CH_LAUNCH_CHECK();
    return 0;
}
int ch_fold_ln(const float *Wd, const float *bd, const float *gamma, const float *beta, int b, int bpad, int D, bf16_t *Wdf,
               float *c, float *d, hipStream_t s) {
    hipLaunchKernelGGL(fold_ln_kernel, dim3((unsigned)ceil_div64(bpad, 4)), dim3(256), 0, s, Wd, bd, gamma, beta, b, bpad, D, Wdf,
                       c, d);
    CH_LAUNCH_CHECK();
    return 0;
}
