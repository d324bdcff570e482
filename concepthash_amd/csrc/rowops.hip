// Memory-bound row kernels of the ConceptHash encoder: im2col for the patch embedding, token assembly + pre-LN,
// LayerNorm (fp32 or bf16 rows -> bf16 GEMM operand).  One wave (64 lanes) per row, values kept in registers, two-pass
// mean/variance like torch.nn.functional.layer_norm; 8-byte vector accesses, 4 rows per 256-thread block.
//
// Reference arithmetic restated: nn.LayerNorm in HF CLIPEncoderLayer (layer_norm1/2), pre_layrnorm
// (models/arch/coop.py:472), Adapter.adapter_layer_norm (models/layers/adapter.py:47-48); token assembly
// models/arch/coop.py:452-471 (class token + position embedding; concept tokens appended after the pos-embed add).
#include "ch_common.h"
#include "kernels.h"

namespace {

constexpr int MAXP = 10;  // D <= 1280, D % 128 == 0

struct Row {
    float2 v[MAXP];
};

__device__ __forceinline__ void row_stats(const Row &r, int npass, int D, float &mean, float &rstd, float eps) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAXP; ++j)
        if (j < npass) s += r.v[j].x + r.v[j].y;
    mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MAXP; ++j)
        if (j < npass) {
            const float a = r.v[j].x - mean, b = r.v[j].y - mean;
            q += a * a + b * b;
        }
    rstd = rsqrtf(wave_sum(q) / (float)D + eps);
}

__device__ __forceinline__ void row_affine(Row &r, int npass, int lane, float mean, float rstd, const float *w,
                                           const float *b) {
#pragma unroll
    for (int j = 0; j < MAXP; ++j)
        if (j < npass) {
            const float2 ww = *(const float2 *)(w + (j * 64 + lane) * 2);
            const float2 bb = *(const float2 *)(b + (j * 64 + lane) * 2);
            r.v[j].x = (r.v[j].x - mean) * rstd * ww.x + bb.x;
            r.v[j].y = (r.v[j].y - mean) * rstd * ww.y + bb.y;
        }
}

__device__ __forceinline__ void row_load_f32(Row &r, int npass, int lane, const float *x) {
#pragma unroll
    for (int j = 0; j < MAXP; ++j)
        if (j < npass) r.v[j] = *(const float2 *)(x + (j * 64 + lane) * 2);
}
__device__ __forceinline__ void row_load_bf16(Row &r, int npass, int lane, const bf16_t *x) {
#pragma unroll
    for (int j = 0; j < MAXP; ++j)
        if (j < npass) {
            const uint32_t u = *(const uint32_t *)(x + (j * 64 + lane) * 2);
            r.v[j].x = bf2f((bf16_t)(u & 0xffff));
            r.v[j].y = bf2f((bf16_t)(u >> 16));
        }
}
__device__ __forceinline__ void row_store_f32(const Row &r, int npass, int lane, float *x) {
#pragma unroll
    for (int j = 0; j < MAXP; ++j)
        if (j < npass) *(float2 *)(x + (j * 64 + lane) * 2) = r.v[j];
}
__device__ __forceinline__ void row_store_bf16(const Row &r, int npass, int lane, bf16_t *x) {
#pragma unroll
    for (int j = 0; j < MAXP; ++j)
        if (j < npass) *(uint32_t *)(x + (j * 64 + lane) * 2) = pack_bf16x2(r.v[j].x, r.v[j].y);
}

template <bool IN_BF16>
__global__ __launch_bounds__(256) void layernorm_kernel(const void *x, int64_t rows, int D, const float *w,
                                                        const float *b, float eps, bf16_t *out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int npass = D >> 7;
    Row r;
    if constexpr (IN_BF16)
        row_load_bf16(r, npass, lane, (const bf16_t *)x + row * D);
    else
        row_load_f32(r, npass, lane, (const float *)x + row * D);
    float mean, rstd;
    row_stats(r, npass, D, mean, rstd, eps);
    row_affine(r, npass, lane, mean, rstd, w, b);
    row_store_bf16(r, npass, lane, out + row * D);
}

// bf16 rows -> bf16 rows with 16-byte accesses: a lane owns 8 consecutive features per chunk, chunks lane and lane + 64
// (D <= 1024).  Twice the bytes per memory instruction of the generic kernel above.
__global__ __launch_bounds__(256) void layernorm_bf16_wide_kernel(const bf16_t *__restrict__ x, int64_t rows, int D,
                                                                  const float *__restrict__ w, const float *__restrict__ b,
                                                                  float eps, bf16_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int chunks = D >> 3;
    float v[2][8];
    bool on[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = lane + 64 * j;
        on[j] = c < chunks;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (on[j]) u = *(const uint4 *)(x + row * D + c * 8);
        const uint32_t ww[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[j][2 * e] = bf2f((bf16_t)(ww[e] & 0xffff));
            v[j][2 * e + 1] = bf2f((bf16_t)(ww[e] >> 16));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[j][e];  // inactive chunks hold zeros
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
        if (on[j]) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[j][e] - mean;
                q += d * d;
            }
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int j = 0; j < 2; ++j)
        if (on[j]) {
            const int c = lane + 64 * j;
            const f32x4 w0 = *(const f32x4 *)(w + c * 8), w1 = *(const f32x4 *)(w + c * 8 + 4);
            const f32x4 b0 = *(const f32x4 *)(b + c * 8), b1 = *(const f32x4 *)(b + c * 8 + 4);
            float y[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = (v[j][e] - mean) * rstd * w0[e] + b0[e];
                y[4 + e] = (v[j][4 + e] - mean) * rstd * w1[e] + b1[e];
            }
            uint4 o;
            o.x = pack_bf16x2(y[0], y[1]);
            o.y = pack_bf16x2(y[2], y[3]);
            o.z = pack_bf16x2(y[4], y[5]);
            o.w = pack_bf16x2(y[6], y[7]);
            *(uint4 *)(out + row * D + c * 8) = o;
        }
}

__global__ __launch_bounds__(256) void assemble_preln_kernel(float *H, int64_t rows, int ntok, int np, int D,
                                                             const float *cls_pos0, const float *ctx,
                                                             const float *pre_w, const float *pre_b,
                                                             const float *ln_w, const float *ln_b, float eps,
                                                             bf16_t *xn) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int t = (int)(row % ntok);
    const int npass = D >> 7;
    const float *src = t == 0 ? cls_pos0 : (t > np ? ctx + (size_t)(t - np - 1) * D : H + row * D);
    Row r;
    row_load_f32(r, npass, lane, src);
    float mean, rstd;
    row_stats(r, npass, D, mean, rstd, eps);
    row_affine(r, npass, lane, mean, rstd, pre_w, pre_b);
    row_store_f32(r, npass, lane, H + row * D);
    row_stats(r, npass, D, mean, rstd, eps);
    row_affine(r, npass, lane, mean, rstd, ln_w, ln_b);
    row_store_bf16(r, npass, lane, xn + row * D);
}

// one thread = 8 consecutive k (same channel, same ky when patch % 8 == 0)
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const T *img, int B, int image, int patch, int grid, int K,
                                                     int Kp, bf16_t *out, bool fast) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int chunks = Kp >> 3;
    const int64_t total = (int64_t)B * grid * grid * chunks;
    if (gid >= total) return;
    const int ck = (int)(gid % chunks);
    const int64_t prow = gid / chunks;  // b*Np + p
    const int p = (int)(prow % (grid * grid));
    const int b = (int)(prow / (grid * grid));
    const int py = p / grid, px = p - py * grid;
    const int pp = patch * patch;
    if (fast && ck * 8 < K) {
        // patch % 8 == 0 and image % 8 == 0: the 8 values are 8 consecutive pixels of one image row (same channel, same ky), 16-byte
        // aligned -> one vector load (bf16 input: a plain 16-byte copy) instead of 8 scalar loads with their index arithmetic
        const int k = ck * 8;
        const int c = k / pp, rem = k - c * pp;
        const int ky = rem / patch, kx = rem - ky * patch;
        const size_t off = (((size_t)b * 3 + c) * image + (size_t)(py * patch + ky)) * image + (px * patch + kx);
        uint4 o;
        if constexpr (sizeof(T) == 4) {
            const f32x4 lo = *(const f32x4 *)((const float *)img + off), hi = *(const f32x4 *)((const float *)img + off + 4);
            o.x = pack_bf16x2(lo[0], lo[1]);
            o.y = pack_bf16x2(lo[2], lo[3]);
            o.z = pack_bf16x2(hi[0], hi[1]);
            o.w = pack_bf16x2(hi[2], hi[3]);
        } else {
            o = *(const uint4 *)((const bf16_t *)img + off);
        }
        *(uint4 *)(out + prow * Kp + ck * 8) = o;
        return;
    }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = ck * 8 + e;
        float val = 0.f;
        if (k < K) {
            const int c = k / pp, rem = k - c * pp;
            const int ky = rem / patch, kx = rem - ky * patch;
            const size_t off = (((size_t)b * 3 + c) * image + (size_t)(py * patch + ky)) * image + (px * patch + kx);
            if constexpr (sizeof(T) == 4)
                val = ((const float *)img)[off];
            else
                val = bf2f(((const bf16_t *)img)[off]);
        }
        v[e] = val;
    }
    uint4 o;
    o.x = pack_bf16x2(v[0], v[1]);
    o.y = pack_bf16x2(v[2], v[3]);
    o.z = pack_bf16x2(v[4], v[5]);
    o.w = pack_bf16x2(v[6], v[7]);
    *(uint4 *)(out + prow * Kp + ck * 8) = o;
}

// out[b * (1 + ncon) + j] = H[b * ntok + (j == 0 ? 0 : ntok - ncon + j - 1)]: the rows the hashing head reads (CLS + concept tokens)
__global__ void gather_head_rows_kernel(const float *__restrict__ H, int ntok, int ncon, int D, float *__restrict__ out) {
    const int r = blockIdx.x, nq = 1 + ncon;
    const int b = r / nq, j = r - b * nq;
    const float *src = H + ((size_t)b * ntok + (j == 0 ? 0 : ntok - ncon + j - 1)) * D;
    float *dst = out + (size_t)r * D;
    for (int i = threadIdx.x * 4; i < D; i += blockDim.x * 4) *(f32x4 *)(dst + i) = *(const f32x4 *)(src + i);
}

}  // namespace

int ch_gather_head_rows(const float *H, int B, int ntok, int ncon, int D, float *out, hipStream_t s) {
    CH_REQUIRE(B > 0 && ncon >= 1 && ncon < ntok && D % 4 == 0, "gather_head_rows: bad shape");
    hipLaunchKernelGGL(gather_head_rows_kernel, dim3(B * (1 + ncon)), dim3(192), 0, s, H, ntok, ncon, D, out);
    CH_LAUNCH_CHECK();
    return 0;
}

int ch_layernorm_f32(const float *x, int64_t rows, int D, const float *w, const float *b, float eps, bf16_t *out,
                     hipStream_t s) {
    CH_REQUIRE(D % 128 == 0 && D <= 128 * MAXP, "layernorm: D must be a multiple of 128 and <= 1280");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(layernorm_kernel<false>, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, s, x, rows, D, w, b,
                       eps, out);
    CH_LAUNCH_CHECK();
    return 0;
}

int ch_layernorm_bf16(const bf16_t *x, int64_t rows, int D, const float *w, const float *b, float eps, bf16_t *out,
                      hipStream_t s) {
    CH_REQUIRE(D % 128 == 0 && D <= 128 * MAXP, "layernorm: D must be a multiple of 128 and <= 1280");
    if (rows == 0) return 0;
    if (D <= 1024) {
        hipLaunchKernelGGL(layernorm_bf16_wide_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, s, x, rows, D, w, b,
                           eps, out);
        CH_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(layernorm_kernel<true>, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, s, x, rows, D, w, b,
                       eps, out);
    CH_LAUNCH_CHECK();
    return 0;
}

int ch_assemble_preln(float *H, int B, int ntok, int np, int D, const float *cls_pos0, const float *ctx,
                      const float *pre_w, const float *pre_b, const float *ln_w, const float *ln_b, float eps,
                      bf16_t *xn, hipStream_t s) {
    CH_REQUIRE(D % 128 == 0 && D <= 128 * MAXP, "assemble: D must be a multiple of 128 and <= 1280");
    const int64_t rows = (int64_t)B * ntok;
    hipLaunchKernelGGL(assemble_preln_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, s, H, rows, ntok, np, D,
                       cls_pos0, ctx, pre_w, pre_b, ln_w, ln_b, eps, xn);
    CH_LAUNCH_CHECK();
    return 0;
}

int ch_im2col(const void *images, int image_dtype, int B, int image, int patch, int Kp, bf16_t *out, hipStream_t s) {
    const int grid = image / patch;
    const int K = 3 * patch * patch;
    CH_REQUIRE(Kp % 8 == 0 && Kp >= K, "im2col: Kp must be >= 3*patch^2 and a multiple of 8");
    const int64_t total = (int64_t)B * grid * grid * (Kp / 8);
    const unsigned blocks = (unsigned)ceil_div64(total, 256);
    const bool fast = patch % 8 == 0 && image % 8 == 0 && ((uintptr_t)images & 15) == 0;
    if (image_dtype == 0)
        hipLaunchKernelGGL(im2col_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float *)images, B, image, patch,
                           grid, K, Kp, out, fast);
    else
        hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t *)images, B, image,
                           patch, grid, K, Kp, out, fast);
    CH_LAUNCH_CHECK();
    return 0;
}
