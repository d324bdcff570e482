// Image pre-processing on the GPU: uint8 HWC images of arbitrary size -> Resize(shorter side, bicubic) -> CenterCrop -> /255 ->
// (x - mean) / std -> the NCHW batch the encoder reads (bf16 or fp32); and, with a crop box and a flip flag per image, the training
// chain RandomResizedCrop(bicubic) -> RandomHorizontalFlip -> ToTensor -> normalize (configs/dataset/cub200.yaml:13-23; the random
// draws stay on the host, in the loader workers, so the random stream is the CPU chain's).
//
// Replaces, for the evaluation loop, the CPU-worker transform chain of the reference's dataset configs
// (configs/dataset/cub200.yaml:31-47: torchvision Resize(256, bicubic) -> CenterCrop(224) -> ToTensor -> normalize; loader
// engine.py:41-54).  The arithmetic is the third-party Pillow resampler those transforms call (ImagingResample, Pillow >= 7):
//   * per output index: support = 2 * max(scale, 1) taps of the Keys bicubic (a = -0.5) around (x + 0.5) * scale, normalised in
//     double precision, quantised to 22 fractional bits -- computed here with the same operations in the same order and with
//     floating-point contraction OFF, so the integer coefficients are the ones Pillow derives;
//   * two passes with an 8-bit intermediate: horizontal over the source rows the vertical pass needs, then vertical; every
//     result is (sum + 2^21) >> 22 clamped to [0, 255].
// Only the cropped window is computed (each output pixel of the resize is independent of the others).  Host code supplies the
// per-image integers whose rounding rules belong to Python (torchvision's int(size * long / short), round-half-even crop origin).
// Bit-exact against Pillow by construction; oracle/preprocess_oracle.py restates the same algorithm and is pinned against Pillow.
#pragma clang fp contract(off)
#include "../../include/concepthash_hip.h"
#include "ch_common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;
constexpr int KMAX = 64;      // taps per output index: 2 * ceil(2 * scale) + 1 <= 64 <=> down-scaling up to ~15.5x
constexpr int ROWS_PER_BLOCK = 16;

__device__ __forceinline__ double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// Pillow precompute_coeffs + normalize_coeffs_8bpc for ONE output index xx of a (0, in_size) -> out_size resize.
// Writes up to KMAX int weights to kk (stride kstride) and returns (xmin, count).
__device__ __forceinline__ int2 coeffs_for(int in_size, int out_size, int xx, int *kk, int kstride) {
    const double scale = (double)((float)in_size - 0.0f) / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const double ss = 1.0 / filterscale;
    const double center = 0.0 + (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    if (xmax > KMAX) xmax = KMAX;  // unreachable: the host rejects scales that need more taps
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += bicubic_filter((x + xmin - center + 0.5) * ss);
    for (int x = 0; x < xmax; ++x) {
        double w = bicubic_filter((x + xmin - center + 0.5) * ss);
        if (ww != 0.0) w /= ww;
        kk[x * kstride] = w < 0 ? (int)(-0.5 + w * (double)(1 << PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << PRECISION_BITS));
    }
    return make_int2(xmin, xmax);
}

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: tmp[b][r][x][c] for the source rows r in [row0, row0 + nrows) and the crop's columns x
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t *__restrict__ pixels, const ch_image_desc *__restrict__ desc,
                                                       int crop, uint8_t *__restrict__ tmp) {
    __shared__ int kk[KMAX * 256];  // [tap][thread]: conflict-free
    const ch_image_desc d = desc[blockIdx.x];
    const int x = threadIdx.x;
    const int r_begin = blockIdx.y * ROWS_PER_BLOCK;
    if (r_begin >= d.nrows) return;  // block-uniform
    int2 b = make_int2(0, 0);
    if (x < crop) b = coeffs_for(d.w, d.nw, x + d.left, kk + x, 256);
    if (x >= crop) return;
    const uint8_t *src = pixels + d.src_offset;
    uint8_t *dst = tmp + d.tmp_offset;
    const int r_end = min(r_begin + ROWS_PER_BLOCK, d.nrows);
    const int stride = d.stride ? d.stride : d.w;   // a crop box inside a wider image (training transforms)
    for (int r = r_begin; r < r_end; ++r) {
        const uint8_t *row = src + ((size_t)(d.row0 + r) * stride + b.x) * 3;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int t = 0; t < b.y; ++t) {
            const int k = kk[t * 256 + x];
            s0 += row[t * 3 + 0] * k;
            s1 += row[t * 3 + 1] * k;
            s2 += row[t * 3 + 2] * k;
        }
        uint8_t *o = dst + ((size_t)r * crop + x) * 3;
        o[0] = (uint8_t)clip8(s0);
        o[1] = (uint8_t)clip8(s1);
        o[2] = (uint8_t)clip8(s2);
    }
}

// vertical pass + ToTensor + normalise: out[b][c][y][x]
template <typename OUT>
__global__ __launch_bounds__(256) void resize_v_kernel(const uint8_t *__restrict__ tmp, const ch_image_desc *__restrict__ desc, int crop,
                                                       float m0, float m1, float m2, float s0, float s1, float s2,
                                                       OUT *__restrict__ out) {
    __shared__ int kk[KMAX];
    __shared__ int2 bounds;
    const ch_image_desc d = desc[blockIdx.x];
    const int y = blockIdx.y, x = threadIdx.x;
    if (d.nrows == 0) return;   // block-uniform: an image the host has marked as its own (down-scaling beyond the tap limit)
    if (x == 0) bounds = coeffs_for(d.h, d.nh, y + d.top, kk, 1);
    __syncthreads();
    if (x >= crop) return;
    const int ymin = bounds.x - d.row0, n = bounds.y;
    const uint8_t *src = tmp + d.tmp_offset + ((size_t)ymin * crop + x) * 3;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int t = 0; t < n; ++t) {
        const int k = kk[t];
        const uint8_t *p = src + (size_t)t * crop * 3;
        a0 += p[0] * k;
        a1 += p[1] * k;
        a2 += p[2] * k;
    }
    // ToTensor: uint8 -> float / 255 ; normalise: (v - mean) / std   (IEEE divisions, as torch computes them)
    const float v0 = ((float)clip8(a0) / 255.0f - m0) / s0;
    const float v1 = ((float)clip8(a1) / 255.0f - m1) / s1;
    const float v2 = ((float)clip8(a2) / 255.0f - m2) / s2;
    const size_t plane = (size_t)crop * crop;
    OUT *o = out + (size_t)blockIdx.x * 3 * plane + (size_t)y * crop + (d.flip ? crop - 1 - x : x);
    if constexpr (sizeof(OUT) == 2) {
        o[0] = f2bf(v0);
        o[plane] = f2bf(v1);
        o[2 * plane] = f2bf(v2);
    } else {
        o[0] = v0;
        o[plane] = v1;
        o[2 * plane] = v2;
    }
}

}  // namespace

extern "C" int ch_preprocess(const uint8_t *pixels, const ch_image_desc *desc_device, int32_t B, int32_t max_rows, int32_t crop,
                             const float *mean3_host, const float *std3_host, void *out, int32_t out_dtype, uint8_t *workspace,
                             void *stream) {
    CH_REQUIRE(B >= 0 && crop >= 1 && crop <= 256, "preprocess: crop must be in [1, 256]");
    if (B == 0) return 0;
    CH_REQUIRE(pixels && desc_device && mean3_host && std3_host && out && workspace, "preprocess: null pointer");
    CH_REQUIRE(max_rows >= 1, "preprocess: max_rows must be >= 1");
    CH_REQUIRE(out_dtype == 0 || out_dtype == 1, "preprocess: out_dtype must be 0 (fp32) or 1 (bf16)");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)B, (unsigned)ceil_div64(max_rows, ROWS_PER_BLOCK)), dim3(256), 0, s, pixels,
                       desc_device, crop, workspace);
    CH_LAUNCH_CHECK();
    if (out_dtype == 1)
        hipLaunchKernelGGL(resize_v_kernel<bf16_t>, dim3((unsigned)B, (unsigned)crop), dim3(256), 0, s, workspace, desc_device, crop,
                           mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2], (bf16_t *)out);
    else
        hipLaunchKernelGGL(resize_v_kernel<float>, dim3((unsigned)B, (unsigned)crop), dim3(256), 0, s, workspace, desc_device, crop,
                           mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2], (float *)out);
    CH_LAUNCH_CHECK();
    return 0;
}

extern "C" int32_t ch_preprocess_max_taps(void) { return KMAX; }
