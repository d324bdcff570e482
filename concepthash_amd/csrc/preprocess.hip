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
__device__ __forceinline__ int2 coeffs_for(int in_size, int out_size, int xx, int *kk, int kstride, int kcap = KMAX) {
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
    if (xmax > kcap) xmax = kcap;  // unreachable: the host rejects scales that need more taps than the kernel it picks holds
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
            s0 += __mul24((int)row[t * 3 + 0], k);   // byte x 23-bit weight: the 24-bit multiplier (v_mul_lo_u32 is quarter rate)
            s1 += __mul24((int)row[t * 3 + 1], k);
            s2 += __mul24((int)row[t * 3 + 2], k);
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
        a0 += __mul24((int)p[0], k);
        a1 += __mul24((int)p[1], k);
        a2 += __mul24((int)p[2], k);
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


// ---------------------------------------------------------------------------------------------------------------------------------
// The same two passes for the shapes the loaders produce (crop a multiple of 4, at most 16 taps per output column), re-cut around
// what bounded the kernels above on MI355X (256 images of 500 x 375: 214 + 128 us for 220 MB): one byte per lane per memory
// instruction -- 21 loads and 3 stores per output pixel in either pass --, 32-bit multiplies (v_mul_lo_u32: quarter rate) where the
// 24-bit multiplier is exact, in the vertical pass one lane's double-precision coefficient set-up per output row with the other 255
// waiting, and 64 KB of LDS for 7 taps in the horizontal pass (two workgroups per CU).  Measured: 65 + 45 us.
//   horizontal: a lane reads its 8 / 16 taps (24 / 48 bytes) as the aligned dwords that hold them -- two / four multi-dword loads per
//     row instead of 21 / 48 byte loads; neighbouring lanes overlap, the L1 serves them -- and funnel-shifts them into place,
//     coefficients live in registers (8 / 16 KB of LDS only hands them over), and the three result bytes of four neighbouring lanes
//     leave as three dword stores (a lane exchange, no LDS); the next row's loads fly under this row's arithmetic.  No barrier in the kernel;
//   vertical: a lane owns four pixels = 12 bytes of the intermediate row (one dwordx3 load per tap: the vertical weight is the same
//     for every byte of a row), 16 output rows per workgroup whose coefficients 16 lanes compute side by side, ToTensor + normalise
//     through a 768-entry table (byte, channel) -> output value built with the IEEE divisions the generic kernel performs per pixel.
// Same integer arithmetic per output byte, same coefficient code: bit-equal to the kernels above (tests/test_preprocess.py runs both).
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int HF_ROWS = 64;                 // source rows per workgroup (amortises the per-lane coefficient set-up)
constexpr int VF_ROWS = 16;                 // output rows per workgroup

template <int KT>
__global__ __launch_bounds__(256) void resize_h_fast_kernel(const uint8_t *__restrict__ pixels, const ch_image_desc *__restrict__ desc,
                                                            int crop, uint8_t *__restrict__ tmp) {
    __shared__ int hand[KT * 256];   // coefficient hand-over [tap][lane] (coeffs_for indexes its output at run time)
    const ch_image_desc d = desc[blockIdx.x];
    const int x = threadIdx.x;
    const int rows_per_block = (d.nrows + (int)gridDim.y - 1) / (int)gridDim.y;   // <= HF_ROWS: the image's rows in equal shares
    const int r_begin = blockIdx.y * rows_per_block;
    if (r_begin >= d.nrows) return;  // block-uniform
    int2 b = make_int2(0, 0);
    if (x < crop) b = coeffs_for(d.w, d.nw, x + d.left, hand + x, 256, KT);
    int k[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) k[t] = t < b.y ? hand[t * 256 + x] : 0;    // own column only: no barrier
    const int stride = d.stride ? d.stride : d.w;   // a crop box inside a wider image (training transforms)
    // byte offsets from `pixels` (4-byte aligned, the host checks): an aligned dword address is pixels + (offset & ~3)
    const int64_t first = d.src_offset + (int64_t)b.x * 3;                                   // this lane's first tap in source row 0
    const int64_t safe_end = d.src_offset + ((int64_t)(d.h - 1) * stride + d.w) * 3;        // one past the image's (the crop box's) last byte
    uint8_t *dst = tmp + d.tmp_offset;
    const int r_end = min(r_begin + rows_per_block, d.nrows);
    const int j = x & 3;
    constexpr int NW = (KT * 3 + 3) / 4;             // dwords that hold KT taps; one more feeds the funnel shift
    auto fetch = [&](int r, uint32_t (&dw)[NW + 1]) {
        const int64_t off = (first + (int64_t)(d.row0 + r) * stride * 3) & ~(int64_t)3;
        const uint32_t *w = (const uint32_t *)(pixels + off);
        if (off + 4 * (NW + 1) <= safe_end) {
#pragma unroll
            for (int i = 0; i <= NW; ++i) dw[i] = w[i];
        } else {                                     // the last lanes of the image's last row(s): byte by byte, inside the buffer
#pragma unroll
            for (int i = 0; i <= NW; ++i) {
                dw[i] = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (off + 4 * i + q < safe_end) dw[i] |= (uint32_t)pixels[off + 4 * i + q] << (8 * q);
            }
        }
    };
    uint32_t cur[NW + 1], nxt[NW + 1];
    fetch(r_begin, cur);
    for (int r = r_begin; r < r_end; ++r) {
        if (r + 1 < r_end) fetch(r + 1, nxt);        // the next row's loads fly under this row's arithmetic
        const uint32_t sh = (uint32_t)((first + (int64_t)(d.row0 + r) * stride * 3) & 3) * 8;
        uint32_t al[NW];
#pragma unroll
        for (int i = 0; i < NW; ++i) al[i] = __builtin_amdgcn_alignbit(cur[i + 1], cur[i], sh);
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
#pragma unroll
        for (int t = 0; t < KT; ++t) {               // byte x 23-bit weight: the 24-bit multiplier (v_mul_lo_u32 is quarter rate)
            s0 += __mul24((int)((al[(3 * t) >> 2] >> (8 * ((3 * t) & 3))) & 255), k[t]);
            s1 += __mul24((int)((al[(3 * t + 1) >> 2] >> (8 * ((3 * t + 1) & 3))) & 255), k[t]);
            s2 += __mul24((int)((al[(3 * t + 2) >> 2] >> (8 * ((3 * t + 2) & 3))) & 255), k[t]);
        }
        const uint32_t v = (uint32_t)clip8(s0) | ((uint32_t)clip8(s1) << 8) | ((uint32_t)clip8(s2) << 16);
        const uint32_t vn = (uint32_t)__shfl_down((int)v, 1);    // the next lane of the quad (crop % 4 == 0: whole quads)
        if (x < crop && j < 3)
            *(uint32_t *)(dst + (size_t)r * crop * 3 + (x >> 2) * 12 + 4 * j) = (v >> (8 * j)) | (vn << (24 - 8 * j));
#pragma unroll
        for (int i = 0; i <= NW; ++i) cur[i] = nxt[i];
    }
}

template <typename OUT>
__global__ __launch_bounds__(256) void resize_v_fast_kernel(const uint8_t *__restrict__ tmp, const ch_image_desc *__restrict__ desc, int crop,
                                                            float m0, float m1, float m2, float s0, float s1, float s2,
                                                            OUT *__restrict__ out) {
    __shared__ int kk[KMAX * VF_ROWS];   // [tap][row]
    __shared__ int2 bounds[VF_ROWS];
    __shared__ OUT lut[3 * 256];         // (channel, byte) -> ToTensor + normalise, the generic kernel's IEEE operations
    const ch_image_desc d = desc[blockIdx.x];
    if (d.nrows == 0) return;            // block-uniform: an image the host has marked as its own
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int y0 = blockIdx.y * VF_ROWS;
    if (tid < VF_ROWS && y0 + tid < crop) bounds[tid] = coeffs_for(d.h, d.nh, y0 + tid + d.top, kk + tid, VF_ROWS);
    for (int i = tid; i < 3 * 256; i += 256) {
        const int c = i >> 8;
        const float v = ((float)(i & 255) / 255.0f - (c == 0 ? m0 : c == 1 ? m1 : m2)) / (c == 0 ? s0 : c == 1 ? s1 : s2);
        if constexpr (sizeof(OUT) == 2) lut[i] = f2bf(v);
        else lut[i] = v;
    }
    __syncthreads();
    const int x0 = threadIdx.x * 4;
    if (x0 >= crop) return;
    const size_t plane = (size_t)crop * crop;
    for (int ry = threadIdx.y; ry < VF_ROWS && y0 + ry < crop; ry += 4) {
        const int ymin = bounds[ry].x - d.row0, n = bounds[ry].y;
        const uint8_t *src = tmp + d.tmp_offset + (size_t)ymin * crop * 3 + x0 * 3;
        int acc[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] = 1 << (PRECISION_BITS - 1);
        for (int t = 0; t < n; ++t) {
            const int k = kk[t * VF_ROWS + ry];
            const uint3 w = *(const uint3 *)(src + (size_t)t * crop * 3);
            const uint32_t ww[3] = {w.x, w.y, w.z};
#pragma unroll
            for (int i = 0; i < 12; ++i) acc[i] += __mul24((int)((ww[i >> 2] >> (8 * (i & 3))) & 255), k);   // 24-bit multiplier
        }
        OUT v[3][4];                     // [channel][pixel]
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c][d.flip ? 3 - p : p] = lut[c * 256 + clip8(acc[3 * p + c])];
        OUT *o = out + (size_t)blockIdx.x * 3 * plane + (size_t)(y0 + ry) * crop + (d.flip ? crop - 4 - x0 : x0);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if constexpr (sizeof(OUT) == 2)
                *(uint2 *)(o + c * plane) = make_uint2((uint32_t)v[c][0] | ((uint32_t)v[c][1] << 16), (uint32_t)v[c][2] | ((uint32_t)v[c][3] << 16));
            else
                *(float4 *)(o + c * plane) = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
        }
    }
}

}  // namespace

extern "C" int ch_preprocess(const uint8_t *pixels, const ch_image_desc *desc_device, int32_t B, int32_t max_rows, int32_t max_taps,
                             int32_t crop, const float *mean3_host, const float *std3_host, void *out, int32_t out_dtype,
                             uint8_t *workspace, void *stream) {
    CH_REQUIRE(B >= 0 && crop >= 1 && crop <= 256, "preprocess: crop must be in [1, 256]");
    if (B == 0) return 0;
    CH_REQUIRE(pixels && desc_device && mean3_host && std3_host && out && workspace, "preprocess: null pointer");
    CH_REQUIRE(max_rows >= 1, "preprocess: max_rows must be >= 1");
    CH_REQUIRE(max_taps >= 0 && max_taps <= KMAX, "preprocess: max_taps must be in [0, ch_preprocess_max_taps()]");
    CH_REQUIRE(out_dtype == 0 || out_dtype == 1, "preprocess: out_dtype must be 0 (fp32) or 1 (bf16)");
    hipStream_t s = (hipStream_t)stream;
    // the dword forms: whole quads of columns, dword-aligned intermediate rows, 16-byte-aligned output rows
    const bool quads = crop % 4 == 0 && ((uintptr_t)workspace & 3) == 0 && ((uintptr_t)out & 15) == 0;
    const bool hfast = quads && ((uintptr_t)pixels & 3) == 0 && max_taps >= 1;   // (its aligned dword reads start inside the buffer)
    if (hfast && max_taps <= 8)
        hipLaunchKernelGGL(resize_h_fast_kernel<8>, dim3((unsigned)B, (unsigned)ceil_div64(max_rows, HF_ROWS)), dim3(256), 0, s, pixels,
                           desc_device, crop, workspace);
    else if (hfast && max_taps <= 16)
        hipLaunchKernelGGL(resize_h_fast_kernel<16>, dim3((unsigned)B, (unsigned)ceil_div64(max_rows, HF_ROWS)), dim3(256), 0, s, pixels,
                           desc_device, crop, workspace);
    else
        hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)B, (unsigned)ceil_div64(max_rows, ROWS_PER_BLOCK)), dim3(256), 0, s, pixels,
                           desc_device, crop, workspace);
    CH_LAUNCH_CHECK();
    const float *m = mean3_host, *sd = std3_host;
    if (quads) {
        const dim3 grid((unsigned)B, (unsigned)ceil_div64(crop, VF_ROWS)), block(64, 4);
        if (out_dtype == 1)
            hipLaunchKernelGGL(resize_v_fast_kernel<bf16_t>, grid, block, 0, s, workspace, desc_device, crop, m[0], m[1], m[2], sd[0], sd[1],
                               sd[2], (bf16_t *)out);
        else
            hipLaunchKernelGGL(resize_v_fast_kernel<float>, grid, block, 0, s, workspace, desc_device, crop, m[0], m[1], m[2], sd[0], sd[1],
                               sd[2], (float *)out);
    } else if (out_dtype == 1) {
        hipLaunchKernelGGL(resize_v_kernel<bf16_t>, dim3((unsigned)B, (unsigned)crop), dim3(256), 0, s, workspace, desc_device, crop,
                           m[0], m[1], m[2], sd[0], sd[1], sd[2], (bf16_t *)out);
    } else {
        hipLaunchKernelGGL(resize_v_kernel<float>, dim3((unsigned)B, (unsigned)crop), dim3(256), 0, s, workspace, desc_device, crop,
                           m[0], m[1], m[2], sd[0], sd[1], sd[2], (float *)out);
    }
    CH_LAUNCH_CHECK();
    return 0;
}

extern "C" int32_t ch_preprocess_max_taps(void) { return KMAX; }
