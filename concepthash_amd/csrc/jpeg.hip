// GPU half of the JPEG decode split (the host half -- marker parsing, Huffman / progressive entropy decode, file reads -- is
// jpeg_host.cpp, where the split, the restated libjpeg-turbo arithmetic and the supported subset are described): dequantisation + 8x8
// inverse DCT (`jpeg_idct_islow`) + fancy chroma upsampling + YCbCr -> RGB on int16 coefficient blocks, bytes bit-equal to Pillow's.
#include <algorithm>

#include "../../include/concepthash_hip.h"
#include "ch_common.h"

namespace {

inline int64_t blocks_of(const ch_jpeg_desc &d) { return d.nblocks; }

// ---------------------------------------------------------------------------------------------------------------------------------
// GPU: dequantise + jpeg_idct_islow, one thread per 8x8 block -> component planes (uint8)
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int CONST_BITS = 13, PASS1_BITS = 2;
constexpr int F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373,
              F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069, F_2_053119869 = 16819,
              F_2_562915447 = 20995, F_3_072711026 = 25172;

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
// the post-IDCT range-limit table of jdmaster.c (prepare_range_limit_table), index = value & RANGE_MASK (1023), level shift included
__device__ __forceinline__ int idct_limit(int x) {
    const int v = x & 1023;
    return v < 128 ? v + 128 : v < 512 ? 255 : v < 896 ? 0 : v - 896;
}

// one 1-D pass of jidctint.c on eight (dequantised) values; SHIFT = CONST_BITS - PASS1_BITS in pass 1, CONST_BITS + PASS1_BITS + 3 in pass 2
template <int SHIFT>
__device__ __forceinline__ void idct_1d(const int (&in)[8], int (&out)[8]) {
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * F_0_541196100;
    int tmp2 = z1 + z3 * (-F_1_847759065);
    int tmp3 = z1 + z2 * F_0_765366865;
    z2 = in[0];
    z3 = in[4];
    int tmp0 = (z2 + z3) << CONST_BITS;
    int tmp1 = (z2 - z3) << CONST_BITS;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7];
    tmp1 = in[5];
    tmp2 = in[3];
    tmp3 = in[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * F_1_175875602;
    tmp0 *= F_0_298631336;
    tmp1 *= F_2_053119869;
    tmp2 *= F_3_072711026;
    tmp3 *= F_1_501321110;
    z1 *= -F_0_899976223;
    z2 *= -F_2_562915447;
    z3 *= -F_1_961570560;
    z4 *= -F_0_390180644;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    out[0] = descale(tmp10 + tmp3, SHIFT);
    out[7] = descale(tmp10 - tmp3, SHIFT);
    out[1] = descale(tmp11 + tmp2, SHIFT);
    out[6] = descale(tmp11 - tmp2, SHIFT);
    out[2] = descale(tmp12 + tmp1, SHIFT);
    out[5] = descale(tmp12 - tmp1, SHIFT);
    out[3] = descale(tmp13 + tmp0, SHIFT);
    out[4] = descale(tmp13 - tmp0, SHIFT);
}

__global__ __launch_bounds__(128) void jpeg_idct_kernel(const int16_t *__restrict__ coef, const ch_jpeg_desc *__restrict__ descs,
                                                        uint8_t *__restrict__ planes) {
    const ch_jpeg_desc &d = descs[blockIdx.y];
    if (d.status) return;
    const int ybw = d.mcu_w * d.hs, ybh = d.mcu_h * d.vs;
    const int nby = ybw * ybh, nbc = d.mcu_w * d.mcu_h;
    const int total = d.ncomp == 3 ? nby + 2 * nbc : nby;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= total) return;
    int comp = 0, local = b, bw = ybw;
    int64_t plane_off = d.plane_offset;
    if (b >= nby) {
        comp = 1 + (b - nby) / nbc;
        local = (b - nby) - (comp - 1) * nbc;
        bw = d.mcu_w;
        plane_off += (int64_t)nby * 64 + (int64_t)(comp - 1) * nbc * 64;
    }
    const int by = local / bw, bx = local - by * bw;
    // 64 coefficients and 64 table entries as eight 16-byte loads each (both 16-byte aligned: blocks are 128 bytes, quant sits at
    // offset 64 of the descriptor)
    union Row {
        uint4 v;
        int16_t s[8];
        uint16_t u[8];
    } cr[8], qr[8];
    const uint4 *c4 = (const uint4 *)(coef + d.coef_offset + (int64_t)b * 64), *q4 = (const uint4 *)d.quant[comp];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        cr[r].v = c4[r];
        qr[r].v = q4[r];
    }
    int ws[8][8];   // [row][col] after pass 1
#pragma unroll
    for (int col = 0; col < 8; ++col) {
        int in[8], out[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) in[r] = (int)cr[r].s[col] * (int)qr[r].u[col];
        idct_1d<CONST_BITS - PASS1_BITS>(in, out);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[r][col] = out[r];
    }
    uint8_t *dst = planes + plane_off + ((int64_t)by * 8 * bw + bx) * 8;
    const int pitch = bw * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int out[8];
        idct_1d<CONST_BITS + PASS1_BITS + 3>(ws[r], out);
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lo |= (uint32_t)idct_limit(out[i]) << (8 * i);
            hi |= (uint32_t)idct_limit(out[4 + i]) << (8 * i);
        }
        *(uint2 *)(dst + (int64_t)r * pitch) = make_uint2(lo, hi);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// GPU: fancy upsampling (jdsample.c) + YCbCr -> RGB (jdcolor.c), one thread per output pixel
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

// chroma sample at full-resolution position (x, y) of a plane of dw x dh REAL samples (pitch >= dw)
__device__ __forceinline__ int chroma_at(const uint8_t *pl, int pitch, int dw, int dh, int hs, int vs, int x, int y) {
    if (hs == 1) return pl[(int64_t)y * pitch + x];   // 4:4:4
    const int i = x >> 1, odd = x & 1;
    if (vs == 1) {   // h2v1_fancy_upsample
        const uint8_t *row = pl + (int64_t)y * pitch;
        const int v = row[i];
        if (odd) return i == dw - 1 ? v : (3 * v + row[i + 1] + 2) >> 2;
        return i == 0 ? v : (3 * v + row[i - 1] + 1) >> 2;
    }
    // h2v2_fancy_upsample: the nearer input row weighs 3, the farther 1; the row above the first / below the last REAL sample row
    // is that row itself (jdmainct.c: make_funny_pointers / set_bottom_pointers)
    const int r = y >> 1;
    int far = (y & 1) ? r + 1 : r - 1;
    far = far < 0 ? 0 : far > dh - 1 ? dh - 1 : far;
    const uint8_t *n0 = pl + (int64_t)r * pitch, *n1 = pl + (int64_t)far * pitch;
    const int cs = 3 * n0[i] + n1[i];
    if (odd) {
        if (i == dw - 1) return (cs * 4 + 7) >> 4;
        return (cs * 3 + (3 * n0[i + 1] + n1[i + 1]) + 7) >> 4;
    }
    if (i == 0) return (cs * 4 + 8) >> 4;
    return (cs * 3 + (3 * n0[i - 1] + n1[i - 1]) + 8) >> 4;
}

__device__ __forceinline__ void ycc_to_rgb(int Y, int cb, int cr, uint8_t *rgb) {
    // jdcolor.c build_ycc_rgb_table: SCALEBITS 16, FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802, FIX(0.34414) = 22554
    rgb[0] = (uint8_t)clamp255(Y + ((91881 * cr + 32768) >> 16));
    rgb[1] = (uint8_t)clamp255(Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
    rgb[2] = (uint8_t)clamp255(Y + ((116130 * cb + 32768) >> 16));
}

// One thread per FOUR horizontally adjacent pixels (x0 = 4 t), a workgroup = 64 threads x 4 rows (256 pixels x 4 rows: a 500-pixel row
// wastes 2 % of the lanes; one 1,024-pixel strip per workgroup wasted half of them; 32 x 8 threads measured 6 % slower): the luma bytes come as one 32-bit load, the twelve
// output bytes leave as three 32-bit stores when the row start allows it (one thread per pixel issued three strided byte stores per
// lane: 281 us per 256 images of 500 x 375 for 190 MB -- six times the traffic floor).  The four pixels share their chroma
// neighbourhood: samples i - 1 .. i + 2 (i = x0 / 2) of one row (h2v1) or of the nearer and the farther row (h2v2), loaded once with
// the index CLAMPED to the plane -- which is jdsample.c's edge rule exactly: (3 c + c + 8) >> 4 == (4 c + 8) >> 4 at the left edge,
// (3 c + c + 7) >> 4 == (4 c + 7) >> 4 at the right one, (3 v + v + 1|2) >> 2 == v for h2v1.
__global__ __launch_bounds__(256) void jpeg_color_kernel(const ch_jpeg_desc *__restrict__ descs, const uint8_t *__restrict__ planes,
                                                         uint8_t *__restrict__ pixels) {
    const ch_jpeg_desc &d = descs[blockIdx.z];
    if (d.status) return;
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= d.width || y >= d.height) return;
    const int npx = min(4, d.width - x0);
    const int ypitch = d.mcu_w * d.hs * 8, yrows = d.mcu_h * d.vs * 8;
    const uint8_t *py = planes + d.plane_offset;
    // the plane rows are padded to whole MCUs (a multiple of 8 bytes) and plane_offset is a multiple of 16: an aligned 32-bit load
    const uint32_t yw = *(const uint32_t *)(py + (int64_t)y * ypitch + x0);
    uint8_t rgb[12];
    if (d.ncomp == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) rgb[3 * j] = rgb[3 * j + 1] = rgb[3 * j + 2] = (uint8_t)(yw >> (8 * j));
    } else {
        const int cpitch = d.mcu_w * 8;
        const uint8_t *pcb = py + (int64_t)ypitch * yrows, *pcr = pcb + (int64_t)cpitch * d.mcu_h * 8;
        const int dw = (d.width + d.hs - 1) / d.hs, dh = (d.height + d.vs - 1) / d.vs;
        int cbv[4], crv[4];
        if (d.hs == 2 && d.vs <= 2) {
            const int i = x0 >> 1;
            int col[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) col[k] = min(max(i - 1 + k, 0), dw - 1);
            int sb[4], sr[4], rnd_even, rnd_odd, shift;
            if (d.vs == 1) {   // h2v1_fancy_upsample
                const int64_t r0 = (int64_t)y * cpitch;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    sb[k] = pcb[r0 + col[k]];
                    sr[k] = pcr[r0 + col[k]];
                }
                rnd_even = 1, rnd_odd = 2, shift = 2;
            } else {           // h2v2_fancy_upsample: column sums 3 * nearer row + farther row (chroma_at)
                const int r = y >> 1;
                int far = (y & 1) ? r + 1 : r - 1;
                far = far < 0 ? 0 : far > dh - 1 ? dh - 1 : far;
                const int64_t r0 = (int64_t)r * cpitch, r1 = (int64_t)far * cpitch;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    sb[k] = 3 * pcb[r0 + col[k]] + pcb[r1 + col[k]];
                    sr[k] = 3 * pcr[r0 + col[k]] + pcr[r1 + col[k]];
                }
                rnd_even = 8, rnd_odd = 7, shift = 4;
            }
            // pixel x0 + j sits on sample i + (j >> 1) = s[1 + (j >> 1)]; an even pixel leans on the sample to its left, an odd one right
            cbv[0] = (3 * sb[1] + sb[0] + rnd_even) >> shift; crv[0] = (3 * sr[1] + sr[0] + rnd_even) >> shift;
            cbv[1] = (3 * sb[1] + sb[2] + rnd_odd) >> shift;  crv[1] = (3 * sr[1] + sr[2] + rnd_odd) >> shift;
            cbv[2] = (3 * sb[2] + sb[1] + rnd_even) >> shift; crv[2] = (3 * sr[2] + sr[1] + rnd_even) >> shift;
            cbv[3] = (3 * sb[2] + sb[3] + rnd_odd) >> shift;  crv[3] = (3 * sr[2] + sr[3] + rnd_odd) >> shift;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = min(x0 + j, d.width - 1);     // (lanes past the right edge recompute the last pixel; not stored)
                cbv[j] = chroma_at(pcb, cpitch, dw, dh, d.hs, d.vs, x, y);
                crv[j] = chroma_at(pcr, cpitch, dw, dh, d.hs, d.vs, x, y);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) ycc_to_rgb((int)((yw >> (8 * j)) & 255), cbv[j] - 128, crv[j] - 128, rgb + 3 * j);
    }
    uint8_t *out = pixels + d.pix_offset + ((int64_t)y * d.width + x0) * 3;
    if (npx == 4 && (((uintptr_t)out) & 3) == 0) {
        uint32_t w[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) w[k] = rgb[4 * k] | (rgb[4 * k + 1] << 8) | (rgb[4 * k + 2] << 16) | ((uint32_t)rgb[4 * k + 3] << 24);
        *(uint32_t *)out = w[0];
        *(uint32_t *)(out + 4) = w[1];
        *(uint32_t *)(out + 8) = w[2];
    } else {
        for (int k = 0; k < 3 * npx; ++k) out[k] = rgb[k];
    }
}

}  // namespace

extern "C" int ch_jpeg_reconstruct(const int16_t *coef_dev, const ch_jpeg_desc *desc_dev, const ch_jpeg_desc *desc_host, int32_t n,
                                   uint8_t *planes_ws, uint8_t *pixels, void *stream) {
    CH_REQUIRE(n >= 0, "jpeg_reconstruct: negative image count");
    if (n == 0) return 0;
    CH_REQUIRE(coef_dev && desc_dev && desc_host && planes_ws && pixels, "jpeg_reconstruct: null argument");
    CH_REQUIRE(n <= 65535, "jpeg_reconstruct: at most 65535 images per call");
    int64_t max_blocks = 0;
    int max_w = 0, max_h = 0;
    for (int i = 0; i < n; ++i) {
        if (desc_host[i].status) continue;
        max_blocks = std::max<int64_t>(max_blocks, blocks_of(desc_host[i]));
        max_w = std::max(max_w, desc_host[i].width);
        max_h = std::max(max_h, desc_host[i].height);
    }
    if (max_blocks == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((max_blocks + 127) / 128), n), dim3(128), 0, s, coef_dev, desc_dev, planes_ws);
    CH_LAUNCH_CHECK();
    CH_REQUIRE(max_h <= 4 * 65535, "jpeg_reconstruct: image taller than 262140 rows");
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((max_w + 255) / 256, (max_h + 3) / 4, n), dim3(64, 4), 0, s, desc_dev, planes_ws, pixels);
    CH_LAUNCH_CHECK();
    return 0;
}
