// Fused GEMM epilogues with LDS-staged, full-line global stores (shared by gemm_bf16.hip and gemm_pp.hip).
//
// After the K loop a wave holds its (MT*16) x 64 output sub-tile as acc[4][MT] in the MFMA C/D layout of
// v_mfma_f32_16x16x32_bf16 with the WEIGHT tile as the A operand: lane (fr = lane & 15, fq = lane >> 4) owns rows
// m = mt*16 + fr and, per n-tile nt, the four CONTIGUOUS columns n = nt*16 + fq*4 + 0..3.  Storing that directly costs
// one 8-byte (bf16) store per lane per tile that touches 16 different rows -> 32-byte row fragments; measured on MI355X the
// direct form is store-issue bound at ~10 us per 256x256 tile (a third of the fc1 GEMM).  Here each wave transposes
// through its own 16 KB of the (now idle) staging LDS and writes whole 128-byte lines with 16-byte per-lane accesses:
//   bf16 outputs:  rows of 128 B in LDS, 16-B chunk index XOR (row & 7); read back lane-linear, one dwordx4 store per
//                  8 rows x 128 B per wave-instruction;
//   fp32 residual: rows of 256 B, 16-B chunk index XOR (row & 15) (conflict-free ds_write_b128); read back lane-linear,
//                  read-modify-write of H with dwordx4 (4 rows x 256 B per wave-instruction), 64 rows per pass.
// Wave-local: no workgroup barrier inside; the caller guarantees (a barrier) that no wave still reads K-loop LDS data.
#pragma once
#include "ch_common.h"
#include "kernels.h"

namespace ch_epi {

// quick_gelu(x) = x * sigmoid(1.702 x) (HF QuickGELUActivation): one v_exp_f32 + one v_rcp_f32 (1 ulp) instead of an IEEE
// division -- the epilogue of fc1 evaluates 128 of these per lane per tile.
__device__ __forceinline__ float quick_gelu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930157f * x));  // 1.702 * log2(e)
}
// exact (erf) GELU, nn.GELU() default, with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16
// rounding of the output) instead of libm erff (~3x the instructions).
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

template <int EPI>
__device__ __forceinline__ f32x4 activate(f32x4 v) {
    if constexpr (EPI == EPI_BIAS_QUICKGELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = quick_gelu_f(v[r]);
    }
    if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf_f(v[r]);
    }
    return v;
}

// wave_lds: this wave's 16 KB staging region; m_base / n_base: global row / column of the wave's sub-tile origin.
template <int EPI, int MT>
__device__ __forceinline__ void store_tile(const GemmParams &p, f32x4 (&acc)[4][MT], char *wave_lds, int m_base, int n_base,
                                           int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    constexpr bool BF16_ONLY = (EPI == EPI_BIAS || EPI == EPI_BIAS_QUICKGELU || EPI == EPI_BIAS_GELU);
    f32x4 bias[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        if constexpr (EPI != EPI_PATCH)
            bias[nt] = *(const f32x4 *)(p.bias + n_base + nt * 16 + fq * 4);
        else
            bias[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    if constexpr (BF16_ONLY) {
        // ---- stage MT*16 rows x 64 cols of bf16 (128-B rows)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = mt * 16 + fr;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4 v = activate<EPI>(acc[nt][mt] + bias[nt]);
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *(uint2 *)(wave_lds + row * 128 + (((nt * 2 + (fq >> 1)) ^ (row & 7)) << 4) + (fq & 1) * 8) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int lrow = lane >> 3, pos = lane & 7;
#pragma unroll
        for (int i = 0; i < MT * 2; ++i) {
            const int row = i * 8 + lrow;
            const int chunk = pos ^ (row & 7);
            const uint4 v = *(const uint4 *)(wave_lds + row * 128 + pos * 16);
            const int m = m_base + row;
            if (m < p.M) *(uint4 *)(p.out_bf16 + (size_t)m * p.ldo + n_base + chunk * 8) = v;
        }
    } else {
        // ---- fp32 staging, 64 rows (4 m-tiles) per pass: 256-B rows
        float scale = 1.0f;
        if constexpr (EPI == EPI_SCALE_RESID) scale = *p.scale_ptr;
        const int lrow = lane >> 4, pos = lane & 15;
#pragma unroll
        for (int pass = 0; pass < MT / 4; ++pass) {
            if (pass) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // previous pass fully read before overwrite
#pragma unroll
            for (int mt4 = 0; mt4 < 4; ++mt4) {
                const int row = mt4 * 16 + fr;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const f32x4 v = acc[nt][pass * 4 + mt4] + bias[nt];
                    *(f32x4 *)(wave_lds + row * 256 + (((nt * 4 + fq) ^ (row & 15)) << 4)) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // Two batches of 8 wave-instructions: ALL global loads of a batch are issued before the first store (the
            // compiler cannot hoist them itself: loads and stores of `resid` may alias), so a wave has 8-16 requests in
            // flight instead of one dependent round trip per row group.
#pragma unroll
            for (int batch = 0; batch < 2; ++batch) {
                f32x4 hv[8], lv[8];
                uint2 av[8];
                size_t off[8];
                bool ok[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = (batch * 8 + i) * 4 + lrow;
                    const int chunk = pos ^ (row & 15);
                    const int m = m_base + pass * 64 + row;
                    const int n = n_base + chunk * 4;
                    ok[i] = m < p.M;
                    const int mc = ok[i] ? m : p.M - 1;  // clamp: padding rows read a valid row, never stored
                    lv[i] = *(const f32x4 *)(wave_lds + row * 256 + pos * 16);
                    if constexpr (EPI == EPI_PATCH) {
                        const int img = mc / p.patches_per_img, pp = mc - img * p.patches_per_img;
                        off[i] = ((size_t)img * p.tokens_per_img + 1 + pp) * p.ldr + n;
                        hv[i] = *(const f32x4 *)(p.pos + (size_t)(1 + pp) * p.N + n);
                    } else {
                        off[i] = (size_t)mc * p.ldr + n;
                        hv[i] = *(const f32x4 *)(p.resid + off[i]);
                        if constexpr (EPI == EPI_SCALE_RESID) {
                            av[i] = make_uint2(0u, 0u);
                            if (p.addend) av[i] = *(const uint2 *)(p.addend + (size_t)mc * p.ld_addend + n);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (!ok[i]) continue;
                    const f32x4 v = lv[i];
                    if constexpr (EPI == EPI_PATCH) {
                        *(f32x4 *)(p.resid + off[i]) = v + hv[i];
                    } else if constexpr (EPI == EPI_BIAS_RESID) {
                        *(f32x4 *)(p.resid + off[i]) = hv[i] + v;
                        const int row = (batch * 8 + i) * 4 + lrow;
                        const int m = m_base + pass * 64 + row;
                        const int n = n_base + (pos ^ (row & 15)) * 4;
                        uint2 o;
                        o.x = pack_bf16x2(v[0], v[1]);
                        o.y = pack_bf16x2(v[2], v[3]);
                        *(uint2 *)(p.out_bf16 + (size_t)m * p.ldo + n) = o;
                    } else {  // EPI_SCALE_RESID: H += [addend] + scale * (acc + bias)
                        f32x4 h = hv[i] + v * scale;
                        h[0] += bf2f((bf16_t)(av[i].x & 0xffff));
                        h[1] += bf2f((bf16_t)(av[i].x >> 16));
                        h[2] += bf2f((bf16_t)(av[i].y & 0xffff));
                        h[3] += bf2f((bf16_t)(av[i].y >> 16));
                        *(f32x4 *)(p.resid + off[i]) = h;
                    }
                }
            }
        }
    }
}

}  // namespace ch_epi
