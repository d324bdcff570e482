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

// The fp32 residual stream is read once and written once per adapter (158 MB each way at batch 256) and not touched again before
// ~800 MB of other traffic has passed: when it is larger than what the 256 MB Infinity Cache can usefully keep (the dispatcher
// decides from the tensor's size and launches the NT instance) the read-modify-write of the scale+residual epilogues goes
// NON-TEMPORAL.  A template parameter, not a run-time flag: with a uniform branch around the two flavours the compiler hoists the
// common load out of both arms and the cache-policy bits are lost (measured: no effect at all).  Measured
// at batch 256: adapter up-projection 101.7 -> 94.0 us, encode step 12.51 -> 12.23 ms (profiles/r03_resid_nt_ab.txt); at batch 32
// (20 MB, cache resident) the default policy is 1.6 % faster, hence the size test.
typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ f32x4 ld_resid(const float *ptr) {
    if constexpr (NT) return __builtin_nontemporal_load((const f32x4 *)ptr);
    else return *(const f32x4 *)ptr;
}
template <bool NT>
__device__ __forceinline__ void st_resid(float *ptr, f32x4 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, (f32x4 *)ptr);
    else *(f32x4 *)ptr = v;
}

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

// derivatives of the two activations (training step): d/dx x*sigmoid(1.702 x) = s (1 + 1.702 x (1 - s));
// d/dx 0.5 x (1 + erf(x / sqrt 2)) = Phi(x) + x phi(x), erf by the same A&S 7.1.26 polynomial as gelu_erf_f
__device__ __forceinline__ float dquick_gelu_f(float x) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930157f * x));
    return s * (1.0f + 1.702f * x * (1.0f - s));
}
__device__ __forceinline__ float dgelu_erf_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);  // exp(-x^2 / 2)
    const float cdf = 0.5f * (1.0f + copysignf(1.0f - poly * e, x));
    return cdf + x * 0.3989422804014327f * e;
}

template <int EPI>
struct traits {
    static constexpr bool dact = (EPI == EPI_BIAS_DACT_QUICK || EPI == EPI_BIAS_DACT_GELU);
    static constexpr bool act2 = (EPI == EPI_FOLD_ACT2_QUICK || EPI == EPI_FOLD_ACT2_GELU);
    static constexpr bool fold = (EPI == EPI_FOLD_BIAS || EPI == EPI_FOLD_QUICKGELU || EPI == EPI_FOLD_GELU || act2);
    static constexpr bool quick = (EPI == EPI_BIAS_QUICKGELU || EPI == EPI_FOLD_QUICKGELU);
    static constexpr bool erf = (EPI == EPI_BIAS_GELU || EPI == EPI_FOLD_GELU);
    static constexpr bool bf16_only = (EPI == EPI_BIAS || EPI == EPI_BIAS_STATS || quick || erf || fold || dact);
    static constexpr bool scale_resid = (EPI == EPI_SCALE_RESID || EPI == EPI_SCALE_RESID_STATS);
    static constexpr bool stats = (EPI == EPI_BIAS_STATS || EPI == EPI_SCALE_RESID_STATS);
};

template <int EPI>
__device__ __forceinline__ f32x4 activate(f32x4 v) {
    if constexpr (traits<EPI>::quick) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = quick_gelu_f(v[r]);
    }
    if constexpr (traits<EPI>::erf) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf_f(v[r]);
    }
    return v;
}

// sums over aligned groups of 8 / 16 lanes with DPP only (no LDS traffic): xor-1, xor-2 inside a quad, then the half-row /
// row mirrors (after the quad steps every lane of a quad holds the quad's sum, so a mirror is as good as an xor)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float sum8(float v) {
    v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);  // row_half_mirror
    return v;
}
__device__ __forceinline__ float sum16(float v) {
    v = sum8(v);
    v += dpp_f<0x140>(v);  // row_mirror
    return v;
}

// LayerNorm-fold consumers: two threads per row (blockDim = 2 * ROWS) turn the producer's partial sums of rows
// m0 .. m0 + ROWS - 1 into (mean, rstd) in LDS.  The loads are inline asm, issued BEFORE the block's first LDS-DMA and
// consumed after the prologue's counted vmcnt wait (vmcnt retires in issue order, so they are complete by then): written as
// ordinary loads the compiler would drain the whole DMA prologue with vmcnt(0) at their first use.
__device__ __forceinline__ void fold_stats_issue(const GemmParams &p, int m0, int tid, f32x4 (&v)[5]) {
    const int m = min(m0 + (tid >> 1), p.M - 1);
    const int n4 = p.K >> 7, H = (n4 + 1) >> 1;  // float4 = two 64-column slices; this thread reads float4 [half*H, half*H + H)
    const f32x4 *st = (const f32x4 *)(p.stats_in + (size_t)m * (p.K >> 6) * 2);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const f32x4 *src = st + min((tid & 1) * H + i, n4 - 1);  // clamped: every load is issued unconditionally
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[i]) : "v"(src) : "memory");
    }
}
__device__ __forceinline__ void fold_stats_finish(const GemmParams &p, int tid, const f32x4 (&v)[5], float *row_ms) {
    const int n4 = p.K >> 7, H = (n4 + 1) >> 1, first = (tid & 1) * H;
    float sm = 0.f, sq = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const bool valid = i < H && first + i < n4;
        sm += valid ? v[i][0] + v[i][2] : 0.f;
        sq += valid ? v[i][1] + v[i][3] : 0.f;
    }
    sm += dpp_f<0xB1>(sm);  // the row's other half (lane ^ 1)
    sq += dpp_f<0xB1>(sq);
    const float inv = 1.0f / (float)p.K;
    const float mean = sm * inv;
    const float var = fmaxf(sq * inv - mean * mean, 0.f);
    if (!(tid & 1)) *(ch_f32x2_t *)(row_ms + (tid & ~1)) = ch_f32x2_t{rsqrtf(var + p.ln_eps), mean};   // (rstd, mean): see store_tile
}

// Residual read-modify-write epilogues of a 64x64 wave tile (MT = 4): the fp32 residual values a lane will
// update, in the lane mapping of store_tile's read-back loop.  Loaded BEFORE the K loop (64 VGPRs, free in the 128x128
// kernel at two waves per SIMD) they arrive under the MFMA work, and the epilogue only stores.
struct ResidPrefetch {
    f32x4 hv[8];  // the first of the two read-back batches
    uint2 av[8];
};
template <int EPI, bool NT = false>
__device__ __forceinline__ void resid_prefetch(const GemmParams &p, int m_base, int n_base, int lane, ResidPrefetch &r) {
    const int lrow = lane >> 4, pos = lane & 15;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = j * 4 + lrow;
        const int m = m_base + row;
        const int mc = m < p.M ? m : p.M - 1;
        const int n = n_base + (pos ^ (row & 15)) * 4;
        r.hv[j] = ld_resid<NT>(p.resid + (size_t)mc * p.ldr + n);
        r.av[j] = make_uint2(0u, 0u);
        if (p.addend) r.av[j] = *(const uint2 *)(p.addend + (size_t)mc * p.ld_addend + n);
    }
}

// wave_lds: this wave's 16 KB staging region; m_base / n_base: global row / column of the wave's sub-tile origin;
// row_ms: (EPI_FOLD_*) the block tile's (mean, rstd) table in LDS, already offset to this wave's first row;
// pf: (PREF) the residual / addend values loaded by resid_prefetch.
template <int EPI, int MT, bool PREF = false, bool NT = false, bool NTOUT = false>
__device__ __forceinline__ void store_tile(const GemmParams &p, f32x4 (&acc)[4][MT], char *wave_lds, int m_base, int n_base,
                                           int lane, const float *row_ms = nullptr, const ResidPrefetch *pf = nullptr) {
    static_assert(!PREF || (MT == 4 && traits<EPI>::scale_resid), "prefetched residual: 64-row wave tiles, scale+residual epilogues");
    const int fr = lane & 15, fq = lane >> 4;
    using T = traits<EPI>;
    constexpr bool BF16_ONLY = T::bf16_only;
    f32x4 bias[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        if constexpr (EPI != EPI_PATCH)
            bias[nt] = *(const f32x4 *)(p.bias + n_base + nt * 16 + fq * 4);
        else
            bias[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    if constexpr (BF16_ONLY) {
        // (training, EPI_BIAS_DACT_*) the saved pre-activations this lane multiplies with, in the read-back loop's mapping: ALL
        // requested before the staging -- inside the loop each load sits behind the previous row group's store (they may alias
        // as far as the compiler knows), i.e. one exposed round trip per row group, 16 per 128-row wave tile
        uint4 ax[T::dact ? MT * 2 : 1];
        float dact_scale = 1.0f;
        if constexpr (T::dact) {
            const int lrow = lane >> 3, pos = lane & 7;
#pragma unroll
            for (int i = 0; i < MT * 2; ++i) {
                const int row = i * 8 + lrow;
                const int m = m_base + row;
                const int mc = m < p.M ? m : p.M - 1;
                ax[i] = *(const uint4 *)(p.aux + (size_t)mc * p.ldo + n_base + (pos ^ (row & 7)) * 8);
            }
            if (EPI == EPI_BIAS_DACT_GELU && p.scale_ptr) dact_scale = *p.scale_ptr;
        }
        // ---- stage MT*16 rows x 64 cols of bf16 (128-B rows)
        f32x4 fc[4];
        if constexpr (T::fold) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) fc[nt] = *(const f32x4 *)(p.fold_c + n_base + nt * 16 + fq * 4);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = mt * 16 + fr;
            float rs = 1.0f, nm = 0.f;
            if constexpr (T::fold) {
                // The table holds (rstd, mean), rstd FIRST: the scalar every accumulator is multiplied by then sits in the LOW half of
                // the loaded register pair and the compiler broadcasts it with op_sel_hi = 0.  With (mean, rstd) it selects the high
                // half with op_sel = 1, and `v_pk_fma_f32 D, acc, ms, C op_sel:[0,1,0]` (high half of SRC1 into the low lane) returns a
                // wrong low lane, about once per 1e7 executions, while a wave of ANOTHER kernel executes MFMAs on the same SIMD --
                // measured on MI355X, reproduced stand-alone (tools/pk_opsel_repro.py), not cured by wait states (DESIGN.md section
                // 3.10; tools/check_dpp_hazards.py rejects the spelling in every built kernel).
                const ch_f32x2_t ms = *(const ch_f32x2_t *)(row_ms + 2 * row);
                rs = ms[0];
                nm = -ms[1] * ms[0];
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                f32x4 pre;
                if constexpr (T::fold)
                    pre = acc[nt][mt] * rs + (fc[nt] * nm + bias[nt]);
                else
                    pre = acc[nt][mt] + bias[nt];
                const f32x4 v = activate<EPI>(pre);
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                // ds_write_b64 is served in groups of 16 consecutive lanes over 32 banks: rows r and r + 8 of a group share
                // (row & 7), so the 8-byte half inside the 16-byte chunk is flipped by row bit 3 (un-flipped at read-back);
                // without it every staging store is a 2-way conflict (SQ_LDS_BANK_CONFLICT: ~1000 cycles per 256x256 tile)
                *(uint2 *)(wave_lds + row * 128 + (((nt * 2 + (fq >> 1)) ^ (row & 7)) << 4) + (((fq & 1) ^ ((row >> 3) & 1)) << 3)) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int lrow = lane >> 3, pos = lane & 7;
#pragma unroll
        for (int i = 0; i < MT * 2; ++i) {
            const int row = i * 8 + lrow;
            const int chunk = pos ^ (row & 7);
            uint4 v = *(const uint4 *)(wave_lds + row * 128 + pos * 16);
            if (i & 1) v = make_uint4(v.z, v.w, v.x, v.y);  // rows with bit 3 set (row = 8 i + lrow) hold their 8-byte halves swapped
            const int m = m_base + row;
            if constexpr (T::dact) {  // training: times the activation's derivative at the saved pre-activation (and the adapter scale)
                const uint4 a = ax[i];
                const float sc = dact_scale;
                const uint32_t vw[4] = {v.x, v.y, v.z, v.w}, aw[4] = {a.x, a.y, a.z, a.w};
                uint32_t ow[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x0 = bf2f((bf16_t)(aw[j] & 0xffff)), x1 = bf2f((bf16_t)(aw[j] >> 16));
                    const float d0 = EPI == EPI_BIAS_DACT_QUICK ? dquick_gelu_f(x0) : dgelu_erf_f(x0);
                    const float d1 = EPI == EPI_BIAS_DACT_QUICK ? dquick_gelu_f(x1) : dgelu_erf_f(x1);
                    ow[j] = pack_bf16x2(sc * bf2f((bf16_t)(vw[j] & 0xffff)) * d0, sc * bf2f((bf16_t)(vw[j] >> 16)) * d1);
                }
                v = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            }
            if (m < p.M) {
                uint4 *dst = (uint4 *)(p.out_bf16 + (size_t)m * p.ldo + n_base + chunk * 8);
                if constexpr (NTOUT) __builtin_nontemporal_store(__builtin_bit_cast(u32x4_nt, v), (u32x4_nt *)dst);
                else *dst = v;
            }
            if constexpr (T::act2) {  // training: the activation of the (bf16-rounded) pre-activation as a second output
                const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
                uint32_t ow[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x0 = bf2f((bf16_t)(vw[j] & 0xffff)), x1 = bf2f((bf16_t)(vw[j] >> 16));
                    ow[j] = EPI == EPI_FOLD_ACT2_QUICK ? pack_bf16x2(quick_gelu_f(x0), quick_gelu_f(x1))
                                                       : pack_bf16x2(gelu_erf_f(x0), gelu_erf_f(x1));
                }
                if (m < p.M) {
                    uint4 *dst2 = (uint4 *)(p.hb_out + (size_t)m * p.ld_hb + n_base + chunk * 8);
                    const uint4 w2 = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                    if constexpr (NTOUT) __builtin_nontemporal_store(__builtin_bit_cast(u32x4_nt, w2), (u32x4_nt *)dst2);
                    else *dst2 = w2;
                }
            }
            if constexpr (T::stats) {  // partial (sum, sumsq) of this row's 64 rounded outputs: 8 lanes x 8 values
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                float sm = 0.f, sq = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = bf2f((bf16_t)(w[j] & 0xffff)), b = bf2f((bf16_t)(w[j] >> 16));
                    sm += a + b;
                    sq += a * a + b * b;
                }
                sm = sum8(sm);
                sq = sum8(sq);
                if (pos == 0 && m < p.M)
                    *(ch_f32x2_t *)(p.stats_out + ((size_t)m * (p.N >> 6) + (n_base >> 6)) * 2) = ch_f32x2_t{sm, sq};
            }
        }
    } else {
        // ---- fp32 staging, 64 rows (4 m-tiles) per pass: 256-B rows
        float scale = 1.0f;
        if constexpr (T::scale_resid) scale = *p.scale_ptr;
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
                        if (PREF && batch == 0)
                            hv[i] = pf->hv[i];
                        else
                            hv[i] = EPI == EPI_BIAS_RESID ? *(const f32x4 *)(p.resid + off[i]) : ld_resid<NT>(p.resid + off[i]);
                        if constexpr (T::scale_resid) {
                            if (PREF && batch == 0) {
                                av[i] = pf->av[i];
                            } else {
                                av[i] = make_uint2(0u, 0u);
                                if (p.addend) av[i] = *(const uint2 *)(p.addend + (size_t)mc * p.ld_addend + n);
                            }
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
                        st_resid<NT>(p.resid + off[i], h);
                        if constexpr (T::stats) {  // bf16 copy for the next (LN-folded) GEMM + its row statistics
                            const int row = (batch * 8 + i) * 4 + lrow;
                            const int m = m_base + pass * 64 + row;
                            const int n = n_base + (pos ^ (row & 15)) * 4;
                            uint2 o;
                            o.x = pack_bf16x2(h[0], h[1]);
                            o.y = pack_bf16x2(h[2], h[3]);
                            *(uint2 *)(p.hb_out + (size_t)m * p.ld_hb + n) = o;
                            const float a = bf2f((bf16_t)(o.x & 0xffff)), b = bf2f((bf16_t)(o.x >> 16));
                            const float c = bf2f((bf16_t)(o.y & 0xffff)), d = bf2f((bf16_t)(o.y >> 16));
                            const float sm = sum16((a + b) + (c + d));
                            const float sq = sum16((a * a + b * b) + (c * c + d * d));
                            if (pos == 0)
                                *(ch_f32x2_t *)(p.stats_out + ((size_t)m * (p.N >> 6) + (n_base >> 6)) * 2) = ch_f32x2_t{sm, sq};
                        }
                    }
                }
            }
        }
    }
}

}  // namespace ch_epi
