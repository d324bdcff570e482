// Backward of the fused multi-head attention (head_dim 64, no mask; HF CLIPAttention as used by
// models/layers/adapter.py:127-177) for the ConceptHash training step -- one workgroup per (image, head), nothing but
// qkv and dO read, nothing but dqkv written; the probabilities are recomputed, never stored.
//
//   P = softmax(0.125 Q K^T),  O = P V                                   (forward, attention.hip)
//   dV = P^T dO,  dP = dO V^T,  D_q = sum_j P_qj dP_qj (= dO_q . O_q),  dS = 0.125 P o (dP - D),  dQ = dS K,  dK = dS^T Q
//
// LDS holds two [rows][128 B] images (row-major, 16-B chunk index XOR (row & 7), filled by LDS-DMA): K and V during phase A, then --
// restaged between the phases -- Q and dO during phase B; the operands a phase needs only as per-lane fragments (Q / dO rows of
// the wave's query tile in A, K / V rows of its key tile in B) come straight from global memory, as the forward kernel reads Q.
// 59 KB instead of 118 KB, so TWO workgroups share a CU and cover each other's latencies (one was alone before).
// Phase A -- a wave owns 16-query tiles (the forward's layout: S^T = K Q^T, a lane holds one query's keys):
//   softmax statistics, dP^T = V dO^T, D_q, dS^T in registers; dQ^T = K^T dS^T with K^T from the hardware transpose read
//   (ds_read_b64_tr_b16), dS^T being already the B operand; (max, 1/sum, D_q) per query go to LDS.
// Phase B -- a wave owns 16-key tiles and walks the query tiles: S = Q K^T and dP = dO V^T in the TRANSPOSED lane layout
//   (a lane holds one key's queries), P and dS rebuilt from the saved statistics; dV^T += dO^T P and dK^T += Q^T dS by
//   v_mfma_f32_16x16x16_bf16, whose reduction index is the 16 queries of the tile: P / dS in the accumulator layout ARE its B
//   operand, dO^T / Q^T come through the transpose read.  The second S / dP product costs 4 MFMAs per tile pair and saves a
//   round trip of P and dS through LDS.
#include "ch_common.h"
#include "kernels.h"

namespace {

constexpr int HD = 64;
constexpr int NW = 8;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s lds_v4s;

// EXT: an extra cotangent on the probabilities themselves -- dpext [B, heads, ncon, ntok - ncon - 1] fp32, the gradient of the
// concept-token attention rows the forward can tap (attention.hip TAP; consumer: the attention-diversity term of the loss,
// models/loss/coop.py:164-189) -- is added to dP = dO V^T on those (query, key) pairs, in both phases.
template <int KB, bool EXT>
__global__ __launch_bounds__(NW * 64, 4) void attention_bwd_kernel(const bf16_t *__restrict__ qkv, const bf16_t *__restrict__ dO, int ntok,
                                                               int heads, float scale_log2e, bf16_t *__restrict__ dqkv,
                                                               const float *__restrict__ dpext, int ncon) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = KB * 2;   // 16-row tiles
    constexpr int KP = KB * 32;  // padded rows (keys and queries)
    char *Ks = smem, *Vs = smem + KP * 128;    // phase A: K, V
    char *Qs = smem, *Gs = smem + KP * 128;    // phase B: Q, dO (same storage, restaged after phase A)
    float *stat = (float *)(smem + 2 * KP * 128);   // [KP][4]: lse (log2 domain), D_q / 8, -, -

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
    const int D = heads * HD;
    const size_t ld = (size_t)3 * D;
    const bf16_t *base = qkv + (size_t)b * ntok * ld + h * HD;
    const bf16_t *gbase = dO + (size_t)b * ntok * D + h * HD;
    bf16_t *obase = dqkv + (size_t)b * ntok * ld + h * HD;

    {  // stage: instruction i covers rows 8i..8i+7; rows past the sequence re-read the last row (finite; masked / zero weight)
        const int lrow = lane >> 3;
        const int src_chunk = (lane & 7) ^ lrow;
        for (int i = wid; i < KP / 8; i += NW) {
            int row = i * 8 + lrow;
            row = row < ntok ? row : ntok - 1;
            const bf16_t *src = base + (size_t)row * ld + src_chunk * 8;
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + D), (lds_void_t *)(Ks + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + 2 * D), (lds_void_t *)(Vs + i * 1024), 16, 0, 0);
        }
    }

    const int fr = lane & 15, fq = lane >> 4;
    const int QT = (ntok + 15) >> 4;  // tiles holding at least one valid row
    const int npatch = ntok - ncon - 1, q_con0 = ntok - ncon;   // EXT: keys 1 .. npatch, queries q_con0 .. ntok - 1
    const float *ext = EXT ? dpext + (size_t)blockIdx.x * ncon * npatch : nullptr;
    // b128 fragment of row (tile*16 + fr): d = 8*fq .. +8 (off0) and 32 + 8*fq .. (off1)
    const int off0 = fr * 128 + ((fq ^ (fr & 7)) << 4), off1 = fr * 128 + (((4 + fq) ^ (fr & 7)) << 4);
    // transpose read: lane 4*tq + tp of group fq addresses row 4*fq + tq of a 16-row tile, features 4*tp.. of feature tile dt
    const int tq = fr >> 2, tp = fr & 3;
    const int trow7 = ((fq & 1) << 2) | tq;
    int toff[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) toff[dt] = (fq * 4 + tq) * 128 + (((dt * 2 + (tp >> 1)) ^ trow7) << 4) + (tp & 1) * 8;

    // fragments of one 16-row tile straight from global memory: lane (row fr, fq) holds columns 8*fq.. and 32 + 8*fq.. of its row
    // (rows past the sequence re-read the last row: finite, masked / zero weight)
    auto row_frag = [&](const bf16_t *mat, size_t ldm, int tile, bf16x8 &f0, bf16x8 &f1) {
        const int r = min(tile * 16 + fr, ntok - 1);
        f0 = *(const bf16x8 *)(mat + (size_t)r * ldm + fq * 8);
        f1 = *(const bf16x8 *)(mat + (size_t)r * ldm + fq * 8 + 32);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ================================ phase A: per query tile -> statistics and dQ ==========================================
    for (int qt = wid; qt < QT; qt += NW) {
        const int q = qt * 16 + fr;
        const bool qvalid = q < ntok;
        bf16x8 qf0, qf1, gf0, gf1;      // not prefetched a tile ahead: 16 VGPRs that the four-waves-per-SIMD budget does not have
        row_frag(base, ld, qt, qf0, qf1);
        row_frag(gbase, D, qt, gf0, gf1);
        // Register budget: only S^T / P^T of the tile (KT x 4 values) stays live; dP^T = V dO^T is RECOMPUTED where it is needed (once
        // for D_q, once for dS) instead of kept -- 28 more MFMAs per query tile on a pipe that is a quarter busy, 56 fewer VGPRs, so
        // that four waves per SIMD fit (the kernel is bound by its dependent LDS -> MFMA -> exp -> MFMA chains, not by throughput).
        f32x4 st[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const bf16x8 k0 = *(const bf16x8 *)(Ks + kt * 2048 + off0), k1 = *(const bf16x8 *)(Ks + kt * 2048 + off1);
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf0, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf1, a, 0, 0, 0);
            st[kt] = a;   // S^T[key = kt*16 + 4*fq + r][query fr]
        }
        const bool ext_row = EXT && qt * 16 + 15 >= q_con0 && q >= q_con0 && qvalid;
        const float *erow = ext_row ? ext + (size_t)(q - q_con0) * npatch : nullptr;
        auto dp_tile = [&](int kt) {   // dP^T[key = kt*16 + 4*fq + r][query fr] (+ the cotangent on the probabilities, EXT)
            const bf16x8 v0 = *(const bf16x8 *)(Vs + kt * 2048 + off0), v1 = *(const bf16x8 *)(Vs + kt * 2048 + off1);
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0, gf0, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, gf1, c, 0, 0, 0);
            if constexpr (EXT) {
                if (ext_row) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt * 16 + fq * 4 + r;
                        if (key >= 1 && key <= npatch) c[r] += erow[key - 1];
                    }
                }
            }
            return c;
        };
        // keys >= ntok exist only in the last (KP - ntok + 15) / 16 <= 2 key tiles: mask those under a wave-uniform test (the PMC
        // counters show this kernel bound by VALU issue -- ~2,300 non-MFMA VALU instructions per wave against ~440 MFMAs)
#pragma unroll
        for (int kt = KT - 2; kt < KT; ++kt)
            if (kt >= 0 && kt * 16 + 16 > ntok) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kt * 16 + fq * 4 + r >= ntok) st[kt][r] = -1e30f;
            }
        float mx = -1e30f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mxs = mx * scale_log2e;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(st[kt][r] * scale_log2e - mxs);
                st[kt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        float pd = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const f32x4 c = dp_tile(kt);
#pragma unroll
            for (int r = 0; r < 4; ++r) pd += st[kt][r] * c[r];
        }
        pd += __shfl_xor(pd, 16, 64);
        pd += __shfl_xor(pd, 32, 64);
        const float inv = 1.0f / sum;
        const float Dq = pd * inv;
        // per query for phase B: lse = mxs + log2(sum) so that P = exp2(s * scale_log2e - lse) needs no multiply by 1/sum (an invalid
        // query gets lse = 1e30 -> P = 0), and D_q / 8 so that dS = P * fma(dP, 0.125, -D_q/8): two VALU ops fewer per probability
        if (fq == 0) *(f32x4 *)(stat + q * 4) = f32x4{qvalid ? mxs + __builtin_amdgcn_logf(sum) : 1e30f, 0.125f * Dq, 0.f, 0.f};
        // dS^T = 0.125 * P^T o (dP^T - D_q), as the bf16 B operand; logical k = 8*fq + j <-> key 32*kb + (j < 4 ? 4*fq + j : 16 + 4*fq + j - 4)
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float w = 0.125f * inv;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            union {
                bf16x8 v;
                uint32_t u[4];
            } pf;
            const f32x4 c0 = dp_tile(2 * kb), c1 = dp_tile(2 * kb + 1);
            pf.u[0] = pack_bf16x2(w * st[2 * kb][0] * (c0[0] - Dq), w * st[2 * kb][1] * (c0[1] - Dq));
            pf.u[1] = pack_bf16x2(w * st[2 * kb][2] * (c0[2] - Dq), w * st[2 * kb][3] * (c0[3] - Dq));
            pf.u[2] = pack_bf16x2(w * st[2 * kb + 1][0] * (c1[0] - Dq), w * st[2 * kb + 1][1] * (c1[1] - Dq));
            pf.u[3] = pack_bf16x2(w * st[2 * kb + 1][2] * (c1[2] - Dq), w * st[2 * kb + 1][3] * (c1[3] - Dq));
            union {
                bf16x8 v;
                v4s hh[2];
            } kf[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                kf[dt].hh[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(Ks + kb * 4096 + toff[dt]));
                kf[dt].hh[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(Ks + kb * 4096 + 2048 + toff[dt]));
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[dt].v, pf.v, o[dt], 0, 0, 0);
        }
        if (qvalid) {
            bf16_t *op = obase + (size_t)q * ld + fq * 4;  // dQ[q][dt*16 + 4*fq + r]
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 wv;
                wv.x = pack_bf16x2(o[dt][0], o[dt][1]);
                wv.y = pack_bf16x2(o[dt][2], o[dt][3]);
                *(uint2 *)(op + dt * 16) = wv;
            }
        }
    }
    __syncthreads();   // every wave is done with K and V in LDS, the statistics are complete
    {  // restage: Q and dO take the place of K and V
        const int lrow = lane >> 3;
        const int src_chunk = (lane & 7) ^ lrow;
        for (int i = wid; i < KP / 8; i += NW) {
            int row = i * 8 + lrow;
            row = row < ntok ? row : ntok - 1;
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(base + (size_t)row * ld + src_chunk * 8), (lds_void_t *)(Qs + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(gbase + (size_t)row * D + src_chunk * 8), (lds_void_t *)(Gs + i * 1024), 16, 0, 0);
        }
    }
    bf16x8 k0, k1, v0, v1;   // this wave's first key tile, from global, under the restaging
    if (wid < QT) {
        row_frag(base + D, ld, wid, k0, k1);
        row_frag(base + 2 * D, ld, wid, v0, v1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ================================ phase B: per key tile -> dK, dV ========================================================
    for (int kt = wid; kt < QT; kt += NW) {
        const int key = kt * 16 + fr;
        const bool kvalid = key < ntok;
        const bool edge_tile = kt * 16 + 16 > ntok;
        if (kt != wid) {
            row_frag(base + D, ld, kt, k0, k1);
            row_frag(base + 2 * D, ld, kt, v0, v1);
        }
        f32x4 dkt[4], dvt[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dkt[dt] = dvt[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int qt = 0; qt < QT; ++qt) {
            const bf16x8 qf0 = *(const bf16x8 *)(Qs + qt * 2048 + off0), qf1 = *(const bf16x8 *)(Qs + qt * 2048 + off1);
            const bf16x8 gf0 = *(const bf16x8 *)(Gs + qt * 2048 + off0), gf1 = *(const bf16x8 *)(Gs + qt * 2048 + off1);
            f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, d4 = {0.f, 0.f, 0.f, 0.f};
            s4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf0, k0, s4, 0, 0, 0);  // S[q = qt*16 + 4*fq + r][key fr]
            s4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf1, k1, s4, 0, 0, 0);
            d4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf0, v0, d4, 0, 0, 0);  // dP, same layout
            d4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf1, v1, d4, 0, 0, 0);
            if constexpr (EXT) {
                if (qt * 16 + 15 >= q_con0 && key >= 1 && key <= npatch) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int qq = qt * 16 + fq * 4 + r;
                        if (qq >= q_con0 && qq < ntok) d4[r] += ext[(size_t)(qq - q_con0) * npatch + key - 1];
                    }
                }
            }
            float p[4], ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x4 sv = *(const f32x4 *)(stat + (qt * 16 + fq * 4 + r) * 4);  // (lse, D_q / 8)
                float pr = __builtin_amdgcn_exp2f(s4[r] * scale_log2e - sv[0]);
                if (edge_tile && !kvalid) pr = 0.f;   // only the tile that straddles the sequence end has invalid keys (wave-uniform test)
                p[r] = pr;
                ds[r] = pr * (d4[r] * 0.125f - sv[1]);
            }
            union {
                v4s v;
                uint32_t u[2];
            } pb, sb;
            pb.u[0] = pack_bf16x2(p[0], p[1]);
            pb.u[1] = pack_bf16x2(p[2], p[3]);
            sb.u[0] = pack_bf16x2(ds[0], ds[1]);
            sb.u[1] = pack_bf16x2(ds[2], ds[3]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const v4s gt = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(Gs + qt * 2048 + toff[dt]));  // dO^T[d][q]
                const v4s qT = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(Qs + qt * 2048 + toff[dt]));  // Q^T[d][q]
                dvt[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(gt, pb.v, dvt[dt], 0, 0, 0);
                dkt[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qT, sb.v, dkt[dt], 0, 0, 0);
            }
        }
        if (kvalid) {
            bf16_t *kp = obase + (size_t)key * ld + D + fq * 4, *vp = kp + D;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 wk, wv;
                wk.x = pack_bf16x2(dkt[dt][0], dkt[dt][1]);
                wk.y = pack_bf16x2(dkt[dt][2], dkt[dt][3]);
                wv.x = pack_bf16x2(dvt[dt][0], dvt[dt][1]);
                wv.y = pack_bf16x2(dvt[dt][2], dvt[dt][3]);
                *(uint2 *)(kp + dt * 16) = wk;
                *(uint2 *)(vp + dt * 16) = wv;
            }
        }
    }
}

template <int KB, bool EXT>
int launch_bwd_inst(const bf16_t *qkv, const bf16_t *dO, int B, int ntok, int heads, bf16_t *dqkv, const float *dpext, int ncon, hipStream_t s) {
    constexpr int KP = KB * 32;
    const size_t lds = (size_t)KP * 128 * 2 + (size_t)KP * 16;
    CH_REQUIRE(lds <= 160 * 1024, "attention backward: sequence too long for the LDS-resident kernel");
    static ch_once_per_device lds_once;
    if (int e = ch_func_max_lds((const void *)attention_bwd_kernel<KB, EXT>, (int)lds, lds_once)) return e;
    const float scale_log2e = 0.125f * 1.4426950408889634f;
    hipLaunchKernelGGL((attention_bwd_kernel<KB, EXT>), dim3(B * heads), dim3(NW * 64), lds, s, qkv, dO, ntok, heads, scale_log2e, dqkv, dpext,
                       ncon);
    CH_LAUNCH_CHECK();
    return 0;
}
template <int KB>
int launch_bwd(const bf16_t *qkv, const bf16_t *dO, int B, int ntok, int heads, bf16_t *dqkv, const float *dpext, int ncon, hipStream_t s) {
    return dpext ? launch_bwd_inst<KB, true>(qkv, dO, B, ntok, heads, dqkv, dpext, ncon, s)
                 : launch_bwd_inst<KB, false>(qkv, dO, B, ntok, heads, dqkv, nullptr, 0, s);
}

}  // namespace

// qkv [B*ntok, 3D] bf16 (q | k | v), dO [B*ntok, D] bf16 -> dqkv [B*ntok, 3D] bf16 (dq | dk | dv); head_dim 64
// dpext (optional): [B, heads, ncon, ntok - ncon - 1] fp32 cotangent of the concept tokens' attention rows over the patch tokens
int ch_attention_bwd(const bf16_t *qkv, const bf16_t *dO, int B, int ntok, int heads, bf16_t *dqkv, hipStream_t s, const float *dpext,
                     int ncon) {
    CH_REQUIRE(B > 0 && ntok > 0 && heads > 0, "attention backward: empty problem");
    CH_REQUIRE(!dpext || (ncon >= 1 && ncon < ntok - 1), "attention backward: the probability cotangent needs 1 <= ncon < ntok - 1");
    const int KB = (ntok + 31) / 32;
    switch (KB) {
        case 1: return launch_bwd<1>(qkv, dO, B, ntok, heads, dqkv, dpext, ncon, s);
        case 2: return launch_bwd<2>(qkv, dO, B, ntok, heads, dqkv, dpext, ncon, s);
        case 3: return launch_bwd<3>(qkv, dO, B, ntok, heads, dqkv, dpext, ncon, s);
        case 4: return launch_bwd<4>(qkv, dO, B, ntok, heads, dqkv, dpext, ncon, s);
        case 5: return launch_bwd<5>(qkv, dO, B, ntok, heads, dqkv, dpext, ncon, s);
        case 6: return launch_bwd<6>(qkv, dO, B, ntok, heads, dqkv, dpext, ncon, s);
        case 7: return launch_bwd<7>(qkv, dO, B, ntok, heads, dqkv, dpext, ncon, s);
        case 8: return launch_bwd<8>(qkv, dO, B, ntok, heads, dqkv, dpext, ncon, s);
        case 9: return launch_bwd<9>(qkv, dO, B, ntok, heads, dqkv, dpext, ncon, s);
    }
    ch_set_error("attention backward: more than 288 tokens per image is not built (LDS-resident Q/K/V/dO)");
    return 2;
}
