// Multi-head self-attention for the ConceptHash ViT encoder (gfx950): head_dim 64, sequence 54..288 tokens, no mask.
//
// Restates HF CLIPAttention's eager path (transformers modeling_clip.py eager_attention_forward: softmax(q k^T / sqrt(d)) v)
// as called from the reference's CLIPEncoderLayerWithAdapter.forward (models/layers/adapter.py:146-152).  The reference
// materialises every layer's (B,heads,N,N) probability map (output_attentions=True, models/arch/coop.py:474-479); retrieval
// never reads it, so this kernel keeps scores in registers and never writes them.
//
// One workgroup (8 waves) per (image, head).  The head's whole K and V are staged ROW-MAJOR into LDS by LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip, all 14 requests of a wave in flight at once); rows are 128 B with the
// 16-B chunk index XOR (row & 7) applied on the per-lane source address.  Each wave takes 16-query tiles round robin:
//   S^T = K Q^T      v_mfma_f32_16x16x32_bf16 with A = K tile (ds_read_b128, conflict free), B = Q^T fragment straight
//                    from global: a lane holds, for ONE query (lane & 15), 4 consecutive keys per 16-key tile, so the
//                    softmax reductions are in-register plus two cross-lane steps (xor 16, xor 32);
//   O^T = V^T P^T    the exponentiated scores, converted to bf16 in place, already ARE the B operand of this product; the
//                    A operand V^T comes from the row-major V image through the hardware transpose read
//                    ds_read_b64_tr_b16 (a 16-lane group fetches 4 keys x 16 features and each lane receives one feature
//                    column) -- conflict free with the same swizzle.  P never goes through LDS.
//   out = O / rowsum (fp32), 4 consecutive d per lane -> 8-byte bf16 stores.
#include "ch_common.h"
#include "kernels.h"

namespace {

constexpr int HD = 64;
// cache policy of the K / V / Q reads (each element is read exactly once): aux 2 (nt) measured, no gain
// (profiles/r03_cache_policy_ab.txt) -> default policy
constexpr int QKV_AUX = 0;
#define LDQ(p) (*(const bf16x8 *)(p))
constexpr int NW = 8;  // waves per workgroup: 13 query tiles (201 tokens) take two rounds instead of four with 4 waves; K/V staged once
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s lds_v4s;

// KB: number of 32-key blocks (keys padded to KB*32); TAP: write the concept-token attention rows; COMPACT: only the rows the
// hashing head reads -- CLS and the `ncon` concept tokens -- are queries, and the output is [B * (1 + ncon), D] (final layer)
template <int KB, bool TAP, bool COMPACT>
__global__ __launch_bounds__(NW * 64, 4) void attention_kernel(const bf16_t *__restrict__ qkv, int ntok, int heads, float scale_log2e,
                                                        bf16_t *__restrict__ out, float *__restrict__ cattn, int ncon, int rev) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = KB * 2;   // 16-key tiles
    constexpr int KP = KB * 32;  // padded keys
    char *Ks = smem;             // [KP][128 B]
    char *Vs = smem + KP * 128;  // [KP][128 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int b = bid / heads, h = bid - b * heads;
    const int D = heads * HD;
    const size_t ld = (size_t)3 * D;
    const bf16_t *base = qkv + (size_t)b * ntok * ld + h * HD;

    // ---- stage K and V: instruction i covers rows 8i..8i+7 (1 KB); rows past the sequence re-read the last row (finite
    // data; those keys are masked to -inf and get probability 0)
    {
        const int lrow = lane >> 3;
        const int src_chunk = (lane & 7) ^ lrow;
        for (int i = wid; i < KP / 8; i += NW) {
            int row = i * 8 + lrow;
            row = row < ntok ? row : ntok - 1;
            const bf16_t *src = base + (size_t)row * ld + src_chunk * 8;
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + D), (lds_void_t *)(Ks + i * 1024), 16, 0, QKV_AUX);
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + 2 * D), (lds_void_t *)(Vs + i * 1024), 16, 0, QKV_AUX);
        }
    }
    const int fr = lane & 15, fq = lane >> 4;
    const int nqc = 1 + ncon;  // COMPACT: queries per image
    const int QT = COMPACT ? (nqc + 15) >> 4 : (ntok + 15) >> 4;
    // query slot j of this image -> (token row, valid); COMPACT: slot 0 = CLS, slots 1.. = the concept tokens (the last ncon rows)
    auto query_of = [&](int j, bool &valid) {
        if constexpr (COMPACT) {
            valid = j < nqc;
            return !valid ? ntok - 1 : (j == 0 ? 0 : ntok - ncon + j - 1);
        } else {
            valid = j < ntok;
            return valid ? j : ntok - 1;
        }
    };
    // first query tile's Q fragments overlap the staging latency
    bool qvalid;
    int qslot = wid * 16 + fr;
    int q = query_of(qslot, qvalid);
    bf16x8 qf0 = LDQ(base + (size_t)q * ld + fq * 8);
    bf16x8 qf1 = LDQ(base + (size_t)q * ld + fq * 8 + 32);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // transpose-read addressing: lane i = 4*qq + pp of a 16-lane group supplies row (key0 + qq), features 4pp..4pp+3 of the
    // 16-feature tile dt: chunk = dt*2 + (pp >> 1), half = pp & 1
    const int tq = fr >> 2, tp = fr & 3;
    // (row & 7) of the rows a lane addresses is lane-constant: rows are kb*32 + fq*4 + tq (+16)
    const int trow7 = ((fq & 1) << 2) | tq;
    int voff[4];  // byte offset of this lane's 8-byte piece inside the 32-key block, per 16-feature tile dt
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
        voff[dt] = (fq * 4 + tq) * 128 + (((dt * 2 + (tp >> 1)) ^ trow7) << 4) + (tp & 1) * 8;
    const int koff0 = fr * 128 + ((fq ^ (fr & 7)) << 4), koff1 = fr * 128 + (((4 + fq) ^ (fr & 7)) << 4);

    for (int qt = wid; qt < QT; qt += NW) {
        // S^T in chunks of two key tiles, fragment reads of chunk c+1 issued before the MFMAs of chunk c; the scheduling fences
        // keep the compiler from hoisting all 28 reads (112 VGPRs) above the first MFMA, which costs the third wave per SIMD
        f32x4 st[KT];
        bf16x8 kf[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            kf[0][2 * t] = *(const bf16x8 *)(Ks + t * 2048 + koff0);
            kf[0][2 * t + 1] = *(const bf16x8 *)(Ks + t * 2048 + koff1);
        }
#pragma unroll
        for (int c = 0; c < KB; ++c) {
            if (c + 1 < KB) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    kf[(c + 1) & 1][2 * t] = *(const bf16x8 *)(Ks + (2 * c + 2 + t) * 2048 + koff0);
                    kf[(c + 1) & 1][2 * t + 1] = *(const bf16x8 *)(Ks + (2 * c + 2 + t) * 2048 + koff1);
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[c & 1][2 * t], qf0, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[c & 1][2 * t + 1], qf1, a, 0, 0, 0);
                st[2 * c + t] = a;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const bool cur_valid = qvalid;
        const int cur_q = q, cur_slot = qslot;
        // prefetch the next tile's Q fragments
        if (qt + NW < QT) {
            qslot = (qt + NW) * 16 + fr;
            q = query_of(qslot, qvalid);
            qf0 = LDQ(base + (size_t)q * ld + fq * 8);
            qf1 = LDQ(base + (size_t)q * ld + fq * 8 + 32);
        }
        // ---- softmax over keys for query (lane & 15): lane holds keys kt*16 + 4*fq + r
        // keys >= ntok exist only in the last (KP - ntok + 15) / 16 <= 2 key tiles: mask those under a wave-uniform test
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
        const float inv = 1.0f / sum;
        // optional interpretability tap (last layer only): softmax rows of the `ncon` concept tokens over the patch tokens,
        // = attn_cache[-1][:, :, -Q:, 1:-Q] of the reference (models/arch/coop.py:481-482, models/loss/coop.py:164-176)
        if (TAP && cur_valid && cur_q >= ntok - ncon) {
            const int np = ntok - ncon - 1;
            float *dst = cattn + (((size_t)b * heads + h) * ncon + (cur_q - (ntok - ncon))) * np;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 16 + fq * 4 + r;
                    if (key >= 1 && key <= np) dst[key - 1] = st[kt][r] * inv;
                }
        }

        // ---- O^T = V^T P^T ; logical k = 8*fq + j  <->  key 32*kb + (j < 4 ? 4*fq + j : 16 + 4*fq + j - 4)
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        union VF {
            bf16x8 v;
            v4s h[2];
        };
        VF vf[2][4];
        // keys 32kb + 4fq + 0..3 (elements 0..3) and + 16 (elements 4..7): immediate offsets off one lane base
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            vf[0][dt].h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(Vs + voff[dt]));
            vf[0][dt].h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(Vs + 2048 + voff[dt]));
        }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            if (kb + 1 < KB) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    vf[(kb + 1) & 1][dt].h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(Vs + (kb + 1) * 4096 + voff[dt]));
                    vf[(kb + 1) & 1][dt].h[1] =
                        __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(Vs + (kb + 1) * 4096 + 2048 + voff[dt]));
                }
            }
            union {
                bf16x8 v;
                uint32_t u[4];
            } pf;
            pf.u[0] = pack_bf16x2(st[2 * kb][0], st[2 * kb][1]);
            pf.u[1] = pack_bf16x2(st[2 * kb][2], st[2 * kb][3]);
            pf.u[2] = pack_bf16x2(st[2 * kb + 1][0], st[2 * kb + 1][1]);
            pf.u[3] = pack_bf16x2(st[2 * kb + 1][2], st[2 * kb + 1][3]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[kb & 1][dt].v, pf.v, o[dt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (cur_valid) {
            bf16_t *op = out + (COMPACT ? (size_t)b * nqc + cur_slot : (size_t)b * ntok + cur_q) * D + h * HD + fq * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 w;
                w.x = pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv);
                w.y = pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv);
                *(uint2 *)(op + dt * 16) = w;
            }
        }
    }
}

template <int KB, bool TAP, bool COMPACT>
int launch_attn_inst(const bf16_t *qkv, int B, int ntok, int heads, bf16_t *out, float *cattn, int ncon, int rev, hipStream_t s) {
    const int KP = KB * 32;
    const size_t lds = (size_t)KP * 128 * 2;
    CH_REQUIRE(lds <= 160 * 1024, "attention: sequence too long for the LDS-resident K/V kernel");
    static ch_once_per_device lds_once;
    if (int e = ch_func_max_lds((const void *)attention_kernel<KB, TAP, COMPACT>, (int)lds, lds_once)) return e;
    const float scale_log2e = 0.125f * 1.4426950408889634f;  // head_dim^-0.5 * log2(e), head_dim = 64
    CH_LAUNCH((attention_kernel<KB, TAP, COMPACT>), dim3(B * heads), dim3(NW * 64), lds, s, qkv, ntok, heads, scale_log2e,
                       out, cattn, ncon, rev);
    CH_LAUNCH_CHECK();
    return 0;
}
template <int KB>
int launch_attn(const bf16_t *qkv, int B, int ntok, int heads, bf16_t *out, float *cattn, int ncon, bool compact, int rev, hipStream_t s) {
    if (compact)
        return cattn ? launch_attn_inst<KB, true, true>(qkv, B, ntok, heads, out, cattn, ncon, rev, s)
                     : launch_attn_inst<KB, false, true>(qkv, B, ntok, heads, out, cattn, ncon, rev, s);
    return cattn ? launch_attn_inst<KB, true, false>(qkv, B, ntok, heads, out, cattn, ncon, rev, s)
                 : launch_attn_inst<KB, false, false>(qkv, B, ntok, heads, out, cattn, ncon, rev, s);
}

}  // namespace

int ch_attention(const bf16_t *qkv, int B, int ntok, int heads, bf16_t *out, hipStream_t s, float *cattn, int ncon, bool compact,
                 bool rev) {
    CH_REQUIRE(B > 0 && ntok > 0 && heads > 0, "attention: empty problem");
    CH_REQUIRE(!compact || (ncon >= 1 && ncon < ntok), "attention: compact mode needs 1 <= ncon < ntok");
    const int KB = (ntok + 31) / 32;
    switch (KB) {
        case 1: return launch_attn<1>(qkv, B, ntok, heads, out, cattn, ncon, compact, rev ? 1 : 0, s);
        case 2: return launch_attn<2>(qkv, B, ntok, heads, out, cattn, ncon, compact, rev ? 1 : 0, s);
        case 3: return launch_attn<3>(qkv, B, ntok, heads, out, cattn, ncon, compact, rev ? 1 : 0, s);
        case 4: return launch_attn<4>(qkv, B, ntok, heads, out, cattn, ncon, compact, rev ? 1 : 0, s);
        case 5: return launch_attn<5>(qkv, B, ntok, heads, out, cattn, ncon, compact, rev ? 1 : 0, s);
        case 6: return launch_attn<6>(qkv, B, ntok, heads, out, cattn, ncon, compact, rev ? 1 : 0, s);
        case 7: return launch_attn<7>(qkv, B, ntok, heads, out, cattn, ncon, compact, rev ? 1 : 0, s);
        case 8: return launch_attn<8>(qkv, B, ntok, heads, out, cattn, ncon, compact, rev ? 1 : 0, s);
        case 9: return launch_attn<9>(qkv, B, ntok, heads, out, cattn, ncon, compact, rev ? 1 : 0, s);
    }
    ch_set_error("attention: more than 288 tokens per image is not supported");
    return 2;
}
