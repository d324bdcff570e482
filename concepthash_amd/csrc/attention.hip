// Multi-head self-attention for the ConceptHash ViT encoder (gfx950): head_dim 64, sequence 54..288 tokens, no mask.
//
// Restates HF CLIPAttention's eager path (transformers modeling_clip.py eager_attention_forward: softmax(q k^T / sqrt(d)) v)
// as called from the reference's CLIPEncoderLayerWithAdapter.forward (models/layers/adapter.py:146-152).  The reference
// materialises every layer's (B,heads,N,N) probability map (output_attentions=True, models/arch/coop.py:474-479); retrieval
// never reads it, so this kernel keeps scores in registers and never writes them.
//
// One workgroup (4 waves) per (image, head).  The head's whole K (row-major, XOR-swizzled 128-B rows) and V (transposed,
// Vt[d][key], row stride == 16 mod 256 bytes so the 8-byte fragment reads are bank-conflict free) live in LDS.
// Each wave takes 16-query tiles round robin:
//   S^T = K Q^T      v_mfma_f32_16x16x32_bf16 with A = K tile, B = Q^T fragment (loaded straight from global):
//                    a lane then holds, for ONE query (lane & 15), 4 consecutive keys per 16-key tile -> the softmax
//                    reductions are in-register plus two cross-lane steps (xor 16, xor 32);
//   O^T = V^T P^T    the exponentiated scores, converted to bf16 in place, already ARE the B operand of this product
//                    (k index permuted identically on the V^T fragment), so P never goes through LDS.
//   out = O / rowsum (fp32), 4 consecutive d per lane -> 8-byte bf16 stores.
#include "ch_common.h"
#include "kernels.h"

namespace {

constexpr int HD = 64;

template <int KB>  // number of 32-key blocks (keys padded to KB*32)
__global__ __launch_bounds__(256) void attention_kernel(const bf16_t *__restrict__ qkv, int ntok, int heads, int vs_bytes,
                                                        float scale_log2e, bf16_t *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = KB * 2;     // 16-key tiles
    constexpr int KP = KB * 32;    // padded keys
    char *Ks = smem;               // [KP][128 B]
    char *Vt = smem + KP * 128;    // [64][vs_bytes]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
    const int D = heads * HD;
    const size_t ld = (size_t)3 * D;
    const bf16_t *base = qkv + (size_t)b * ntok * ld + h * HD;

    // ---- stage K (swizzled rows) and V (transposed) ----
    for (int c = tid; c < KP * 8; c += 256) {
        const int row = c >> 3, ch = c & 7;
        uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
        if (row < ntok) {
            kv = *(const uint4 *)(base + (size_t)row * ld + D + ch * 8);
            vv = *(const uint4 *)(base + (size_t)row * ld + 2 * D + ch * 8);
        }
        *(uint4 *)(Ks + row * 128 + ((ch ^ (row & 7)) << 4)) = kv;
        const uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bf16_t val = (bf16_t)((w[e >> 1] >> ((e & 1) * 16)) & 0xffff);
            *(bf16_t *)(Vt + (size_t)(ch * 8 + e) * vs_bytes + row * 2) = val;
        }
    }
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    const int QT = (ntok + 15) >> 4;
    for (int qt = wid; qt < QT; qt += 4) {
        int q = qt * 16 + fr;
        const bool qvalid = q < ntok;
        if (!qvalid) q = ntok - 1;
        const bf16_t *qp = base + (size_t)q * ld + fq * 8;
        const bf16x8 qf0 = *(const bf16x8 *)(qp);
        const bf16x8 qf1 = *(const bf16x8 *)(qp + 32);

        f32x4 st[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int row = kt * 16 + fr;
            const bf16x8 k0 = *(const bf16x8 *)(Ks + row * 128 + (((0 + fq) ^ (row & 7)) << 4));
            const bf16x8 k1 = *(const bf16x8 *)(Ks + row * 128 + (((4 + fq) ^ (row & 7)) << 4));
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf0, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf1, a, 0, 0, 0);
            st[kt] = a;
        }
        // ---- softmax over keys for query (lane & 15): lane holds keys kt*16 + 4*fq + r
        float mx = -1e30f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + fq * 4 + r;
                if (key >= ntok) st[kt][r] = -1e30f;
                mx = fmaxf(mx, st[kt][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = exp2f((st[kt][r] - mx) * scale_log2e);
                st[kt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;

        // ---- O^T = V^T P^T ; logical k = 8*fq + j  <->  key 32*kb + (j < 4 ? 4*fq + j : 16 + 4*fq + j - 4)
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            union {
                bf16x8 v;
                uint32_t u[4];
            } pf;
            pf.u[0] = pack_bf16x2(st[2 * kb][0], st[2 * kb][1]);
            pf.u[1] = pack_bf16x2(st[2 * kb][2], st[2 * kb][3]);
            pf.u[2] = pack_bf16x2(st[2 * kb + 1][0], st[2 * kb + 1][1]);
            pf.u[3] = pack_bf16x2(st[2 * kb + 1][2], st[2 * kb + 1][3]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const char *vrow = Vt + (size_t)(dt * 16 + fr) * vs_bytes + (kb * 32 + fq * 4) * 2;
                union {
                    bf16x8 v;
                    uint2 h[2];
                } vf;
                vf.h[0] = *(const uint2 *)(vrow);
                vf.h[1] = *(const uint2 *)(vrow + 32);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf.v, o[dt], 0, 0, 0);
            }
        }
        if (qvalid) {
            bf16_t *op = out + ((size_t)b * ntok + q) * D + h * HD + fq * 4;
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

template <int KB>
int launch_attn(const bf16_t *qkv, int B, int ntok, int heads, bf16_t *out, hipStream_t s) {
    const int KP = KB * 32;
    int vs = KP * 2;
    vs = ((vs - 16 + 255) / 256) * 256 + 16;  // smallest value >= KP*2 that is == 16 (mod 256)
    const size_t lds = (size_t)KP * 128 + (size_t)64 * vs;
    CH_REQUIRE(lds <= 160 * 1024, "attention: sequence too long for the LDS-resident K/V kernel");
    static bool attr_set = false;
    if (!attr_set) {
        CH_CHECK_HIP(hipFuncSetAttribute((const void *)attention_kernel<KB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds));
        attr_set = true;
    }
    const float scale_log2e = 0.125f * 1.4426950408889634f;  // head_dim^-0.5 * log2(e), head_dim = 64
    hipLaunchKernelGGL(attention_kernel<KB>, dim3(B * heads), dim3(256), lds, s, qkv, ntok, heads, vs, scale_log2e, out);
    CH_LAUNCH_CHECK();
    return 0;
}

}  // namespace

int ch_attention(const bf16_t *qkv, int B, int ntok, int heads, bf16_t *out, hipStream_t s) {
    CH_REQUIRE(B > 0 && ntok > 0 && heads > 0, "attention: empty problem");
    const int KB = (ntok + 31) / 32;
    switch (KB) {
        case 1: return launch_attn<1>(qkv, B, ntok, heads, out, s);
        case 2: return launch_attn<2>(qkv, B, ntok, heads, out, s);
        case 3: return launch_attn<3>(qkv, B, ntok, heads, out, s);
        case 4: return launch_attn<4>(qkv, B, ntok, heads, out, s);
        case 5: return launch_attn<5>(qkv, B, ntok, heads, out, s);
        case 6: return launch_attn<6>(qkv, B, ntok, heads, out, s);
        case 7: return launch_attn<7>(qkv, B, ntok, heads, out, s);
        case 8: return launch_attn<8>(qkv, B, ntok, heads, out, s);
        case 9: return launch_attn<9>(qkv, B, ntok, heads, out, s);
    }
    ch_set_error("attention: more than 288 tokens per image is not supported");
    return 2;
}
