// Fused adapter + residual update for the ConceptHash encoder (gfx950):
//
//     H[m, :] += a[m, :] + scale * ( GELU( LN(a[m, :]) @ Wd^T + bd ) @ Wu^T + bu )
//
// i.e. `hidden = residual + sub_block_out + Adapter(sub_block_out)` of the reference's CLIPEncoderLayerWithAdapter.forward
// (models/layers/adapter.py:154-159 and :165-170) with Adapter.forward (:46-60: LayerNorm -> down_proj -> exact GELU ->
// up_proj -> * scale).  The unfused chain (LayerNorm kernel, down GEMM, up GEMM) moves 710 MB per call at B=256 for
// 60 GFLOP; this kernel reads `a` (bf16) and read-modify-writes H (fp32): 395 MB, its HBM floor.
//
//   * LayerNorm is folded into the down-projection:  LN(a) @ Wd^T + bd = rstd * (a @ Wd'^T - mean * c) + d  with
//     Wd' = bf16(Wd * gamma), c[n] = sum_k Wd'[n][k], d[n] = sum_k beta[k] Wd[n][k] + bd[n]  (built once at model load).
//     Row statistics come from the same bf16 `a` the MFMA consumes (phase 0), so no extra rounding is introduced.
//   * a workgroup (8 waves, 2 x 4) owns 128 rows:  phase A  down[128 x b]  (K = D, 32-deep K-tiles, wave tile 64 x b/4);
//     phase B  fold + GELU -> bf16 G[128 x b] written straight into LDS in the K-tile layout phase C reads;
//     phase C  for each 256-column chunk of D: up[128 x 256] (K = b) with G as the resident operand and Wu streamed, then
//     the residual epilogue for that chunk (batched 16-byte read-modify-writes) while the next chunk's weights stream in.
//   * both operand streams are 4-deep LDS-DMA rings retired by counted `s_waitcnt vmcnt(N)` (three stages in flight, one
//     barrier per K-tile): a 2-deep version measured 221 us per call, latency-bound on L2 round trips.
//   * LDS rows are 64 B (K-tile 32); chunk swizzle chunk ^ (((row >> 3) & 1) << 1) on the LDS-DMA source address and on
//     every ds_read_b128 (conflict free).
// STATUS: parity-green but NOT the default path -- measured 200 us per call (B = 256) against 179 us for the unfused chain:
// ablations (tools/adapter_bench.py) show its phases add up (statistics 27 + down 35 + up 40 + residual RMW 80 + 31 us)
// because all workgroups run the same phase at the same time; enable with CH_FUSED_ADAPTER=1.
// Requirements (else the host uses the unfused chain): D % 256 == 0, D <= 1024, b_pad in {128, 256, 384}.
#include "ch_common.h"
#include "kernels.h"
#include "gemm_epilogue.h"

namespace {

// Two configurations (measured, tools/adapter_bench.py):
//   BM = 128, 8 waves, 4-deep rings, 160 KB LDS, one workgroup per CU
//   BM =  64, 4 waves, 2-deep rings,  80 KB LDS, TWO workgroups per CU (one's memory phases overlap the other's MFMA phases)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

// wait until at most `stages_in_flight` (<= 2) younger stages (LPS loads each) are outstanding
template <int LPS>
__device__ __forceinline__ void wait_stages(int stages_in_flight) {
    if (stages_in_flight >= 2)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
    else if (stages_in_flight == 1)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int BPAD, int BM, int NST>
__global__ __launch_bounds__(BM * 4, 2) void adapter_fused_kernel(AdapterParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NWAVES = BM / 16;
    constexpr int LDS_BYTES = BM == 64 ? 80 * 1024 : 160 * 1024;
    constexpr int STATS_OFF = LDS_BYTES - BM * 8;  // (mean, rstd) per row: inside the last Wu stage, dead before it is written
    constexpr int UPW = 16 / NWAVES;               // Wu LDS-DMA instructions per wave per stage
    static_assert(NST == 2 || NST == 4, "ring depth");
    constexpr int NT1 = BPAD / 64;                  // 16-col n-tiles per wave in phase A (wave tile 64 x BPAD/4)
    constexpr int SA_BYTES = (BM + BPAD) * 64;      // phase-A stage: [a 128 rows][Wd' BPAD rows] x 64 B  (32 KB at BPAD 384)
    constexpr int IPW = (BM + BPAD) / 16 / NWAVES;  // LDS-DMA instructions per wave per phase-A stage
    constexpr int G_BYTES = (BPAD / 32) * BM * 64;  // G: [BPAD/32 K-tiles][128 rows][64 B]  (96 KB at BPAD 384)
    constexpr int U_OFF = G_BYTES;                  // Wu ring: NST x [256 rows][64 B] = NST x 16 KB
    static_assert(G_BYTES + NST * 16384 <= LDS_BYTES && NST * SA_BYTES <= LDS_BYTES - BM * 8, "LDS budget");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int m0 = blockIdx.x * BM;
    const int D = p.D;
    const int fr = lane & 15, fq = lane >> 4;

    // ---- LDS-DMA addressing: one wave-instruction = 16 rows x 64 B; lane l -> row 16i + (l >> 2),
    // LDS chunk (l & 3) <- source chunk (l & 3) ^ (((l >> 5) & 1) << 1)
    const int src_chunk = (lane & 3) ^ (((lane >> 5) & 1) << 1);
    const int lrow = lane >> 2;
    uint32_t offA[IPW];
    const char *baseA[IPW];
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
        const int row = (wid * IPW + j) * 16 + lrow;  // row of the stage image
        if (row < BM) {
            int m = m0 + row;
            m = m < p.M ? m : p.M - 1;
            baseA[j] = (const char *)p.A;
            offA[j] = (uint32_t)(((size_t)m * D + src_chunk * 8) * 2);
        } else {
            baseA[j] = (const char *)p.Wd;
            offA[j] = (uint32_t)(((size_t)(row - BM) * D + src_chunk * 8) * 2);
        }
    }
    auto stageA = [&](int kt) {
        char *dst = smem + (kt % NST) * SA_BYTES + wid * IPW * 1024;
        const uint32_t kb = (uint32_t)kt * 64;
#pragma unroll
        for (int j = 0; j < IPW; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(baseA[j] + (offA[j] + kb)), (lds_void_t *)(dst + j * 1024), 16, 0, 0);
    };
    const int nkA = D / 32;
    stageA(0);
    if constexpr (NST == 4) {
        if (nkA > 1) stageA(1);
        if (nkA > 2) stageA(2);
    }

    // ================= phase 0: row statistics of a (wave w: rows 16w .. 16w+15), overlapping the first stages' flight =====
    // Two groups of 8 rows; all 16 loads of a group are issued before the first reduction (a per-row load -> wait ->
    // reduce loop costs one HBM round trip per row: 16 x ~2.5 us).
    if (!(p.dbg & 1)) {
        float *stats = (float *)(smem + STATS_OFF);
        const int chunks = D >> 3;                       // 16-byte chunks per row (<= 128)
        const bool on1 = lane + 64 < chunks;
        const int c1 = on1 ? lane + 64 : chunks - 1;     // clamped second chunk (its values are zeroed below)
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
            uint4 u[8][2];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                int m = m0 + wid * 16 + grp * 8 + r;
                m = m < p.M ? m : p.M - 1;
                const bf16_t *src = p.A + (size_t)m * D;
                u[r][0] = *(const uint4 *)(src + (lane < chunks ? lane : 0) * 8);
                u[r][1] = *(const uint4 *)(src + c1 * 8);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float v[2][8];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const uint32_t ww[4] = {u[r][j].x, u[r][j].y, u[r][j].z, u[r][j].w};
                    const bool on = j == 0 ? lane < chunks : on1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[j][2 * e] = on ? bf2f((bf16_t)(ww[e] & 0xffff)) : 0.f;
                        v[j][2 * e + 1] = on ? bf2f((bf16_t)(ww[e] >> 16)) : 0.f;
                    }
                }
                float sm = 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) sm += v[j][e];
                const float mean = wave_sum(sm) / (float)D;
                float q = 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bool on = j == 0 ? lane < chunks : on1;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = v[j][e] - mean;
                        q += on ? d * d : 0.f;
                    }
                }
                const float rstd = rsqrtf(wave_sum(q) / (float)D + p.eps);
                if (lane == 0) {
                    const int row = wid * 16 + grp * 8 + r;
                    stats[row * 2] = mean;
                    stats[row * 2 + 1] = rstd;
                }
            }
        }
    }

    // ================= phase A: down[128 x BPAD] = a[128 x D] @ Wd'^T =================
    const int fsw = (fq ^ (((fr >> 3) & 1) << 1)) << 4;  // swizzled chunk of a fragment read (row bases are multiples of 16)
    f32x4 acc1[NT1][4];
#pragma unroll
    for (int a = 0; a < NT1; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc1[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < ((p.dbg & 2) ? 0 : nkA); ++kt) {
        // phase 0's plain loads have all been consumed, so only LDS-DMA stages are outstanding here
        wait_stages<IPW>(min(nkA - 1 - kt, NST - 2));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + NST - 1 < nkA) stageA(kt + NST - 1);
        const char *sb = smem + (kt % NST) * SA_BYTES;
        bf16x8 xf[4], wf[NT1];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) xf[mt] = *(const bf16x8 *)(sb + (wr * 64 + mt * 16 + fr) * 64 + fsw);
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) wf[nt] = *(const bf16x8 *)(sb + BM * 64 + (wc * (BPAD / 4) + nt * 16 + fr) * 64 + fsw);
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc1[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc1[nt][mt], 0, 0, 0);
    }

    // ================= phase B: fold LN, GELU, G -> LDS =================
    float mean[4], rstd[4];
    {
        const float *stats = (const float *)(smem + STATS_OFF);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            mean[mt] = stats[(wr * 64 + mt * 16 + fr) * 2];
            rstd[mt] = stats[(wr * 64 + mt * 16 + fr) * 2 + 1];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();  // every wave is past its last phase-A fragment reads; statistics are in registers

    // ---- phase C streaming set-up: Wu stage = 256 rows x 64 B = 16 instructions, 2 per wave
    constexpr int nkC = BPAD / 32;  // K-tiles per chunk
    const int nchunk = D / 256;
    const int total = nchunk * nkC;
    uint32_t uoff[UPW];
#pragma unroll
    for (int j = 0; j < UPW; ++j) uoff[j] = (uint32_t)(((size_t)((wid * UPW + j) * 16 + lrow) * BPAD + src_chunk * 8) * 2);
    const char *Wub = (const char *)p.Wu;
    auto stageU = [&](int g) {  // g = chunk * nkC + kt
        const int c = g / nkC, kt = g - c * nkC;
        char *dst = smem + U_OFF + (g % NST) * 16384 + wid * UPW * 1024;
        const uint32_t base = (uint32_t)((size_t)c * 256 * BPAD * 2) + (uint32_t)kt * 64;
#pragma unroll
        for (int j = 0; j < UPW; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(Wub + (uoff[j] + base)), (lds_void_t *)(dst + j * 1024), 16, 0, 0);
    };
    stageU(0);
    if constexpr (NST == 4) {
        if (total > 1) stageU(1);
        if (total > 2) stageU(2);
    }

    {
        char *G = smem;
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) {
            const int nloc = wc * (BPAD / 4) + nt * 16 + fq * 4;  // bottleneck index of this lane's 4 values
            const f32x4 cc = *(const f32x4 *)(p.c + nloc);
            const f32x4 dd = *(const f32x4 *)(p.d + nloc);
            const int kt2 = nloc >> 5, chunk = (nloc & 31) >> 3, half = (nloc >> 2) & 1;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int row = wr * 64 + mt * 16 + fr;
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    v[r] = ch_epi::gelu_erf_f(rstd[mt] * (acc1[nt][mt][r] - mean[mt] * cc[r]) + dd[r]);
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *(uint2 *)(G + kt2 * (BM * 64) + row * 64 + ((chunk ^ (((row >> 3) & 1) << 1)) << 4) + half * 8) = o;
            }
        }
    }
    // G visibility is established by the first barrier of the loop below (after every wave's lgkmcnt(0))
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ================= phase C: per 256-column chunk: up GEMM (K = BPAD) + residual epilogue =================
    const float scale = *p.scale;
    f32x4 acc2[4][4];
    int g = 0;
    for (int c = 0; c < nchunk; ++c) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc2[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < ((p.dbg & 4) ? 0 : nkC); ++kt, ++g) {
            // epilogue loads/stores of the previous chunk are older than every outstanding stage, so the counted wait
            // also retires them (vmcnt is in issue order)
            wait_stages<UPW>(min(total - 1 - g, NST - 2));
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (g + NST - 1 < total) stageU(g + NST - 1);
            const char *ub = smem + U_OFF + (g % NST) * 16384;
            const char *gb = smem + kt * (BM * 64);
            bf16x8 xf[4], wf[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) xf[mt] = *(const bf16x8 *)(gb + (wr * 64 + mt * 16 + fr) * 64 + fsw);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wf[nt] = *(const bf16x8 *)(ub + (wc * 64 + nt * 16 + fr) * 64 + fsw);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc2[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc2[nt][mt], 0, 0, 0);
        }
        // ---- residual epilogue of chunk c: H += a + scale * (acc2 + bu); lane owns 4 contiguous columns of row m.
        // All loads of a half (8 tiles) are issued before the first store.
        const int nb = c * 256 + wc * 64 + fq * 4;
        if (p.dbg & 8) continue;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 hv[8];
            uint2 av[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int nt = i & 3, mt = half * 2 + (i >> 2);
                int m = m0 + wr * 64 + mt * 16 + fr;
                m = m < p.M ? m : p.M - 1;
                const int n = nb + nt * 16;
                hv[i] = *(const f32x4 *)(p.H + (size_t)m * D + n);
                av[i] = *(const uint2 *)(p.A + (size_t)m * D + n);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int nt = i & 3, mt = half * 2 + (i >> 2);
                const int m = m0 + wr * 64 + mt * 16 + fr;
                const int n = nb + nt * 16;
                const f32x4 bu = *(const f32x4 *)(p.bu + n);
                f32x4 h = hv[i] + (acc2[nt][mt] + bu) * scale;
                h[0] += bf2f((bf16_t)(av[i].x & 0xffff));
                h[1] += bf2f((bf16_t)(av[i].x >> 16));
                h[2] += bf2f((bf16_t)(av[i].y & 0xffff));
                h[3] += bf2f((bf16_t)(av[i].y >> 16));
                if (m < p.M) *(f32x4 *)(p.H + (size_t)m * D + n) = h;
            }
        }
    }
}

template <int BPAD, int BM, int NST>
int launch_adapter_cfg(const AdapterParams &p, hipStream_t s) {
    constexpr int LDS_BYTES = BM == 64 ? 80 * 1024 : 160 * 1024;
    static ch_once_per_device lds_once;
    if (int e = ch_func_max_lds((const void *)adapter_fused_kernel<BPAD, BM, NST>, LDS_BYTES, lds_once)) return e;
    const int blocks = (p.M + BM - 1) / BM;
    hipLaunchKernelGGL((adapter_fused_kernel<BPAD, BM, NST>), dim3(blocks), dim3(BM * 4), LDS_BYTES, s, p);
    CH_LAUNCH_CHECK();
    return 0;
}
template <int BPAD>
int launch_adapter(const AdapterParams &p, hipStream_t s) {
    return (p.dbg & 16) ? launch_adapter_cfg<BPAD, 128, 4>(p, s) : launch_adapter_cfg<BPAD, 64, 2>(p, s);
}

}  // namespace

bool ch_adapter_fused_supported(int D, int bpad) {
    return D % 256 == 0 && D <= 1024 && (bpad == 128 || bpad == 256 || bpad == 384);
}

int ch_adapter_fused(const AdapterParams &p, hipStream_t s) {
    CH_REQUIRE(p.M > 0, "adapter: empty problem");
    CH_REQUIRE(ch_adapter_fused_supported(p.D, p.bpad), "adapter_fused: needs D % 256 == 0, D <= 1024, b_pad in {128,256,384}");
    CH_REQUIRE((size_t)p.M * p.D * 2 < (1ull << 32), "adapter_fused: activation matrix >= 4 GiB");
    switch (p.bpad) {
        case 128: return launch_adapter<128>(p, s);
        case 256: return launch_adapter<256>(p, s);
        default: return launch_adapter<384>(p, s);
    }
}
