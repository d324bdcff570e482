// bf16 MFMA GEMM, 128x128x32 block tile, 4 waves, FOUR-stage LDS-DMA ring, two workgroups per CU.
//
//   C[m][n] = sum_k X[m][k] * W[n][k]      (same contract and fused epilogues as gemm_bf16.hip; see kernels.h)
//
// Why: the 128x128x64 kernel (gemm_bf16.hip) keeps ONE K-step in flight (`vmcnt(0)` + `__syncthreads()` per step), so the
// adapter down-projection -- N = 384, a long K = 768, no second tile shape that fits it -- runs at the L2 round-trip time per
// K-step (12 steps x ~0.9 us per 128x128 tile, MfmaUtil 0.21).  Here a K-step is 32 deep (16 KB per stage), four stages fit
// the same 64 KB, and THREE K-steps of loads stay in flight behind a counted `s_waitcnt vmcnt(8)` and a raw `s_barrier`.
//   * 4 waves = 2 (M) x 2 (N), wave tile 64x64 = acc[4][4] of v_mfma_f32_16x16x32_bf16; the same MFMA and the same k order per
//     output element as gemm_bf16.hip -> bit-identical results (tests/test_gemm_gpu.py).
//   * stage image = [X 128 rows ; W 128 rows] x 64 B, 16 rows per wave-instruction (wave w issues instructions 4w .. 4w+3);
//     16-B chunk swizzle for 64-B rows: chunk ^ (((row >> 3) & 1) << 1), applied on the per-lane global source address (the
//     LDS-DMA image is lane-linear) and on the ds_read_b128 address -- conflict free for the 16-lane groups (four 64-B rows
//     share one 256-B bank row; the proof is gemm_dp.hip's).
//   * iteration kt: s_waitcnt vmcnt(8) (stage kt landed, stages kt+1, kt+2 may still fly) -> s_barrier (every wave's part
//     of stage kt landed; every wave finished reading stage kt-1, whose MFMAs it has issued) -> issue stage kt+3 into the
//     buffer of stage kt-1 -> 8 fragment reads -> 16 MFMAs.
// N must be a multiple of 128, K a multiple of 32, X padded to a multiple of 128 rows.
//
// RESULT (round 2, MI355X): bit-identical, and SLOWER than the kernel it was meant to replace at the benchmark's size -- down-projection
// 60.6 vs 54.8 us, up-projection 107.1 vs 99.9 us per launch at 51,456 rows; encode 20,350-20,522 vs 20,460-20,505 images/s: a 128x128
// tile of 64x64 wave tiles is LDS-bandwidth bound there, not latency bound -- per 32-deep K-step a workgroup reads 32 KB of fragments
// and receives 16 KB of LDS-DMA for 256 MFMA cycles per SIMD -- so keeping more K-steps in flight buys nothing and the extra barriers
// cost.  SMALL grids are the other regime (round 4, profiles/r04_small_batch_ring_ab.txt): at 1,608 rows (batch 8) a GEMM is 39-78
// workgroups, one per CU with nothing beside it, and a step is 115 launches of 17 us each, every one a chain of L2 round trips; there
// the ring wins -- 1.73 -> 1.63 ms at batch 1, 1.92 -> 1.78 ms at batch 8, 2.76 -> 2.69 ms at batch 32.  The dispatcher (gemm_bf16.hip)
// sends the forward epilogues here up to CH_RING_MAX_ROWS rows; same bits either way.
#include "ch_common.h"
#include "kernels.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 16 KiB
// NS (template): 4 = the ring above (64 KB, two workgroups per CU); 2 = one K-step in flight, 32 KB of LDS and <= 128 VGPRs so that FOUR
// workgroups share a CU (bf16-output epilogues only: their staging is 8 KB per wave) -- the occupancy experiment of round 3
// (not instantiated any more; the bandwidth-bound adapter launches lose 24-34 % when a CU holds one workgroup instead of two).
constexpr int NTHREADS = 256;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

template <int EPI, int NSTAGE>
__global__ __launch_bounds__(NTHREADS, NSTAGE == 2 ? 4 : 2) void gemm_r4_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid & 1, wn = wid >> 1;

    const int tiles_n = p.N / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int wg = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int per_group = tiles_m * p.group_n;
    const int g = wg / per_group, rem = wg - g * per_group;
    const int gn = min(p.group_n, tiles_n - g * p.group_n);
    const int tm_fwd = rem / gn, tn = g * p.group_n + (rem - tm_fwd * gn);
    const int tm = p.rev ? tiles_m - 1 - tm_fwd : tm_fwd;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging: 256 rows x 64 B = 16 wave-instructions of 16 rows; lane l -> row 16i + (l >> 2), LDS chunk (l & 3) holding
    // source chunk (l & 3) ^ (((row >> 3) & 1) << 1), and (row >> 3) & 1 == (l >> 5) & 1
    const int src_chunk = (lane & 3) ^ (((lane >> 5) & 1) << 1);
    const char *gsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wid * 4 + j) * 16 + (lane >> 2);  // 0..255
        const bf16_t *base = row < BM ? p.X + (size_t)(m0 + row) * p.K : p.W + (size_t)(n0 + row - BM) * p.K;
        gsrc[j] = (const char *)(base + src_chunk * 8);
    }
    auto stage = [&](int buf, int kt) {
        char *dst = smem + buf * STAGE_BYTES + wid * 4 * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(gsrc[j] + (size_t)kt * BK * 2), (lds_void_t *)(dst + j * 1024), 16, 0, 0);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addressing: row = base + t*16 + (lane & 15) with base a multiple of 16, chunk = (lane >> 4) ^ (((row >> 3) & 1) << 1)
    const int fr = lane & 15, fq = lane >> 4;
    const int fsw = (fq ^ (((fr >> 3) & 1) << 1)) << 4;
    const int xoff = (wm * 64 + fr) * 64 + fsw;             // + t*1024
    const int woff = BM * 64 + (wn * 64 + fr) * 64 + fsw;   // + t*1024
    const int nk = p.K / BK;

    f32x4 fold_v[5];
    if constexpr (ch_epi::traits<EPI>::fold) ch_epi::fold_stats_issue(p, m0, tid, fold_v);  // older than every DMA below
    constexpr bool PREF = ch_epi::traits<EPI>::scale_resid;
    ch_epi::ResidPrefetch rp;
    if constexpr (PREF) ch_epi::resid_prefetch<EPI>(p, m0 + wm * 64, n0 + wn * 64, lane, rp);  // arrives under the K loop
    stage(0, 0);
    if (NSTAGE > 2 && nk > 1) stage(1, 1);
    if (NSTAGE > 2 && nk > 2) stage(2, 2);
    if constexpr (ch_epi::traits<EPI>::fold) {
        // the statistics loads are older than the stages: retire them (and nothing else) with a counted wait
        if (NSTAGE > 2 && nk > 2)
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (NSTAGE == 2)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        ch_epi::fold_stats_finish(p, tid, fold_v, (float *)(smem + NSTAGE * STAGE_BYTES));
    }

    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = nk - 1 - kt;  // stages issued after stage kt that may still be in flight: min(NSTAGE - 2, ahead)
        if (NSTAGE > 2 && ahead >= 2)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (NSTAGE > 2 && ahead == 1)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + NSTAGE - 1 < nk) stage((buf + NSTAGE - 1) & (NSTAGE - 1), kt + NSTAGE - 1);
        const char *sb = smem + buf * STAGE_BYTES;
        bf16x8 wf[4], xf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            wf[t] = *(const bf16x8 *)(sb + woff + t * 1024);
            xf[t] = *(const bf16x8 *)(sb + xoff + t * 1024);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
        buf = (buf + 1) & (NSTAGE - 1);
    }
    // every LDS-DMA has landed (vmcnt(0) in the last iteration); wait until every wave has finished its last fragment reads
    // (and, PREF, until the prefetched residual has arrived: it is older than the stages and was retired by the loop's waits)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    ch_epi::store_tile<EPI, 4, PREF>(p, acc, smem + wid * (NSTAGE == 2 ? 8192 : 16384), m0 + wm * 64, n0 + wn * 64, lane,
                                     (const float *)(smem + NSTAGE * STAGE_BYTES) + 2 * (wm * 64), &rp);
}

template <int EPI, int NSTAGE = 4>
int launch_r4(const GemmParams &p0, hipStream_t s) {
    GemmParams p = p0;
    p.group_n = ch_gemm_group_n(p.M, p.N, p.K, BM, BN);
    const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
    constexpr int lds = NSTAGE * STAGE_BYTES + (ch_epi::traits<EPI>::fold ? CH_FOLD_LDS_BYTES : 0);
    static ch_once_per_device lds_once;
    if (int e = ch_func_max_lds((const void *)gemm_r4_kernel<EPI, NSTAGE>, lds, lds_once)) return e;
    hipLaunchKernelGGL((gemm_r4_kernel<EPI, NSTAGE>), dim3(tiles), dim3(NTHREADS), lds, s, p);
    CH_LAUNCH_CHECK();
    return 0;
}

}  // namespace

bool ch_gemm_r4_supported(const GemmParams &p, int epi) {   // the forward epilogues (the training-only ones stay with gemm_bf16.hip)
    return epi >= EPI_BIAS && epi <= EPI_FOLD_GELU && p.N % BN == 0 && p.K % BK == 0 && p.K >= 3 * BK && p.X_rows_alloc >= round_up64(p.M, BM);
}

int ch_gemm_bf16_r4(const GemmParams &p, int epi, hipStream_t s) {
    CH_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem");
    CH_REQUIRE(epi == EPI_PATCH || p.bias != nullptr, "gemm: bias is required");
    CH_REQUIRE(ch_gemm_r4_supported(p, epi), "gemm_r4: needs a forward epilogue, N % 128 == 0, K % 32 == 0, K >= 96, X padded to 128 rows");
    switch (epi) {
        case EPI_BIAS: return launch_r4<EPI_BIAS>(p, s);
        case EPI_BIAS_QUICKGELU: return launch_r4<EPI_BIAS_QUICKGELU>(p, s);
        case EPI_BIAS_GELU: return launch_r4<EPI_BIAS_GELU>(p, s);
        case EPI_BIAS_RESID: return launch_r4<EPI_BIAS_RESID>(p, s);
        case EPI_SCALE_RESID: return launch_r4<EPI_SCALE_RESID>(p, s);
        case EPI_PATCH: return launch_r4<EPI_PATCH>(p, s);
        case EPI_BIAS_STATS: return launch_r4<EPI_BIAS_STATS>(p, s);
        case EPI_SCALE_RESID_STATS: return launch_r4<EPI_SCALE_RESID_STATS>(p, s);
        case EPI_FOLD_BIAS: return launch_r4<EPI_FOLD_BIAS>(p, s);
        case EPI_FOLD_QUICKGELU: return launch_r4<EPI_FOLD_QUICKGELU>(p, s);
        case EPI_FOLD_GELU: return launch_r4<EPI_FOLD_GELU>(p, s);
    }
    ch_set_error("gemm: unknown epilogue");
    return 2;
}
