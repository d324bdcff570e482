// bf16 MFMA GEMM, 256x128x32 block tile, 4 waves, three-stage LDS-DMA ring, TWO workgroups per CU.
//
//   C[m][n] = sum_k X[m][k] * W[n][k]      (same contract and fused epilogues as gemm_bf16.hip; see kernels.h)
//
// Why this shape: measured on MI355X (tools/gemm_bench.py ablations) a third of the 256x256
// one-workgroup-per-CU kernel's time is its epilogue -- the whole chip stops computing and bursts its outputs to HBM, then
// HBM idles while everybody computes.  Two independent workgroups per CU drift apart, so one's epilogue (and prologue)
// runs under the other's K loop; on a SIMD the two resident waves (one per workgroup) alternate MFMA and memory segments
// without any explicit choreography.
//   * 4 waves = 2 (M) x 2 (N); a wave owns 128(M) x 64(N): acc[4 n-tiles][8 m-tiles] of v_mfma_f32_16x16x32_bf16 (the same
//     per-wave tile as the 256x256 kernel: 12 ds_read_b128 per 32 MFMAs).
//   * K-tile = 32: rows are 64 B (4 chunks of 16 B); stage image = [X 256 rows][W 128 rows] = 24 KB; 3 stages = 72 KB per
//     workgroup.  Chunk swizzle for 64-B rows: chunk ^ (((row >> 3) & 1) << 1)  -- conflict free for the 16-lane groups of
//     ds_read_b128 (four LDS rows share one 256-B bank row; proof in DESIGN.md).  LDS-DMA writes lane-linear, so the
//     swizzle is applied to the per-lane global source address.
//   * per K-tile: s_waitcnt vmcnt(6) (stage kt landed, stage kt+1's 6 loads may still fly) -> s_barrier -> issue stage
//     kt+2 into the buffer everybody finished reading before the barrier -> 12 fragment reads -> 32 MFMAs.
// N must be a multiple of 128, K a multiple of 32, X padded to a multiple of 256 rows.
#include "ch_common.h"
#include "kernels.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BM = 256, BN = 128, BK = 32;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 24 KiB
constexpr int NSTAGE = 3;
constexpr int NTHREADS = 256;
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;  // 72 KiB (the epilogue reuses the first 64 KiB)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

template <int EPI>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_dp_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;

    const int tiles_n = p.N / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int wg = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int per_group = tiles_m * p.group_n;
    const int g = wg / per_group, rem = wg - g * per_group;
    const int gn = min(p.group_n, tiles_n - g * p.group_n);
    const int tm = rem / gn, tn = g * p.group_n + (rem - tm * gn);
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging: the stage image is 384 rows x 64 B = 24 wave-instructions of 16 rows; wave w issues i = 6w .. 6w+5.
    // lane l -> row 16i + (l >> 2), LDS chunk (l & 3) <- source chunk (l & 3) ^ (((row >> 3) & 1) << 1), (row>>3)&1 == (l>>5)&1
    const int src_chunk = (lane & 3) ^ (((lane >> 5) & 1) << 1);
    uint32_t goff[6];
    const char *gbase[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int row = (wid * 6 + j) * 16 + (lane >> 2);  // 0..383
        if (row < BM) {
            gbase[j] = (const char *)p.X;
            goff[j] = (uint32_t)(((size_t)(m0 + row) * p.K + src_chunk * 8) * 2);
        } else {
            gbase[j] = (const char *)p.W;
            goff[j] = (uint32_t)(((size_t)(n0 + row - BM) * p.K + src_chunk * 8) * 2);
        }
    }
    auto stage = [&](int buf, int kt) {
        char *dst = smem + buf * STAGE_BYTES + wid * 6 * 1024;
        const uint32_t kb = (uint32_t)kt * (BK * 2);
#pragma unroll
        for (int j = 0; j < 6; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(gbase[j] + (goff[j] + kb)), (lds_void_t *)(dst + j * 1024), 16, 0, 0);
    };

    // ---- fragment addressing: row = base + t*16 + (lane & 15); chunk = (lane >> 4) ^ (((row >> 3) & 1) << 1)
    const int fr = lane & 15, fq = lane >> 4;
    const int fsw = (fq ^ (((fr >> 3) & 1) << 1)) << 4;  // (row >> 3) & 1 == (fr >> 3) & 1 because bases are multiples of 16
    const int xrow = (wr * 128 + fr) * 64 + fsw;         // + mt*1024
    const int wrow = BM * 64 + (wc * 64 + fr) * 64 + fsw;  // + nt*1024

    f32x4 acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    stage(0, 0);
    if (nk > 1) stage(1, 1);

    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk)
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < nk) {
            int nb = buf + 2;
            nb = nb >= NSTAGE ? nb - NSTAGE : nb;
            stage(nb, kt + 2);
        }
        const char *sb = smem + buf * STAGE_BYTES;
        bf16x8 wf[4], xf[8];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wf[nt] = *(const bf16x8 *)(sb + wrow + nt * 1024);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) xf[mt] = *(const bf16x8 *)(sb + xrow + mt * 1024);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        buf = buf + 1 == NSTAGE ? 0 : buf + 1;
    }
    // all LDS-DMA has landed (vmcnt(0) in the last iteration); wait until every wave has finished its last fragment reads
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    ch_epi::store_tile<EPI, 8>(p, acc, smem + wid * 16384, m0 + wr * 128, n0 + wc * 64, lane);
}

template <int EPI>
int launch_dp(const GemmParams &p0, hipStream_t s) {
    GemmParams p = p0;
    p.group_n = ch_gemm_group_n(p.M, p.N, p.K, BM, BN);
    const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
    static ch_once_per_device lds_once;
    if (int e = ch_func_max_lds((const void *)gemm_dp_kernel<EPI>, LDS_BYTES, lds_once)) return e;
    hipLaunchKernelGGL(gemm_dp_kernel<EPI>, dim3(tiles), dim3(NTHREADS), LDS_BYTES, s, p);
    CH_LAUNCH_CHECK();
    return 0;
}

}  // namespace

bool ch_gemm_dp_supported(const GemmParams &p) {
    return p.N % BN == 0 && p.K % BK == 0 && p.X_rows_alloc >= round_up64(p.M, BM) &&
           (size_t)round_up64(p.M, BM) * p.K * 2 < (1ull << 32) && (size_t)p.N * p.K * 2 < (1ull << 32);
}

int ch_gemm_bf16_dp(const GemmParams &p, int epi, hipStream_t s) {
    CH_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem");
    CH_REQUIRE(epi == EPI_PATCH || p.bias != nullptr, "gemm: bias is required");
    CH_REQUIRE(ch_gemm_dp_supported(p), "gemm_dp: needs N % 128 == 0, K % 32 == 0, X padded to 256 rows, operands < 4 GiB");
    switch (epi) {
        case EPI_BIAS: return launch_dp<EPI_BIAS>(p, s);
        case EPI_BIAS_QUICKGELU: return launch_dp<EPI_BIAS_QUICKGELU>(p, s);
        case EPI_BIAS_GELU: return launch_dp<EPI_BIAS_GELU>(p, s);
        case EPI_BIAS_RESID: return launch_dp<EPI_BIAS_RESID>(p, s);
        case EPI_SCALE_RESID: return launch_dp<EPI_SCALE_RESID>(p, s);
        case EPI_PATCH: return launch_dp<EPI_PATCH>(p, s);
    }
    ch_set_error("gemm: unknown epilogue");
    return 2;
}
