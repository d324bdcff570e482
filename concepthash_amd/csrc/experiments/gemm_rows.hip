// bf16 MFMA GEMM whose workgroup owns WHOLE output rows: 128 rows x N = 384 columns (the adapter bottleneck), 4 waves of 32 rows.
//
//   C[m][n] = sum_k X[m][k] * W[n][k]      (same contract and fused epilogues as gemm_bf16.hip; see kernels.h)
//
// Why this shape: the adapter down-projection (models/layers/adapter.py:46-60, N = b = 384, K = D) does
// not fit the 256x256 kernel (N % 256 != 0) and as 128x128 tiles it is 1,206 workgroups = 2.36 rounds of the 512 slots, each
// re-reading its X rows three times through a two-barrier K loop (MFMA busy 0.21).  Here one workgroup produces all 384 columns of
// its 128 rows, so
//   * X is read once, and 402 workgroups are ONE round of the chip's 512 slots (two per CU: 64 KB of LDS, <= 256 VGPRs);
//   * a wave owns 32 full rows: acc[24 n-tiles][2 m-tiles] of v_mfma_f32_16x16x32_bf16 (weights = MFMA A operand, activations = B
//     operand, as in the other kernels: a lane holds 4 contiguous n of one row) -- 48 MFMAs per 32-deep K-step for 26 ds_read_b128;
//   * it is the first half of the fused adapter (the accumulator of a wave = complete rows of the bottleneck activations, which
//     can feed the up-projection's MFMAs from registers).
// K-step = 32: LDS rows are 64 B; a stage = [X 128 rows][W 384 rows] x 64 B = 32 KB, two stages.  16-B chunk swizzle for 64-B rows:
// chunk ^ (((row >> 3) & 1) << 1) (conflict free for ds_read_b128's 16-lane groups: four LDS rows share one 256-B bank row; same
// proof as gemm_dp.hip), applied on the per-lane global SOURCE address (the LDS-DMA image is lane-linear) and on the read address.
// Same MFMA, same ascending-k accumulation order as gemm_bf16.hip: bit-identical to it (tests/test_gemm_gpu.py).
// MEASURED (round 3, profiles/r03_gemm_rows_ab.txt): 61.3 us vs 58.6 us for the 128x128 kernel at 51,456 rows, 46.2 vs 39.9 us at
// 25,728 rows (201 workgroups do not fill 256 CUs), end to end 12.52-12.69 vs 12.50-12.55 ms per step: NOT faster -- a two-stage
// loop with one 48-MFMA step of prefetch distance waits for its LDS-DMA (MFMA duty 0.32), and a third 32 KB stage costs the second
// workgroup per CU.  Experiment (variant 9 of the taps, CH_GEMM_ROWS=1), not dispatched.
// K must be a multiple of 32, N == 384, X padded to a multiple of 128 rows.
#include "ch_common.h"
#include "kernels.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BM = 128, BN = 384, BK = 32;
constexpr int NT = BN / 16;                     // 24 n-tiles per wave
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 32 KiB
constexpr int NTHREADS = 256;
constexpr int LDS_BYTES = 2 * STAGE_BYTES + CH_FOLD_LDS_BYTES;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

template <int EPI>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_rows_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = (int)blockIdx.x * BM;

    // ---- staging: the stage image is 512 rows x 64 B = 32 wave-instructions of 16 rows; wave w issues i = 8w .. 8w+7.
    // lane l -> row 16 i + (l >> 2), LDS chunk (l & 3) <- source chunk (l & 3) ^ (((row >> 3) & 1) << 1); (row >> 3) & 1 == (l >> 5) & 1
    const int src_chunk = (lane & 3) ^ (((lane >> 5) & 1) << 1);
    uint32_t goff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = (wid * 8 + j) * 16 + (lane >> 2);  // 0..511; instructions 0..7 (wave 0) are the X rows
        goff[j] = row < BM ? (uint32_t)(((size_t)(m0 + row) * p.K + src_chunk * 8) * 2)
                           : (uint32_t)(((size_t)(row - BM) * p.K + src_chunk * 8) * 2);
    }
    const char *Xb = (const char *)p.X, *Wb = (const char *)p.W;
    auto stage = [&](int buf, int kt) {
        char *dst = smem + buf * STAGE_BYTES + wid * 8 * 1024;
        const uint32_t kb = (uint32_t)kt * (BK * 2);
#pragma unroll
        for (int j = 0; j < 8; ++j)   // wid is wave-uniform: wave 0 stages X, waves 1-3 stage W
            __builtin_amdgcn_global_load_lds((gbl_void_t *)((wid == 0 ? Xb : Wb) + (goff[j] + kb)), (lds_void_t *)(dst + j * 1024), 16, 0, 0);
    };

    // ---- fragment addressing: row = base + t*16 + (lane & 15); chunk = (lane >> 4) ^ (((row >> 3) & 1) << 1)
    const int fr = lane & 15, fq = lane >> 4;
    const int fsw = (fq ^ (((fr >> 3) & 1) << 1)) << 4;  // bases are multiples of 16 rows
    const int xrow = (wid * 32 + fr) * 64 + fsw;         // + mt*1024
    const int wrow = BM * 64 + fr * 64 + fsw;            // + nt*1024

    f32x4 acc[NT][2];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        acc[a][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[a][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int nk = p.K / BK;
    f32x4 fold_v[5];
    if constexpr (ch_epi::traits<EPI>::fold) ch_epi::fold_stats_issue(p, m0, tid, fold_v);  // older than every DMA below
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ch_epi::traits<EPI>::fold) {
        ch_epi::fold_stats_finish(p, tid, fold_v, (float *)(smem + 2 * STAGE_BYTES));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    for (int kt = 0; kt < nk; ++kt) {
        // stage kt is in LDS and visible (wait + barrier at the end of the previous iteration); the other buffer was last read in
        // iteration kt - 1, which every wave finished before that barrier
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        const char *sb = smem + (kt & 1) * STAGE_BYTES;
        bf16x8 xf[2];
        xf[0] = *(const bf16x8 *)(sb + xrow);
        xf[1] = *(const bf16x8 *)(sb + xrow + 1024);
#pragma unroll
        for (int g = 0; g < NT / 6; ++g) {
            bf16x8 wf[6];
#pragma unroll
            for (int t = 0; t < 6; ++t) wf[t] = *(const bf16x8 *)(sb + wrow + (g * 6 + t) * 1024);
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                acc[g * 6 + t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], xf[0], acc[g * 6 + t][0], 0, 0, 0);
                acc[g * 6 + t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], xf[1], acc[g * 6 + t][1], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stage kt + 1 landed (this wave's part)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's fragment reads of stage kt are complete
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: six 64-column groups through this wave's 16 KB of the (idle) staging LDS; the loop's last barrier guarantees no
    // wave still reads staged operands and no LDS-DMA is in flight
    const float *row_ms = (const float *)(smem + 2 * STAGE_BYTES) + 2 * (wid * 32);
#pragma unroll
    for (int g = 0; g < NT / 4; ++g) {
        f32x4 part[4][2];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            part[nt][0] = acc[g * 4 + nt][0];
            part[nt][1] = acc[g * 4 + nt][1];
        }
        ch_epi::store_tile<EPI, 2>(p, part, smem + wid * 16384, m0 + wid * 32, g * 64, lane, row_ms);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // group g fully read back before group g + 1 restages the slice
    }
}

template <int EPI>
int launch_rows(const GemmParams &p, hipStream_t s) {
    static ch_once_per_device lds_once;
    if (int e = ch_func_max_lds((const void *)gemm_rows_kernel<EPI>, LDS_BYTES, lds_once)) return e;
    CH_LAUNCH(gemm_rows_kernel<EPI>, dim3((unsigned)((p.M + BM - 1) / BM)), dim3(NTHREADS), LDS_BYTES, s, p);
    CH_LAUNCH_CHECK();
    return 0;
}

}  // namespace

bool ch_gemm_rows_supported(const GemmParams &p, int epi) {
    const bool epi_ok = epi == EPI_BIAS || epi == EPI_BIAS_GELU || epi == EPI_BIAS_QUICKGELU || epi == EPI_FOLD_GELU ||
                        epi == EPI_FOLD_QUICKGELU || epi == EPI_FOLD_BIAS;
    return epi_ok && p.N == BN && p.K % BK == 0 && p.K >= 2 * BK && p.X_rows_alloc >= round_up64(p.M, BM) &&
           (size_t)round_up64(p.M, BM) * p.K * 2 < (1ull << 32);
}

int ch_gemm_bf16_rows(const GemmParams &p, int epi, hipStream_t s) {
    CH_REQUIRE(p.M > 0 && p.bias != nullptr, "gemm_rows: empty problem or missing bias");
    CH_REQUIRE(ch_gemm_rows_supported(p, epi), "gemm_rows: needs N == 384, K % 32 == 0, X padded to 128 rows, a bf16-output epilogue");
    switch (epi) {
        case EPI_BIAS: return launch_rows<EPI_BIAS>(p, s);
        case EPI_BIAS_QUICKGELU: return launch_rows<EPI_BIAS_QUICKGELU>(p, s);
        case EPI_BIAS_GELU: return launch_rows<EPI_BIAS_GELU>(p, s);
        case EPI_FOLD_BIAS: return launch_rows<EPI_FOLD_BIAS>(p, s);
        case EPI_FOLD_QUICKGELU: return launch_rows<EPI_FOLD_QUICKGELU>(p, s);
        case EPI_FOLD_GELU: return launch_rows<EPI_FOLD_GELU>(p, s);
    }
    ch_set_error("gemm_rows: unsupported epilogue");
    return 2;
}
