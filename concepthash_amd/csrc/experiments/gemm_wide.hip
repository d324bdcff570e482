// bf16 MFMA GEMM for N = 384 (the adapter bottleneck of ViT-B/16): 256 x 384 x 32 block tile, 8 waves, three-stage LDS-DMA ring,
// two-phase ping-pong schedule.
//
//   C[m][n] = sum_k X[m][k] * W[n][k]      (same contract and fused epilogues as gemm_bf16.hip; see kernels.h)
//
// EXPERIMENT (round 4; variant 10 of the test taps, model option "wide_kernel", experiments build): bit-identical to the 128x128 kernel,
// 9-14 % faster than it as an isolated launch at 51,456 rows (48.3 vs 53.2 us, exact GELU; 40.4 vs 47.0 us bias only), SLOWER at
// 25,728 rows (one chain of the two-chain default: 42.5 vs 35.7 us) and worth nothing end to end (11.962 vs 11.964 ms per step with
// two chains, 12.691 vs 12.715 with one): NOT dispatched.  profiles/r04_gemm_wide_ab.txt has the A/B, the ablations (bare MFMA +
// barriers 26.4 us, no epilogue 34.7, no loads 37.6, no LDS reads 40.3, full 40.4) and the counters.
//
// Why this shape.  The adapter down-projection (models/layers/adapter.py:46-60: Linear(D -> b), b = D/2 = 384)
// does not fit the 256x256 ping-pong kernel (N % 256 != 0), and as 128x128 tiles it is 1,206 workgroups = 2.36 rounds of the chip
// whose 64x64 wave tiles read 0.75 KB of LDS per MFMA: LDS-port bound at 0.22 of the MFMA peak (49-52 us per launch, 22 launches per
// step).  Every variant with 64x64 or 128x32 wave tiles (256x128 ping-pong, 4-stage ring, whole-row 128x384) landed at the same
// 58-61 us in rounds 2-3.  This kernel gives a wave a 64 x 192 tile -- 0.33 KB of LDS per MFMA, below the port's rate -- and one
// workgroup all 384 columns of 256 rows: X is read ONCE, 201 workgroups are one round of the chip.
//   * 8 waves = 4 (M) x 2 (N); a wave owns 64 (M) x 192 (N): acc[12 n-tiles][4 m-tiles] of v_mfma_f32_16x16x32_bf16 = 192 VGPRs
//     (weights are the MFMA A operand, activations the B operand, as in the other GEMM kernels -> same epilogue code).
//   * K-tile = 32 (rows of 64 B, chunk swizzle chunk ^ (((row >> 3) & 1) << 1) as gemm_dp.hip: conflict-free ds_read_b128); a stage
//     = [X 256 rows | W_h0 192 rows | W_h1 192 rows] = 40 KB, W_h{h} = for both wave columns the h-th 96 of their 192 N-rows, i.e.
//     split by the PHASE that consumes them; three stages = 120 KB.  A 64-deep K-tile would need 160 KB for two stages: the whole LDS.
//   * per K-tile two phases of 24 MFMAs:  A: read X (4 x b128) + W_h0 (6 x b128), issue X of K-tile kt+2;  B: read W_h1 (6 x b128),
//     issue W of K-tile kt+2, `s_waitcnt vmcnt(5)` (retires K-tile kt+1: this wave's 2 + 3 loads of K-tile kt+2 stay in flight).
//     Each phase is {reads + issue + waits} s_barrier {24 MFMA} s_barrier; waves 4-7 run one barrier behind waves 0-3, so on every SIMD one
//     wave is in its MFMA segment while its partner is in its memory segment (the 256x256 kernel's ping-pong).
//   Reuse distances (same argument as gemm_pp.hip / DESIGN.md section 3): a stage is re-staged two phases after its last read, whose
//   lgkmcnt(0) precedes the reading wave's barrier; K-tile kt+1 is first read one phase (two barriers) after the wait that retires it,
//   and every wave passes its own wait before the barrier the readers pass.
// N must be a multiple of 384, K a multiple of 32 and >= 64, X padded to a multiple of 256 rows.
#include "ch_common.h"
#include "kernels.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BM = 256, BN = 384, BK = 32;
constexpr int X_BYTES = BM * BK * 2;             // 16 KiB
constexpr int WH_BYTES = (BN / 2) * BK * 2;      // 12 KiB per half
constexpr int STAGE_BYTES = X_BYTES + 2 * WH_BYTES;  // 40 KiB
constexpr int NSTAGE = 3;
constexpr int NTHREADS = 512;
constexpr int FOLD_OFF = NSTAGE * STAGE_BYTES;   // (rstd, mean) table of the block's 256 rows
constexpr int WIDE_LDS_BYTES = FOLD_OFF + CH_FOLD_LDS_BYTES;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

#define WD_BARRIER()                       \
    do {                                   \
        __builtin_amdgcn_sched_barrier(0); \
        __builtin_amdgcn_s_barrier();      \
        __builtin_amdgcn_sched_barrier(0); \
    } while (0)
#define WD_WAIT_LGKM0()                                       \
    do {                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
        __builtin_amdgcn_sched_barrier(0);                    \
    } while (0)
#define WD_WAIT_VM(n)                                         \
    do {                                                      \
        asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                    \
    } while (0)

// DBG bits (timing-only builds behind the test tap, results are garbage): 1 = no global loads, 2 = no LDS fragment reads, 4 = no epilogue
template <int EPI, bool NTOUT = false, int DBG = 0>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_wide_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;   // wr >= 2 also selects the stagger group (waves 4-7 run one barrier behind)

    const int tiles_n = p.N / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int wg = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging.  A wave-instruction moves 16 rows x 64 B; lane l -> row 16 i + (l >> 2), LDS chunk (l & 3) <- source chunk
    // (l & 3) ^ (((row >> 3) & 1) << 1), and (row >> 3) & 1 == (l >> 5) & 1 because instructions start at multiples of 16 rows.
    //   X: 16 instructions, wave w issues 2w, 2w + 1 (rows 32w .. 32w + 31);
    //   W: 24 instructions over the [W_h0 ; W_h1] image, wave w issues 3w .. 3w + 2 (image rows 48w .. 48w + 47);
    //      image row L -> h = L / 192, c = (L % 192) / 96, i = L % 96  ->  n = c * 192 + h * 96 + i.
    const int src_chunk = (lane & 3) ^ (((lane >> 5) & 1) << 1);
    uint32_t xoff[2], woff[3];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wid * 2 + j) * 16 + (lane >> 2);
        xoff[j] = (uint32_t)(((size_t)(m0 + row) * p.K + src_chunk * 8) * 2);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int L = (wid * 3 + j) * 16 + (lane >> 2);
        const int h = L / 192, lr = L - h * 192, c = lr / 96, i = lr - c * 96;
        woff[j] = (uint32_t)(((size_t)(n0 + c * 192 + h * 96 + i) * p.K + src_chunk * 8) * 2);
    }
    const char *Xb = (const char *)p.X;
    const char *Wb = (const char *)p.W;
    auto issue_x = [&](int kt) {
        if constexpr (DBG & 1) return;
        char *dst = smem + (kt % NSTAGE) * STAGE_BYTES + wid * 2048;
        const uint32_t kb = (uint32_t)kt * (BK * 2);
        __builtin_amdgcn_global_load_lds((gbl_void_t *)(Xb + (xoff[0] + kb)), (lds_void_t *)(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t *)(Xb + (xoff[1] + kb)), (lds_void_t *)(dst + 1024), 16, 0, 0);
    };
    auto issue_w = [&](int kt) {
        if constexpr (DBG & 1) return;
        char *dst = smem + (kt % NSTAGE) * STAGE_BYTES + X_BYTES + wid * 3072;
        const uint32_t kb = (uint32_t)kt * (BK * 2);
        __builtin_amdgcn_global_load_lds((gbl_void_t *)(Wb + (woff[0] + kb)), (lds_void_t *)(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t *)(Wb + (woff[1] + kb)), (lds_void_t *)(dst + 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t *)(Wb + (woff[2] + kb)), (lds_void_t *)(dst + 2048), 16, 0, 0);
    };

    // ---- fragment addressing: row = base + t * 16 + (lane & 15); chunk = (lane >> 4) ^ (((row >> 3) & 1) << 1)
    const int fr = lane & 15, fq = lane >> 4;
    const int fsw = (fq ^ (((fr >> 3) & 1) << 1)) << 4;   // bases are multiples of 16 rows
    const int xrow = (wr * 64 + fr) * 64 + fsw;                         // + mt * 1024
    const int wrow = X_BYTES + (wc * 96 + fr) * 64 + fsw;               // + h * WH_BYTES + nt * 1024

    f32x4 acc[12][4];
#pragma unroll
    for (int a = 0; a < 12; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 Xf[4] = {}, Wf[6] = {};

    auto read_x = [&](int st) {
        if constexpr (DBG & 2) {
            asm volatile("" : "+v"(Xf[0]), "+v"(Xf[1]), "+v"(Xf[2]), "+v"(Xf[3]));
            return;
        }
        const char *b = smem + st * STAGE_BYTES + xrow;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) Xf[mt] = *(const bf16x8 *)(b + mt * 1024);
    };
    auto read_w = [&](int st, int h) {
        if constexpr (DBG & 2) {
            asm volatile("" : "+v"(Wf[0]), "+v"(Wf[1]), "+v"(Wf[2]), "+v"(Wf[3]), "+v"(Wf[4]), "+v"(Wf[5]));
            return;
        }
        const char *b = smem + st * STAGE_BYTES + wrow + h * WH_BYTES;
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) Wf[nt] = *(const bf16x8 *)(b + nt * 1024);
    };
#define WD_MFMA(H)                                                                                                   \
    do {                                                                                                             \
        __builtin_amdgcn_s_setprio(1);                                                                               \
        _Pragma("unroll") for (int nt = 0; nt < 6; ++nt) _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)           \
            acc[(H) * 6 + nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[nt], Xf[mt], acc[(H) * 6 + nt][mt], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                               \
    } while (0)

    const int nk = p.K / BK;   // >= 2
    // ---- prologue: K-tiles 0 and 1 requested (2 + 3 + 2 + 3 loads per wave), K-tile 0 retired by vmcnt(5)
    f32x4 fold_v[5];
    if constexpr (ch_epi::traits<EPI>::fold) ch_epi::fold_stats_issue(p, m0, tid, fold_v);   // older than every DMA below
    issue_x(0);
    issue_w(0);
    issue_x(1);
    issue_w(1);
    WD_WAIT_VM(5);
    if constexpr (ch_epi::traits<EPI>::fold) {   // per-row (rstd, mean) of the LN-folded input
        ch_epi::fold_stats_finish(p, tid, fold_v, (float *)(smem + FOLD_OFF));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    WD_BARRIER();
    if (wr >= 2) WD_BARRIER();   // stagger

    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // phase A: X + W_h0 of this stage; X of K-tile kt + 2 goes to the stage last read two phases ago
        read_x(st);
        read_w(st, 0);
        if (kt + 2 < nk) issue_x(kt + 2);
        WD_WAIT_LGKM0();
        WD_BARRIER();
        WD_MFMA(0);
        WD_BARRIER();
        // phase B: W_h1; W of K-tile kt + 2; retire K-tile kt + 1 (read from the next phase on)
        read_w(st, 1);
        if (kt + 2 < nk) {
            issue_w(kt + 2);
            WD_WAIT_VM(5);
        } else if (kt + 1 < nk) {
            WD_WAIT_VM(0);
        }
        WD_WAIT_LGKM0();
        WD_BARRIER();
        WD_MFMA(1);
        WD_BARRIER();
        st = st + 1 == NSTAGE ? 0 : st + 1;
    }
    if (wr < 2) WD_BARRIER();   // balance the stagger barrier
#undef WD_MFMA

    if constexpr (DBG & 4) {   // keep the accumulators alive with one store per lane
        f32x4 t = acc[0][0];
#pragma unroll
        for (int a = 0; a < 12; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) t += acc[a][b];
        if (t[0] == 12345.678f) p.out_bf16[tid] = (bf16_t)1;
        return;
    }
    // ---- epilogue (gemm_epilogue.h): every wave has passed the balance barrier and no LDS-DMA is in flight; each wave transposes its
    // three 64 x 64 column groups through its own 8 KB of the idle staging LDS, one after the other
    const float *row_ms = (const float *)(smem + FOLD_OFF) + 2 * (wr * 64);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        if (g) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous group fully read back before this one restages
        ch_epi::store_tile<EPI, 4, false, false, NTOUT>(p, *reinterpret_cast<f32x4(*)[4][4]>(&acc[g * 4]), smem + wid * 8192, m0 + wr * 64,
                                                        n0 + wc * 192 + g * 64, lane, row_ms);
    }
}

template <int EPI>
int launch_wide(const GemmParams &p, hipStream_t s) {
    const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
    constexpr int lds = WIDE_LDS_BYTES;
    static ch_once_per_device lds_once;
    if (int e = ch_func_max_lds((const void *)gemm_wide_kernel<EPI>, lds, lds_once)) return e;
    CH_LAUNCH((gemm_wide_kernel<EPI>), dim3(tiles), dim3(NTHREADS), lds, s, p);
    CH_LAUNCH_CHECK();
    return 0;
}

}  // namespace

bool ch_gemm_wide_supported(const GemmParams &p, int epi) {
    const bool epi_ok = epi == EPI_BIAS || epi == EPI_BIAS_GELU || epi == EPI_FOLD_GELU || epi == EPI_FOLD_ACT2_GELU || epi == EPI_BIAS_DACT_GELU;
    return epi_ok && p.N % BN == 0 && p.K % BK == 0 && p.K >= 2 * BK && p.X_rows_alloc >= round_up64(p.M, BM) &&
           (size_t)round_up64(p.M, BM) * p.K * 2 < (1ull << 32) && (size_t)p.N * p.K * 2 < (1ull << 32);
}

int ch_gemm_bf16_wide_dbg(const GemmParams &p, int dbg, hipStream_t s) {   // timing-only ablations (tools/gemm_bench.py variants 41..47)
    if (!ch_gemm_wide_supported(p, EPI_BIAS)) return 2;
    const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
#define WD_DBG_CASE(D)                                                                                                          \
    case D:                                                                                                                     \
        (void)hipFuncSetAttribute((const void *)gemm_wide_kernel<EPI_BIAS, false, D>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  WIDE_LDS_BYTES);                                                                              \
        hipLaunchKernelGGL((gemm_wide_kernel<EPI_BIAS, false, D>), dim3(tiles), dim3(NTHREADS), WIDE_LDS_BYTES, s, p);          \
        break;
    switch (dbg) {
        WD_DBG_CASE(1)
        WD_DBG_CASE(2)
        WD_DBG_CASE(3)
        WD_DBG_CASE(4)
        WD_DBG_CASE(5)
        WD_DBG_CASE(6)
        WD_DBG_CASE(7)
        default: return 2;
    }
    CH_LAUNCH_CHECK();
    return 0;
}

int ch_gemm_bf16_wide(const GemmParams &p, int epi, hipStream_t s) {
    CH_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem");
    CH_REQUIRE(p.bias != nullptr, "gemm: bias is required");
    CH_REQUIRE(ch_gemm_wide_supported(p, epi), "gemm_wide: needs N % 384 == 0, K % 32 == 0, K >= 64, X padded to 256 rows, a bf16-output epilogue");
    switch (epi) {
        case EPI_BIAS: return launch_wide<EPI_BIAS>(p, s);
        case EPI_BIAS_GELU: return launch_wide<EPI_BIAS_GELU>(p, s);
        case EPI_FOLD_GELU: return launch_wide<EPI_FOLD_GELU>(p, s);
        case EPI_FOLD_ACT2_GELU: return launch_wide<EPI_FOLD_ACT2_GELU>(p, s);
        case EPI_BIAS_DACT_GELU: return launch_wide<EPI_BIAS_DACT_GELU>(p, s);
    }
    ch_set_error("gemm_wide: unsupported epilogue");
    return 2;
}
