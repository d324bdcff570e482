// bf16 MFMA GEMM, 256x128x64 block tile, 8 waves, ping-pong schedule with two phases per K-tile (gfx950).
//
//   C[m][n] = sum_k X[m][k] * W[n][k]      (same contract and fused epilogues as gemm_bf16.hip / gemm_pp.hip; see kernels.h)
//
// Why a second tile shape: 51456 rows are 201 row tiles of 256, so a GEMM with N = 768 has 603 tiles of 256x256 = 2.36 rounds
// of 256 workgroups and runs 3 (21 % of the CU-rounds idle); as 1206 tiles of 256x128 it needs 4.71 rounds of half the size and
// runs 5 (6 % idle).  The dispatcher (gemm_bf16.hip) picks the shape with the smaller rounds x tile-cost product.
//
// Structure (the 256x256 kernel's ideas, re-cut so that a phase keeps 16 MFMAs per wave):
//   * 8 waves = 4 (M) x 2 (N); a wave owns a 64 x 64 sub-tile: acc[4 n-tiles][4 m-tiles] of v_mfma_f32_16x16x32_bf16 (weights
//     are the MFMA A operand, activations the B operand -> a lane holds 4 contiguous n), the same wave tile -- and therefore
//     the same epilogue code -- as the 128x128 kernel.
//   * a K-tile (64 deep) is staged as THREE units of 128 rows x 128 B (16 KB, two global_load_lds_dwordx4 per wave):
//     X_a = tile rows 0..127, X_b = rows 128..255, W = the 128 weight rows.  Two K-tile buffers (even / odd), 96 KB; the
//     16-B chunk index is XOR-swizzled with (row & 7) on the per-lane source address and on the ds_read_b128 address.
//   * per K-tile two phases of 16 MFMAs:  A reads the wave's X fragments (8 x b128) + W n-tiles 0,1 (4) -> acc[0..1][*];
//     B reads W n-tiles 2,3 (4) -> acc[2..3][*] (X fragments kept).
//   * prefetch issue per iteration (K-tiles ke = 2j in the even buffer E, ko = 2j+1 in the odd buffer O):
//       phase 1 (ke,A): O.W(ko)        phase 2 (ke,B): E.X_a, E.X_b (ke+2), then vmcnt(4)
//       phase 3 (ko,A): E.W(ke+2)      phase 4 (ko,B): O.X_a, O.X_b (ko+2), then vmcnt(4)
//     RAW: the wait of phase 2 leaves only its own two units in flight, i.e. retires O.X (issued in the previous phase 4) and
//     O.W (phase 1): the whole odd buffer, first read in phase 3, one phase -- two barriers -- later; symmetric in phase 4 for
//     the even buffer of the next iteration.  WAR: every unit is re-staged exactly one phase after its last ds_read (X in
//     phases 1 / 3, W's second half in 2 / 4), and that read has completed (lgkmcnt(0) BEFORE the reading phase's first
//     barrier) before any wave of either stagger group can issue the overwrite.
//   * each phase is {LDS reads + prefetch issue + waits} s_barrier {16 MFMA} s_barrier; waves 4-7 run one barrier behind
//     waves 0-3, so on every SIMD one wave is in its MFMA segment while its partner is in its memory segment.
// K must be a multiple of 128 (an even number of K-tiles), N a multiple of 128, X padded to a multiple of 256 rows.
#include "ch_common.h"
#include "kernels.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BM = 256, BN = 128, BK = 64;
constexpr int UNIT_BYTES = 128 * BK * 2;   // 16 KiB
constexpr int BUF_BYTES = 3 * UNIT_BYTES;  // 48 KiB: [X_a][X_b][W]
constexpr int STAGE_BYTES = 8 * 16384;     // epilogue staging: 16 KB per wave (re-uses the operand buffers)
constexpr int NTHREADS = 512;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

#define PQ_BARRIER()                       \
    do {                                   \
        __builtin_amdgcn_sched_barrier(0); \
        __builtin_amdgcn_s_barrier();      \
        __builtin_amdgcn_sched_barrier(0); \
    } while (0)
#define PQ_WAIT_LGKM0()                                    \
    do {                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                 \
    } while (0)
#define PQ_WAIT_VM(n)                                         \
    do {                                                      \
        asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                    \
    } while (0)

template <int EPI>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_pq_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;  // 4 x 2 waves
    const int grp = wid >> 2;               // stagger group: waves 4-7 run one barrier behind waves 0-3

    const int tiles_n = p.N / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int wg = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    // tile order inside an XCD's contiguous chunk: see gemm_bf16.hip (n-tile groups with L2-resident weight panels)
    const int per_group = tiles_m * p.group_n;
    const int g = wg / per_group, rem = wg - g * per_group;
    const int gn = min(p.group_n, tiles_n - g * p.group_n);
    const int tm = rem / gn, tn = g * p.group_n + (rem - tm * gn);
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- prefetch addressing.  A unit = 16 wave-instructions of 8 rows; wave w issues instructions 2w, 2w+1:
    // lane l -> unit row lr = 16w + 8j + (l>>3), LDS chunk (l&7) <- source chunk (l&7)^(lr&7) = (l&7)^(l>>3).
    const int src_chunk = (lane & 7) ^ (lane >> 3);
    uint32_t xoff[2][2], woff[2];  // [unit a/b][instr j] byte offsets from p.X / p.W (K-tile 0)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int lr = wid * 16 + j * 8 + (lane >> 3);
        xoff[0][j] = (uint32_t)(((size_t)(m0 + lr) * p.K + src_chunk * 8) * 2);
        xoff[1][j] = (uint32_t)(((size_t)(m0 + 128 + lr) * p.K + src_chunk * 8) * 2);
        woff[j] = (uint32_t)(((size_t)(n0 + lr) * p.K + src_chunk * 8) * 2);
    }
    const char *Xb = (const char *)p.X;
    const char *Wb = (const char *)p.W;
    // unit: 0 X_a, 1 X_b, 2 W ; kt = K-tile index ; buffer = kt & 1
    auto issue = [&](int unit, int kt) {
        char *dst = smem + (kt & 1) * BUF_BYTES + unit * UNIT_BYTES + wid * 2048;
        const uint32_t kb = (uint32_t)kt * (BK * 2);
        const char *base = unit < 2 ? Xb : Wb;
        const uint32_t o0 = unit == 0 ? xoff[0][0] : unit == 1 ? xoff[1][0] : woff[0];
        const uint32_t o1 = unit == 0 ? xoff[0][1] : unit == 1 ? xoff[1][1] : woff[1];
        __builtin_amdgcn_global_load_lds((gbl_void_t *)(base + (o0 + kb)), (lds_void_t *)(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t *)(base + (o1 + kb)), (lds_void_t *)(dst + 1024), 16, 0, 0);
    };

    // ---- fragment addressing: row = base_row + t*16 + (lane & 15); (row & 7) == (lane & 7)
    const int fr = lane & 15, fq = lane >> 4;
    const int sw0 = ((fq ^ (lane & 7)) << 4);        // kk = 0 chunk, swizzled
    const int sw1 = (((4 + fq) ^ (lane & 7)) << 4);  // kk = 1
    const int xbase = (wr >> 1) * UNIT_BYTES + ((wr & 1) * 64 + fr) * 128;  // + mt*2048
    const int wbase = 2 * UNIT_BYTES + (wc * 64 + fr) * 128;                // + nt*2048

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 Wf[2][2] = {}, Xf[4][2] = {};  // [tile][kk]

    auto read_w = [&](int buf, int h) {
        const char *b = smem + buf * BUF_BYTES + wbase + h * 2 * 2048;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            Wf[nt][0] = *(const bf16x8 *)(b + nt * 2048 + sw0);
            Wf[nt][1] = *(const bf16x8 *)(b + nt * 2048 + sw1);
        }
    };
    auto read_x = [&](int buf) {
        const char *b = smem + buf * BUF_BYTES + xbase;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            Xf[mt][0] = *(const bf16x8 *)(b + mt * 2048 + sw0);
            Xf[mt][1] = *(const bf16x8 *)(b + mt * 2048 + sw1);
        }
    };
#define PQ_MFMA(NH)                                                                                                       \
    do {                                                                                                                  \
        __builtin_amdgcn_s_setprio(1);                                                                                    \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) acc[(NH) * 2 + nt][mt] =                                     \
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[nt][kk], Xf[mt][kk], acc[(NH) * 2 + nt][mt], 0, 0, 0);         \
        __builtin_amdgcn_s_setprio(0);                                                                                    \
    } while (0)

    const int nk = p.K / BK;  // even, >= 2
    const int J = nk >> 1;

    // ---- prologue: K-tile 0 complete (3 units) + X of K-tile 1; retire K-tile 0 with vmcnt(4)
    f32x4 fold_v[5];
    if constexpr (ch_epi::traits<EPI>::fold) ch_epi::fold_stats_issue(p, m0, tid, fold_v);  // older than every DMA below
    issue(0, 0);
    issue(1, 0);
    issue(2, 0);
    issue(0, 1);
    issue(1, 1);
    PQ_WAIT_VM(4);
    if constexpr (ch_epi::traits<EPI>::fold) {  // per-row (mean, rstd) of the LN-folded input
        ch_epi::fold_stats_finish(p, tid, fold_v, (float *)(smem + STAGE_BYTES));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    PQ_BARRIER();
    if (grp == 1) PQ_BARRIER();  // stagger: waves 4-7 run one barrier behind waves 0-3

    for (int j = 0; j < J; ++j) {
        const bool more = (j + 1 < J);  // K-tiles ke+2 / ko+2 exist
        const int ke = 2 * j, ko = ke + 1;
        // ---- phase 1: even buffer, n-tiles 0,1
        read_x(0);
        read_w(0, 0);
        issue(2, ko);  // O.W of K-tile ko (always exists)
        PQ_WAIT_LGKM0();
        PQ_BARRIER();
        PQ_MFMA(0);
        PQ_BARRIER();
        // ---- phase 2: even buffer, n-tiles 2,3; retire the odd buffer
        read_w(0, 1);
        if (more) {
            issue(0, ke + 2);
            issue(1, ke + 2);
            PQ_WAIT_VM(4);
        } else {
            PQ_WAIT_VM(0);
        }
        PQ_WAIT_LGKM0();
        PQ_BARRIER();
        PQ_MFMA(1);
        PQ_BARRIER();
        // ---- phase 3: odd buffer, n-tiles 0,1
        read_x(1);
        read_w(1, 0);
        if (more) issue(2, ke + 2);
        PQ_WAIT_LGKM0();
        PQ_BARRIER();
        PQ_MFMA(0);
        PQ_BARRIER();
        // ---- phase 4: odd buffer, n-tiles 2,3; retire the even buffer of the next iteration
        read_w(1, 1);
        if (more) {
            issue(0, ko + 2);
            issue(1, ko + 2);
            PQ_WAIT_VM(4);
        }
        PQ_WAIT_LGKM0();
        PQ_BARRIER();
        PQ_MFMA(1);
        PQ_BARRIER();
    }
    if (grp == 0) PQ_BARRIER();  // balance the stagger barrier

    // ---- epilogue (gemm_epilogue.h): every wave has passed the balance barrier, so no wave still reads staged operands and
    // no LDS-DMA is in flight (vmcnt(0) in the last iteration); each wave transposes through its own 16 KB.
    ch_epi::store_tile<EPI, 4>(p, acc, smem + wid * 16384, m0 + wr * 64, n0 + wc * 64, lane,
                               (const float *)(smem + STAGE_BYTES) + 2 * (wr * 64));
}

template <int EPI>
int launch_pq(const GemmParams &p0, hipStream_t s) {
    GemmParams p = p0;
    p.group_n = ch_gemm_group_n(p.M, p.N, p.K, BM, BN);
    const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
    constexpr int lds = STAGE_BYTES + CH_FOLD_LDS_BYTES;  // staging (>= the two operand buffers) + (mean, rstd) table
    static ch_once_per_device lds_once;
    if (int e = ch_func_max_lds((const void *)gemm_pq_kernel<EPI>, lds, lds_once)) return e;
    hipLaunchKernelGGL(gemm_pq_kernel<EPI>, dim3(tiles), dim3(NTHREADS), lds, s, p);
    CH_LAUNCH_CHECK();
    return 0;
}

}  // namespace

bool ch_gemm_pq_supported(const GemmParams &p) {
    return p.N % BN == 0 && p.K % (2 * BK) == 0 && p.X_rows_alloc >= round_up64(p.M, BM) &&
           (size_t)round_up64(p.M, BM) * p.K * 2 < (1ull << 32) && (size_t)p.N * p.K * 2 < (1ull << 32);
}

int ch_gemm_bf16_pq(const GemmParams &p, int epi, hipStream_t s) {
    CH_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem");
    CH_REQUIRE(epi == EPI_PATCH || p.bias != nullptr, "gemm: bias is required");
    CH_REQUIRE(ch_gemm_pq_supported(p), "gemm_pq: needs N % 128 == 0, K % 128 == 0, X padded to 256 rows, operands < 4 GiB");
    switch (epi) {
        case EPI_BIAS: return launch_pq<EPI_BIAS>(p, s);
        case EPI_BIAS_QUICKGELU: return launch_pq<EPI_BIAS_QUICKGELU>(p, s);
        case EPI_BIAS_GELU: return launch_pq<EPI_BIAS_GELU>(p, s);
        case EPI_BIAS_RESID: return launch_pq<EPI_BIAS_RESID>(p, s);
        case EPI_SCALE_RESID: return launch_pq<EPI_SCALE_RESID>(p, s);
        case EPI_PATCH: return launch_pq<EPI_PATCH>(p, s);
        case EPI_BIAS_STATS: return launch_pq<EPI_BIAS_STATS>(p, s);
        case EPI_SCALE_RESID_STATS: return launch_pq<EPI_SCALE_RESID_STATS>(p, s);
        case EPI_FOLD_BIAS: return launch_pq<EPI_FOLD_BIAS>(p, s);
        case EPI_FOLD_QUICKGELU: return launch_pq<EPI_FOLD_QUICKGELU>(p, s);
        case EPI_FOLD_GELU: return launch_pq<EPI_FOLD_GELU>(p, s);
    }
    ch_set_error("gemm: unknown epilogue");
    return 2;
}
