// bf16 MFMA GEMM, 256x256x64 ping-pong schedule (see gemm_pp.hip), PERSISTENT over output tiles.
//
//   C[m][n] = sum_k X[m][k] * W[n][k]   with the bf16-output epilogues (bias / bias+quick_gelu / bias+gelu).
//
// What changes with respect to gemm_pp.hip (same wave layout, half-tiles, phases, vmcnt(6) retire points, stagger):
//   * one workgroup per CU walks tiles  v = blockIdx.x, blockIdx.x + gridDim.x, ...  (XCD chunking and the L2 weight
//     groups of the tile order are unchanged);
//   * the K-tile stream is continuous across tiles: in a tile's last iteration the prefetch slots that would have
//     fetched K-tiles nk, nk+1 fetch K-tiles 0, 1 of the NEXT tile, so the next tile starts with its operands in
//     LDS -- no per-tile prologue latency, and no workgroup launch/teardown between tiles;
//   * the epilogue transposes through a 32 KB staging area of its own (4 KB per wave, four 32-row passes) beside the
//     128 KB operand buffers, issues only stores (the tile's bias is loaded before the K loop), and returns without waiting
//     for them: the stores drain under the next tile's K loop.
// Measured motivation: of a 27 us fc1 tile in the one-tile-per-workgroup kernel ~7 us are epilogue
// drain + workgroup turnaround + prologue latency.
// K % 128 == 0, N % 256 == 0, X padded to a multiple of 256 rows.
#include "ch_common.h"
#include "kernels.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;   // 16 KiB
constexpr int BUF_BYTES = 4 * HALF_BYTES;  // 64 KiB: [X_h0][X_h1][W_h0][W_h1]
constexpr int STAGE_OFF = 2 * BUF_BYTES;   // epilogue staging: 8 waves x 4 KiB
constexpr int LDS_BYTES = STAGE_OFF + 8 * 4096;  // 160 KiB
constexpr int NTHREADS = 512;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

#define PP_BARRIER()                       \
    do {                                   \
        __builtin_amdgcn_sched_barrier(0); \
        __builtin_amdgcn_s_barrier();      \
        __builtin_amdgcn_sched_barrier(0); \
    } while (0)
#define PP_WAIT_LGKM0()                                       \
    do {                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
        __builtin_amdgcn_sched_barrier(0);                    \
    } while (0)
#define PP_WAIT_VM(n)                                         \
    do {                                                      \
        asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                    \
    } while (0)

struct TileOff {  // wave-uniform (lives in SGPRs)
    int m0, n0;
    uint32_t xbase, wbase;  // byte offsets of the tile's first X / W row
};

template <int EPI>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_ppp_kernel(GemmParams p, int ntiles, int skew_ticks, int flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;  // wr also selects the stagger group (waves 4-7 run one barrier behind)
    const int tiles_n = p.N / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int src_chunk = (lane & 7) ^ (lane >> 3);

    // per-lane source offsets are tile independent: (row-in-tile * K + chunk * 8) * 2; the tile adds a wave-uniform base
    uint32_t lx[2][2], lw[2][2];  // [half][instr]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int lr = wid * 16 + j * 8 + (lane >> 3);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int xm = (lr >> 6) * 128 + h * 64 + (lr & 63);
            const int wn = (lr >> 5) * 64 + h * 32 + (lr & 31);
            lx[h][j] = (uint32_t)(((size_t)xm * p.K + src_chunk * 8) * 2);
            lw[h][j] = (uint32_t)(((size_t)wn * p.K + src_chunk * 8) * 2);
        }
    }
    auto tile_offsets = [&](int v, TileOff &t) {
        const int wg = xcd_remap(v, ntiles);
        const int per_group = tiles_m * p.group_n;
        const int g = wg / per_group, rem = wg - g * per_group;
        const int gn = min(p.group_n, tiles_n - g * p.group_n);
        const int tm = rem / gn, tn = g * p.group_n + (rem - tm * gn);
        t.m0 = tm * BM;
        t.n0 = tn * BN;
        t.xbase = (uint32_t)((size_t)t.m0 * p.K * 2);
        t.wbase = (uint32_t)((size_t)t.n0 * p.K * 2);
    };
    const char *Xb = (const char *)p.X;
    const char *Wb = (const char *)p.W;
    // which: 0 X_h0, 1 X_h1, 2 W_h0, 3 W_h1 ; kt = K-tile index inside tile t ; buffer = kt & 1 (nk is even)
    auto issue = [&](const TileOff &t, int which, int kt) {
        char *dst = smem + (kt & 1) * BUF_BYTES + which * HALF_BYTES + wid * 2048;
        const uint32_t kb = (uint32_t)kt * (BK * 2);
        const char *base = (which < 2 ? Xb : Wb) + ((which < 2 ? t.xbase : t.wbase) + kb);  // uniform part
        const uint32_t o0 = which == 0 ? lx[0][0] : which == 1 ? lx[1][0] : which == 2 ? lw[0][0] : lw[1][0];
        const uint32_t o1 = which == 0 ? lx[0][1] : which == 1 ? lx[1][1] : which == 2 ? lw[0][1] : lw[1][1];
        __builtin_amdgcn_global_load_lds((gbl_void_t *)(base + o0), (lds_void_t *)(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t *)(base + o1), (lds_void_t *)(dst + 1024), 16, 0, 0);
    };

    const int fr = lane & 15, fq = lane >> 4;
    const int sw0 = ((fq ^ (lane & 7)) << 4);
    const int sw1 = (((4 + fq) ^ (lane & 7)) << 4);
    const int xrow = (wr * 64 + fr) * 128;
    const int wrow = (wc * 32 + fr) * 128;
    bf16x8 Wa[2][2], Wbf[2][2], Xf[4][2];
    auto read_w = [&](bf16x8 (&dst)[2][2], int buf, int h) {
        const char *b = smem + buf * BUF_BYTES + (2 + h) * HALF_BYTES + wrow;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            dst[nt][0] = *(const bf16x8 *)(b + nt * 2048 + sw0);
            dst[nt][1] = *(const bf16x8 *)(b + nt * 2048 + sw1);
        }
    };
    auto read_x = [&](int buf, int h) {
        const char *b = smem + buf * BUF_BYTES + h * HALF_BYTES + xrow;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            Xf[mt][0] = *(const bf16x8 *)(b + mt * 2048 + sw0);
            Xf[mt][1] = *(const bf16x8 *)(b + mt * 2048 + sw1);
        }
    };
    f32x4 acc[4][8];
#define PP_MFMA(WF, NH, MH)                                                                                              \
    do {                                                                                                                 \
        __builtin_amdgcn_s_setprio(1);                                                                                   \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)               \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) acc[(NH) * 2 + nt][(MH) * 4 + mt] =                         \
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[nt][kk], Xf[mt][kk], acc[(NH) * 2 + nt][(MH) * 4 + mt], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                                   \
    } while (0)

    const int nk = p.K / BK;  // even, >= 2
    const int J = nk >> 1;
    const int G = gridDim.x;
    int v = blockIdx.x;
    TileOff cur, nxt;
    tile_offsets(v, cur);
    // start skew: workgroups that own one tile fewer than the busiest ones have a tile time of slack; spreading their start
    // over it moves their epilogue store bursts away from everyone else's (the stores of a synchronized round otherwise
    // arrive together and drain at HBM write rate while the matrix cores idle)
    {
        const int rem = ntiles % G;
        if (skew_ticks > 0 && rem != 0 && (int)blockIdx.x >= rem) {
            const int local = (int)blockIdx.x >> 3, first = rem >> 3, span = (G >> 3) - first;
            const int slot = (local - first) * 8 + (((int)blockIdx.x & 7) ^ (local & 7));   // neighbours on an XCD differ
            const long long wait = (long long)skew_ticks * slot / (span * 8);
            const long long t0 = wall_clock64();
            while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(8);
        }
    }

    // ---- prologue (first tile only): K-tile 0 complete + 3 half-tiles of K-tile 1
    issue(cur, 0, 0);
    issue(cur, 2, 0);
    issue(cur, 3, 0);
    issue(cur, 1, 0);
    issue(cur, 0, 1);
    issue(cur, 2, 1);
    issue(cur, 3, 1);
    PP_WAIT_VM(6);
    PP_BARRIER();
    if (wr == 1) PP_BARRIER();  // stagger: waves 4-7 run one barrier behind waves 0-3

    for (; v < ntiles; v += G) {
        const bool has_next = v + G < ntiles;
        if (has_next) tile_offsets(v + G, nxt);
        // bias of this lane's 16 columns: fetched BEFORE the K loop (a global load in the epilogue would make the compiler
        // drain the next tile's in-flight LDS-DMA) and parked in this lane's 64 bytes of the idle staging slot, so that it
        // does not occupy 16 VGPRs across the K loop
        {
            char *park = smem + STAGE_OFF + wid * 4096 + lane * 64;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                *(f32x4 *)(park + nt * 16) = *(const f32x4 *)(p.bias + cur.n0 + wc * 64 + nt * 16 + fq * 4);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int j = 0; j < J; ++j) {
            const bool last = (j + 1 == J);
            const bool more = !last || has_next;        // the prefetch slots of this iteration have something to fetch
            const TileOff &pt = last ? nxt : cur;       // ... from this tile, or K-tiles 0 / 1 of the next one
            const int pe = last ? 0 : 2 * j + 2, po = last ? 1 : 2 * j + 3;
            const int ko = 2 * j + 1;
            // ================= even buffer =================
            read_w(Wa, 0, 0);
            read_x(0, 0);
            issue(cur, 1, ko);
            PP_WAIT_LGKM0();
            PP_BARRIER();
            PP_MFMA(Wa, 0, 0);
            PP_BARRIER();
            read_w(Wbf, 0, 1);
            if (more) issue(pt, 0, pe);
            PP_WAIT_LGKM0();
            PP_BARRIER();
            PP_MFMA(Wbf, 1, 0);
            PP_BARRIER();
            read_x(0, 1);
            if (more) issue(pt, 2, pe);
            PP_WAIT_LGKM0();
            PP_BARRIER();
            PP_MFMA(Wbf, 1, 1);
            PP_BARRIER();
            if (more) {
                issue(pt, 3, pe);
                PP_WAIT_VM(6);
            } else {
                PP_WAIT_VM(0);
            }
            PP_BARRIER();
            PP_MFMA(Wa, 0, 1);
            PP_BARRIER();
            // ================= odd buffer =================
            read_w(Wa, 1, 0);
            read_x(1, 0);
            if (more) issue(pt, 1, pe);
            PP_WAIT_LGKM0();
            PP_BARRIER();
            PP_MFMA(Wa, 0, 0);
            PP_BARRIER();
            read_w(Wbf, 1, 1);
            if (more) issue(pt, 0, po);
            PP_WAIT_LGKM0();
            PP_BARRIER();
            PP_MFMA(Wbf, 1, 0);
            PP_BARRIER();
            read_x(1, 1);
            if (more) issue(pt, 2, po);
            PP_WAIT_LGKM0();
            PP_BARRIER();
            PP_MFMA(Wbf, 1, 1);
            PP_BARRIER();
            if (more) {
                issue(pt, 3, po);
                PP_WAIT_VM(6);
            }
            PP_BARRIER();
            PP_MFMA(Wa, 0, 1);
            PP_BARRIER();
        }

        // ---- epilogue: wave-local, 4 KB of staging per wave, four passes of 32 rows; stores only
        {
            char *wl = smem + STAGE_OFF + wid * 4096;
            f32x4 bias[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bias[nt] = *(const f32x4 *)(wl + lane * 64 + nt * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // parked bias read out before pass 0 overwrites the slot
            const int m_base = cur.m0 + wr * 128, n_base = cur.n0 + wc * 64;
            const int lrow = lane >> 3, pos = lane & 7;
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                if (pass) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // previous pass read out before it is overwritten
#pragma unroll
                for (int mt2 = 0; mt2 < 2; ++mt2) {
                    const int row = mt2 * 16 + fr;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const f32x4 val = ch_epi::activate<EPI>(acc[nt][pass * 2 + mt2] + bias[nt]);
                        uint2 o;
                        o.x = pack_bf16x2(val[0], val[1]);
                        o.y = pack_bf16x2(val[2], val[3]);
                        *(uint2 *)(wl + row * 128 + (((nt * 2 + (fq >> 1)) ^ (row & 7)) << 4) + (fq & 1) * 8) = o;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = i * 8 + lrow;
                    const int chunk = pos ^ (row & 7);
                    const uint4 val = *(const uint4 *)(wl + row * 128 + pos * 16);
                    const int m = m_base + pass * 32 + row;
                    if (m < p.M && !(flags & 1)) *(uint4 *)(p.out_bf16 + (size_t)m * p.ldo + n_base + chunk * 8) = val;
                }
            }
        }
        cur = nxt;
    }
    if (wr == 0) PP_BARRIER();  // balance the stagger barrier
}

template <int EPI>
int launch_ppp(const GemmParams &p0, hipStream_t s) {
    GemmParams p = p0;
    p.group_n = ch_gemm_group_n(p.M, p.N, p.K, BM, BN);
    const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
    static ch_once_per_device lds_once;
    if (int e = ch_func_max_lds((const void *)gemm_ppp_kernel<EPI>, LDS_BYTES, lds_once)) return e;
    const int grid = tiles < 256 ? tiles : 256;  // one workgroup per CU (160 KB of LDS each)
    const char *e1 = getenv("CH_PPP_SKEW_NS"), *e2 = getenv("CH_PPP_FLAGS");
    const int skew_ticks = e1 ? atoi(e1) / 10 : 0, flags = e2 ? atoi(e2) : 0;   // wall_clock64: 100 MHz
    hipLaunchKernelGGL(gemm_ppp_kernel<EPI>, dim3(grid), dim3(NTHREADS), LDS_BYTES, s, p, tiles, skew_ticks, flags);
    CH_LAUNCH_CHECK();
    return 0;
}

}  // namespace

bool ch_gemm_ppp_supported(const GemmParams &p, int epi) {
    return (epi == EPI_BIAS || epi == EPI_BIAS_QUICKGELU || epi == EPI_BIAS_GELU) && ch_gemm_pp_supported(p);
}

int ch_gemm_bf16_ppp(const GemmParams &p, int epi, hipStream_t s) {
    CH_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem");
    CH_REQUIRE(p.bias != nullptr, "gemm: bias is required");
    CH_REQUIRE(ch_gemm_ppp_supported(p, epi), "gemm_ppp: bf16-output epilogues only; N % 256 == 0, K % 128 == 0, X padded to 256 rows");
    switch (epi) {
        case EPI_BIAS: return launch_ppp<EPI_BIAS>(p, s);
        case EPI_BIAS_QUICKGELU: return launch_ppp<EPI_BIAS_QUICKGELU>(p, s);
        default: return launch_ppp<EPI_BIAS_GELU>(p, s);
    }
}
