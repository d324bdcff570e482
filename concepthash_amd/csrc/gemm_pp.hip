// bf16 MFMA GEMM, 256x256x64 block tile, 8 waves, ping-pong ("8-phase") schedule for gfx950.
//
//   C[m][n] = sum_k X[m][k] * W[n][k]      (same contract and fused epilogues as gemm_bf16.hip; see kernels.h)
//
// Structure (after the 256^2 8-phase template of the CDNA4 guide):
//   * 8 waves = 2 (M) x 4 (N); a wave owns a 128(M) x 64(N) sub-tile: acc[4 n-tiles][8 m-tiles] of
//     v_mfma_f32_16x16x32_bf16 (weights are the MFMA A operand, activations the B operand -> a lane holds 4 contiguous n).
//   * a K-tile (64 deep) is staged as FOUR half-tiles of 128 rows x 128 B (16 KB, two global_load_lds_dwordx4 per wave):
//       X_h0 / X_h1 = the first / second 64 rows of every wave-row's 128 M-rows, W_h0 / W_h1 = the first / second 32 rows of
//       every wave-column's 64 N-rows -- i.e. split by which QUADRANT phase consumes them, not by position in the tile.
//     Two K-tile buffers (even/odd), 128 KB of LDS; rows are XOR-swizzled on the 16-B chunk index with (row & 7), applied
//     on the per-lane global source address (the LDS-DMA image is lane-linear) and on the ds_read_b128 address.
//   * per K-tile four phases, 16 MFMAs each (one 64x32 quadrant of the wave's tile over the whole K-tile):
//       P1 reads W_h0 (4 x b128) + X_h0 (8 x b128) -> Q(m0,n0);  P2 reads W_h1 (4) -> Q(m0,n1);
//       P3 reads X_h1 (8) -> Q(m1,n1);                          P4 reads nothing -> Q(m1,n0) (W_h0 fragments kept).
//     every phase also issues ONE half-tile of prefetch; loads are retired only by a counted `s_waitcnt vmcnt(6)` at
//     phases 4 and 8 (three half-tiles stay in flight across all barriers; never vmcnt(0) in the steady state).
//   * each phase is {LDS reads + prefetch issue + waits} s_barrier {16 MFMA} s_barrier; waves 4-7 run one barrier behind
//     waves 0-3, so on every SIMD one wave is in its MFMA segment while its partner is in its memory segment.
//   Buffer-reuse distances (proved in DESIGN.md section 3): a half-tile is re-staged no earlier than the phase after its
//   last LDS read -- and that read is complete (lgkmcnt(0)) before the reading wave's barrier; a half-tile is first read
//   one phase after the vmcnt/barrier that retires it.
// K must be a multiple of 128 (an even number of K-tiles), N a multiple of 256, X padded to a multiple of 256 rows.
#include <cstdlib>

#include "ch_common.h"
#include "kernels.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;       // 16 KiB
constexpr int BUF_BYTES = 4 * HALF_BYTES;      // 64 KiB: [X_h0][X_h1][W_h0][W_h1]
constexpr int NTHREADS = 512;
constexpr int PP_LDS_BYTES = 2 * BUF_BYTES + CH_FOLD_LDS_BYTES + 16;  // operands + (mean, rstd) table + split-K ticket

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
// SCHED 2 (free tail): the barriers of the LAST K-tile's four phases are skipped (see the block in front of the K loop)
#define PP_BARRIER_T()                     \
    do {                                   \
        if (!free_tail || more) PP_BARRIER(); \
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

// DBG bits (timing-only builds, results are garbage): 1 = no global loads, 2 = no LDS fragment reads, 4 = no epilogue;
// 8 = (results valid) wave 0 stamps s_memtime / s_memrealtime at entry, after the prologue, after the K loop, after the epilogue's
// last store is issued and after the stores have completed, plus HW_ID / XCC_ID, into p.resid as uint64[tiles][8]
// (tools/tile_timeline.py)
// SCHED 0: four phases of 16 MFMAs per K-tile (the guide's 8-phase template); SCHED 1: two phases of 32 MFMAs per K-tile
// (section "coarse schedule" below): same buffers, same fragments, half the barriers.
// TAG 1: no effect on the code -- a second symbol name for launches the caller marks (GemmParams::tag), so that per-kernel
// profiles (rocprofv3 --stats groups by name) keep fc2 (K = 4 N) apart from out_proj, which shares its epilogue and grid.
// TAG 2: out_bf16 is stored non-temporally (GemmParams::nt_out, set by ch_gemm_bf16 for outputs past the cache size; a template
// parameter because behind a run-time branch the compiler merges the two stores and drops the cache-policy bits).
template <int EPI, int DBG = 0, int SCHED = 0, int TAG = 0>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_pp_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;  // wr also selects the stagger group (waves 4-7 run one barrier behind)
    unsigned long long stamp[6] = {};
    if constexpr (DBG & 8) {
        stamp[0] = __builtin_amdgcn_s_memrealtime();
        stamp[1] = __builtin_amdgcn_s_memtime();
    }

    const int tiles_n = p.N / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    // blocks [0, split_full) own one tile each; the tiles of the last, partial round are cut along K into split_s slices
    // (blocks split_full + tile * split_s + slice) whose fp32 partial tiles the last arriver sums -- see the split-K block
    // after the K loop.  Without a split split_full = tiles and split_s = 1.
    const bool split = (int)blockIdx.x >= p.split_full;  // workgroup-uniform
    const int unit = (int)blockIdx.x - p.split_full;
    const int split_tile = split ? unit / p.split_s : 0, slice = split ? unit - split_tile * p.split_s : 0;
    const int wg = xcd_remap(split ? p.split_full + split_tile : (int)blockIdx.x, tiles_m * tiles_n);
    // tile order inside an XCD's contiguous chunk: n-tiles are taken in groups of p.group_n whose weight panels stay
    // L2-resident (<= ~2.0 MB) while the m-tiles sweep past; inside a group n is fastest so the co-resident workgroups of
    // an XCD share activation panels too.  (host: gemm_group_n)
    const int per_group = tiles_m * p.group_n;
    const int g = wg / per_group, rem = wg - g * per_group;
    const int gn = min(p.group_n, tiles_n - g * p.group_n);
    const int tm_fwd = rem / gn, tn = g * p.group_n + (rem - tm_fwd * gn);
    const int tm = p.rev ? tiles_m - 1 - tm_fwd : tm_fwd;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- prefetch addressing.  A half-tile = 16 wave-instructions of 8 local rows; wave w issues instructions 2w, 2w+1:
    // lane l -> local row lr = 16w + 8j + (l>>3), LDS chunk (l&7) <- source chunk (l&7)^(lr&7) = (l&7)^(l>>3).
    // local row -> tile row:  X_h{h}: lr = r*64 + i  ->  m = r*128 + h*64 + i      (r = lr >> 6, i = lr & 63)
    //                         W_h{h}: lr = c*32 + i  ->  n = c*64  + h*32 + i      (c = lr >> 5, i = lr & 31)
    const int src_chunk = (lane & 7) ^ (lane >> 3);
    uint32_t xoff[2][2], woff[2][2];  // [half][instr j] byte offsets from p.X / p.W (k-tile 0)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int lr = wid * 16 + j * 8 + (lane >> 3);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int xm = (lr >> 6) * 128 + h * 64 + (lr & 63);
            const int wn = (lr >> 5) * 64 + h * 32 + (lr & 31);
            xoff[h][j] = (uint32_t)(((size_t)(m0 + xm) * p.K + src_chunk * 8) * 2);
            woff[h][j] = (uint32_t)(((size_t)(n0 + wn) * p.K + src_chunk * 8) * 2);
        }
    }
    const char *Xb = (const char *)p.X;
    const char *Wb = (const char *)p.W;
    // which: 0 X_h0, 1 X_h1, 2 W_h0, 3 W_h1 ; kt = K-tile index ; buf = kt & 1
    auto issue = [&](int which, int kt) {
        if constexpr (DBG & 1) return;
        char *dst = smem + (kt & 1) * BUF_BYTES + which * HALF_BYTES + wid * 2048;
        const uint32_t kb = (uint32_t)kt * (BK * 2);
        const char *base = which < 2 ? Xb : Wb;
        const uint32_t o0 = which == 0 ? xoff[0][0] : which == 1 ? xoff[1][0] : which == 2 ? woff[0][0] : woff[1][0];
        const uint32_t o1 = which == 0 ? xoff[0][1] : which == 1 ? xoff[1][1] : which == 2 ? woff[0][1] : woff[1][1];
        // (aux 2 = nt on fc2's activation panel, fc1's 316 MB output: measured, no gain -- profiles/r03_cache_policy_ab.txt)
        __builtin_amdgcn_global_load_lds((gbl_void_t *)(base + (o0 + kb)), (lds_void_t *)(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t *)(base + (o1 + kb)), (lds_void_t *)(dst + 1024), 16, 0, 0);
    };

    // ---- fragment addressing (bytes inside a half-tile): row = base_row + t*16 + (lane&15); (row&7) == (lane&7)
    const int fr = lane & 15, fq = lane >> 4;
    const int sw0 = ((fq ^ (lane & 7)) << 4);         // kk = 0 chunk, swizzled
    const int sw1 = (((4 + fq) ^ (lane & 7)) << 4);   // kk = 1
    const int xrow = (wr * 64 + fr) * 128;            // + mt*2048
    const int wrow = (wc * 32 + fr) * 128;            // + nt*2048

    f32x4 acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 Wa[2][2] = {}, Wbf[2][2] = {}, Xf[4][2] = {};  // [tile][kk]

    auto read_w = [&](bf16x8 (&dst)[2][2], int buf, int h) {
        if constexpr (DBG & 2) {
            asm volatile("" : "+v"(dst[0][0]), "+v"(dst[0][1]), "+v"(dst[1][0]), "+v"(dst[1][1]));
            return;
        }
        const char *b = smem + buf * BUF_BYTES + (2 + h) * HALF_BYTES + wrow;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            dst[nt][0] = *(const bf16x8 *)(b + nt * 2048 + sw0);
            dst[nt][1] = *(const bf16x8 *)(b + nt * 2048 + sw1);
        }
    };
    auto read_x = [&](int buf, int h) {
        if constexpr (DBG & 2) {
            asm volatile("" : "+v"(Xf[0][0]), "+v"(Xf[0][1]), "+v"(Xf[1][0]), "+v"(Xf[1][1]), "+v"(Xf[2][0]), "+v"(Xf[2][1]), "+v"(Xf[3][0]), "+v"(Xf[3][1]));
            return;
        }
        const char *b = smem + buf * BUF_BYTES + h * HALF_BYTES + xrow;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            Xf[mt][0] = *(const bf16x8 *)(b + mt * 2048 + sw0);
            Xf[mt][1] = *(const bf16x8 *)(b + mt * 2048 + sw1);
        }
    };
#define PP_MFMA(WF, NH, MH)                                                                                              \
    do {                                                                                                                 \
        __builtin_amdgcn_s_setprio(1);                                                                                   \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)               \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) acc[(NH) * 2 + nt][(MH) * 4 + mt] =                         \
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[nt][kk], Xf[mt][kk], acc[(NH) * 2 + nt][(MH) * 4 + mt], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                                   \
    } while (0)

    const int nk = p.K / BK;  // even, >= 2
    const int J = split ? (nk >> 1) / p.split_s : (nk >> 1);  // iterations (pairs of K-tiles) of this block
    const int j0 = slice * J;
    const int kb0 = 2 * j0;  // first K-tile (even, so the buffer parity of the schedule is unchanged)
    const bool free_tail = SCHED == 2 && ch_epi::traits<EPI>::bf16_only && !split;  // workgroup-uniform

    if constexpr (SCHED == 1) {
        // ---- coarse schedule: per K-tile kt (buffer b = kt & 1) two phases of 32 MFMAs
        //   A(kt): read W_h0, W_h1, X_h0 of buffer b (16 x b128); issue X_h1 of K-tile kt+1 (its slot was last read at B(kt-1));
        //          wait vmcnt(8): retires X_h1 of K-tile kt (read at B(kt), the NEXT phase);  MFMA Q(m0,n0), Q(m0,n1)
        //   B(kt): read X_h1 of buffer b (8 x b128); issue W_h0, W_h1, X_h0 of K-tile kt+2 into buffer b (last read at A(kt),
        //          complete behind that phase's lgkmcnt(0) + barrier); wait vmcnt(8): retires W_h0, W_h1, X_h0 of K-tile kt+1
        //          (read at A(kt+1));  MFMA Q(m1,n1), Q(m1,n0)
        // Issue order per wave: ..., [W_h0 W_h1 X_h0](kt+1) at B(kt-1), X_h1(kt+1) at A(kt), [W_h0 W_h1 X_h0](kt+2) at B(kt), ...
        // = 6, 2, 6, 2 loads, so "all but the youngest 8" is exactly what each wait needs.  Every staged half-tile is read one
        // phase (two barriers) after the wait that retires it; every slot is re-staged one phase after its last read, whose
        // lgkmcnt(0) precedes the reading wave's barrier -- the same distances as the 8-phase schedule, and the stagger
        // (waves 4-7 one barrier behind) is unchanged.  No split-K in this schedule.
        const int nkt = p.K / BK;
        f32x4 fold_v[5];
        if constexpr (ch_epi::traits<EPI>::fold) ch_epi::fold_stats_issue(p, m0, tid, fold_v);  // older than every DMA below
        issue(2, 0);
        issue(3, 0);
        issue(0, 0);
        issue(1, 0);
        issue(2, 1);
        issue(3, 1);
        issue(0, 1);
        PP_WAIT_VM(8);
        if constexpr (ch_epi::traits<EPI>::fold) {
            ch_epi::fold_stats_finish(p, tid, fold_v, (float *)(smem + 2 * BUF_BYTES));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        PP_BARRIER();
        if (wr == 1) PP_BARRIER();
        for (int kt = 0; kt < nkt; kt += 2) {
            // ===== even buffer, K-tile kt =====
            read_w(Wa, 0, 0);
            read_w(Wbf, 0, 1);
            read_x(0, 0);
            issue(1, kt + 1);  // nkt is even: K-tile kt + 1 exists
            PP_WAIT_VM(8);
            PP_WAIT_LGKM0();
            PP_BARRIER();
            PP_MFMA(Wa, 0, 0);
            PP_MFMA(Wbf, 1, 0);
            PP_BARRIER();
            read_x(0, 1);
            if (kt + 2 < nkt) {
                issue(2, kt + 2);
                issue(3, kt + 2);
                issue(0, kt + 2);
                PP_WAIT_VM(8);
            } else {
                PP_WAIT_VM(2);
            }
            PP_WAIT_LGKM0();
            PP_BARRIER();
            PP_MFMA(Wbf, 1, 1);
            PP_MFMA(Wa, 0, 1);
            PP_BARRIER();
            // ===== odd buffer, K-tile kt + 1 =====
            read_w(Wa, 1, 0);
            read_w(Wbf, 1, 1);
            read_x(1, 0);
            if (kt + 2 < nkt) {
                issue(1, kt + 2);
                PP_WAIT_VM(8);
            } else {
                PP_WAIT_VM(0);
            }
            PP_WAIT_LGKM0();
            PP_BARRIER();
            PP_MFMA(Wa, 0, 0);
            PP_MFMA(Wbf, 1, 0);
            PP_BARRIER();
            read_x(1, 1);
            if (kt + 3 < nkt) {
                issue(2, kt + 3);
                issue(3, kt + 3);
                issue(0, kt + 3);
                PP_WAIT_VM(8);
            }
            PP_WAIT_LGKM0();
            PP_BARRIER();
            PP_MFMA(Wbf, 1, 1);
            PP_MFMA(Wa, 0, 1);
            PP_BARRIER();
        }
        if (wr == 0) PP_BARRIER();  // balance the stagger barrier
    } else {
        // ---- prologue: K-tile 0 complete (4 half-tiles) + 3 half-tiles of K-tile 1; retire K-tile 0 with vmcnt(6)
        f32x4 fold_v[5];
        if constexpr (ch_epi::traits<EPI>::fold) ch_epi::fold_stats_issue(p, m0, tid, fold_v);  // older than every DMA below
        issue(0, kb0);
        issue(2, kb0);
        issue(3, kb0);
        issue(1, kb0);
        issue(0, kb0 + 1);
        issue(2, kb0 + 1);
        issue(3, kb0 + 1);
        PP_WAIT_VM(6);
        if constexpr (ch_epi::traits<EPI>::fold) {  // per-row (mean, rstd) of the LN-folded input
            ch_epi::fold_stats_finish(p, tid, fold_v, (float *)(smem + 2 * BUF_BYTES));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        PP_BARRIER();
        if constexpr (DBG & 8) stamp[2] = __builtin_amdgcn_s_memtime();
        if (wr == 1) PP_BARRIER();  // stagger: waves 4-7 run one barrier behind waves 0-3

        // ---- SCHED 2, "free tail" (bf16-output epilogues, no split): in the block's last K-tile nothing is staged any more and every
        // byte of the odd buffer landed behind phase 4's vmcnt(0) + barrier, so its four phases need no barrier for any hazard --
        // the barriers only keep the ping-pong cadence.  Here they are dropped: waves 0-3 take their balancing barrier right after
        // phase 4 (waves 4-7 meet it as their phase-4 closing barrier), then every wave runs phases 5-8 and its epilogue at its own
        // pace.  The epilogue stages through the EVEN buffer (last read in phase 3, i.e. >= 2 barriers ago for every wave; 8 KB per
        // wave, two passes of 64 rows), so a wave that is done starts storing while its SIMD partner still issues MFMAs.
        for (int jj = 0; jj < J; ++jj) {
            const bool more = (jj + 1 < J);  // K-tiles ke+2 / ke+3 belong to this block
            const int ke = 2 * (j0 + jj), ko = ke + 1;
            // ================= even buffer, K-tile ke =================
            // phase 1
            read_w(Wa, 0, 0);
            read_x(0, 0);
            issue(1, ko);  // X_h1[odd] of K-tile 2j+1 (always exists)
            PP_WAIT_LGKM0();
            PP_BARRIER();
            PP_MFMA(Wa, 0, 0);
            PP_BARRIER();
            // phase 2
            read_w(Wbf, 0, 1);
            if (more) issue(0, ke + 2);
            PP_WAIT_LGKM0();
            PP_BARRIER();
            PP_MFMA(Wbf, 1, 0);
            PP_BARRIER();
            // phase 3
            read_x(0, 1);
            if (more) issue(2, ke + 2);
            PP_WAIT_LGKM0();
            PP_BARRIER();
            PP_MFMA(Wbf, 1, 1);
            PP_BARRIER();
            // phase 4: retire the odd buffer (everything issued up to phase 1)
            if (more) {
                issue(3, ke + 2);
                PP_WAIT_VM(6);
            } else {
                PP_WAIT_VM(0);
            }
            PP_BARRIER();
            PP_MFMA(Wa, 0, 1);
            PP_BARRIER();
            if (free_tail && !more && wr == 0) PP_BARRIER();  // the balancing barrier of the stagger, taken early
            // ================= odd buffer, K-tile ko =================
            // phase 5
            read_w(Wa, 1, 0);
            read_x(1, 0);
            if (more) issue(1, ke + 2);
            PP_WAIT_LGKM0();
            PP_BARRIER_T();
            PP_MFMA(Wa, 0, 0);
            PP_BARRIER_T();
            // phase 6
            read_w(Wbf, 1, 1);
            if (more) issue(0, ko + 2);
            PP_WAIT_LGKM0();
            PP_BARRIER_T();
            PP_MFMA(Wbf, 1, 0);
            PP_BARRIER_T();
            // phase 7
            read_x(1, 1);
            if (more) issue(2, ko + 2);
            PP_WAIT_LGKM0();
            PP_BARRIER_T();
            PP_MFMA(Wbf, 1, 1);
            PP_BARRIER_T();
            // phase 8: retire the even buffer of the next iteration (everything issued up to phase 5)
            if (more) {
                issue(3, ko + 2);
                PP_WAIT_VM(6);
            }
            PP_BARRIER_T();
            PP_MFMA(Wa, 0, 1);
            PP_BARRIER_T();
        }
        if (!free_tail && wr == 0) PP_BARRIER();  // balance the stagger barrier
    }

    // ---- split-K tail: publish this slice's partial tile; the LAST arriver (agent-scope ticket) sums all slices in slice
    // order -- its own included, read back from memory, so the result does not depend on who arrives last -- and runs the
    // epilogue.  No workgroup ever waits for another one.  Visibility without cache-wide fences (an agent-scope release /
    // acquire pair writes back and invalidates the whole XCD L2 under the other workgroups' feet: measured +25..40 us per
    // launch): every slab byte is stored write-through (sc1) and drained by the storing wave before the barrier that
    // precedes the ticket, and every slab load is an sc1 load.
    if (split) {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        constexpr unsigned SLAB_BYTES = BM * BN * 4;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(p.splitk_ws, 0, (int)CH_SPLITK_WS_BYTES, 0x00020000);
        const unsigned tile_off = (unsigned)(split_tile * p.split_s) * SLAB_BYTES;
        const unsigned lane_off = (unsigned)tid * 16;
        {
            const unsigned base = tile_off + (unsigned)slice * SLAB_BYTES + lane_off;
#pragma unroll
            for (int i = 0; i < 32; ++i)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, acc[i >> 3][i & 7]), rsrc, base + i * (NTHREADS * 16), 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned *ticket_lds = (unsigned *)(smem + 2 * BUF_BYTES + CH_FOLD_LDS_BYTES);
        if (tid == 0)
            *ticket_lds = __hip_atomic_fetch_add(p.splitk_cnt + split_tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned ticket = *ticket_lds;
        if (ticket != (unsigned)(p.split_s - 1)) return;
        if (tid == 0)  // ready for the next launch
            __hip_atomic_store(p.splitk_cnt + split_tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // no instruction: keeps the slab loads below the ticket
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int sl = 0; sl < p.split_s; ++sl) {
            const unsigned base = tile_off + (unsigned)sl * SLAB_BYTES + lane_off;
#pragma unroll
            for (int i = 0; i < 32; ++i)
                acc[i >> 3][i & 7] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + i * (NTHREADS * 16), 0, 16));
        }
    }

    if constexpr (DBG & 4) {  // keep the accumulators alive with one store per lane
        f32x4 t = acc[0][0];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) t += acc[a][b];
        if (t[0] == 12345.678f) p.out_bf16[tid] = (bf16_t)1;
        return;
    }
    // ---- epilogue (gemm_epilogue.h): every wave has passed the balance barrier, so no wave still reads staged operands
    // and no LDS-DMA is in flight (vmcnt(0) in the last iteration); each wave transposes through its own 16 KB.
    if constexpr (DBG & 8) stamp[3] = __builtin_amdgcn_s_memtime();
    if constexpr (SCHED == 2 && ch_epi::traits<EPI>::bf16_only) {
        if (free_tail) {  // no barrier behind the K loop: stage through this wave's 8 KB of the even buffer, 64 rows per pass
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                f32x4 half[4][4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) half[nt][mt] = acc[nt][pass * 4 + mt];
                ch_epi::store_tile<EPI, 4, false, false, TAG == 2>(p, half, smem + wid * 8192, m0 + wr * 128 + pass * 64, n0 + wc * 64, lane,
                                           (const float *)(smem + 2 * BUF_BYTES) + 2 * (wr * 128 + pass * 64));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // pass 0 fully read back before pass 1 restages the slice
            }
            return;
        }
    }
    ch_epi::store_tile<EPI, 8, false, false, TAG == 2>(p, acc, smem + wid * 16384, m0 + wr * 128, n0 + wc * 64, lane,
                               (const float *)(smem + 2 * BUF_BYTES) + 2 * (wr * 128));
    if constexpr (DBG & 8) {
        stamp[4] = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp[5] = __builtin_amdgcn_s_memtime();
        if (tid == 0) {
            unsigned long long *st = (unsigned long long *)p.resid + (size_t)blockIdx.x * 8;
            st[0] = stamp[0];
            st[1] = stamp[1];
            st[2] = stamp[2];
            st[3] = stamp[3];
            st[4] = stamp[4];
            st[5] = stamp[5];
            st[6] = (unsigned long long)__builtin_amdgcn_s_getreg(0xF804) | ((unsigned long long)__builtin_amdgcn_s_getreg(0xF814) << 32);
            st[7] = __builtin_amdgcn_s_memrealtime();
        }
    }
}

// Tail split: with G compute units, tiles = full rounds * G + R.  The R tiles of the last round would keep R units busy for
// a whole tile time; cut along K into S slices they keep R * S <= G units busy for 1/S of it.  S must divide the K-loop's
// iteration count.
// OPT-IN (model option "splitk", or the debug tap): measured on MI355X at M = 51456 the fix-up costs more than the shorter tail
// saves -- one workgroup moves its 256 KB fp32 slab at only ~25-30 GB/s (write-through stores 3-11 us per slice, sc1 loads
// 9.5 us per slab), against 8 us (K = 768) to 24 us (K = 3072) saved: qkv 206 -> 223 us (S = 2) / 252 us (S = 6),
// fc2 223 -> 229 us.  With agent-scope release/acquire fences instead of sc1 accesses it is another 5-20 us slower (the
// fences write back / invalidate the XCD's whole L2 under the other workgroups).  DESIGN.md section 3.7.
void ch_pp_choose_split(GemmParams &p, int tiles) {
    static const int ncu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    p.split_full = tiles;
    p.split_s = 1;
    if (!(p.splitk_opt || p.force_split) || !p.splitk_ws || !p.splitk_cnt) return;
    const int R = tiles % ncu, J = p.K / (2 * BK);
    if (R == 0) return;
    int S = 1;
    for (int c = 2; c <= 8; ++c)
        if (J % c == 0 && R * c <= ncu && R * c <= 256) S = c;
    if (S < 2 || J / S < 1) return;
    p.split_full = tiles - R;
    p.split_s = S;
}

template <int EPI, int SCHED>
int launch_pp_sched(GemmParams &p, int tiles, hipStream_t s) {
    constexpr int lds = PP_LDS_BYTES;
    const dim3 grid(p.split_full + (tiles - p.split_full) * p.split_s);
    if constexpr (EPI == EPI_BIAS_STATS && SCHED == 0) {
        if (p.tag == 1) {  // same code under a second name (fc2 of the encoder chain): see TAG above
            static ch_once_per_device lds_once_t;
            if (int e = ch_func_max_lds((const void *)gemm_pp_kernel<EPI, 0, SCHED, 1>, lds, lds_once_t)) return e;
            CH_LAUNCH((gemm_pp_kernel<EPI, 0, SCHED, 1>), grid, dim3(NTHREADS), lds, s, p);
            CH_LAUNCH_CHECK();
            return 0;
        }
    }
    if constexpr ((EPI == EPI_BIAS || EPI == EPI_FOLD_BIAS || EPI == EPI_FOLD_QUICKGELU || EPI == EPI_FOLD_GELU || EPI == EPI_FOLD_ACT2_QUICK ||
                   EPI == EPI_FOLD_ACT2_GELU || EPI == EPI_BIAS_DACT_QUICK || EPI == EPI_BIAS_DACT_GELU) && SCHED == 0) {
        if (p.nt_out) {  // TAG 2: out_bf16 stored non-temporally (an output larger than the caches that is read once)
            static ch_once_per_device lds_once_n;
            if (int e = ch_func_max_lds((const void *)gemm_pp_kernel<EPI, 0, SCHED, 2>, lds, lds_once_n)) return e;
            CH_LAUNCH((gemm_pp_kernel<EPI, 0, SCHED, 2>), grid, dim3(NTHREADS), lds, s, p);
            CH_LAUNCH_CHECK();
            ch_gemm_count_nt_launch(1);
            return 0;
        }
    }
    static ch_once_per_device lds_once;
    if (int e = ch_func_max_lds((const void *)gemm_pp_kernel<EPI, 0, SCHED>, lds, lds_once)) return e;
    CH_LAUNCH((gemm_pp_kernel<EPI, 0, SCHED>), grid, dim3(NTHREADS), lds, s, p);
    CH_LAUNCH_CHECK();
    return 0;
}

template <int EPI>
int launch_pp(const GemmParams &p0, hipStream_t s) {
    GemmParams p = p0;
    p.group_n = ch_gemm_group_n(p.M, p.N, p.K, BM, BN, p.group_n_opt);
    const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
    if (p.pp_sched == 1) {  // coarse schedule (experiment, DESIGN.md section 3.8: bit-identical, no faster): no split-K
#ifdef CH_EXPERIMENTS
        p.split_full = tiles;
        p.split_s = 1;
        return launch_pp_sched<EPI, 1>(p, tiles, s);
#else
        return ch_experiments_not_built();
#endif
    }
    if (p.pp_sched == 2) {  // free tail (experiment): barrier-free last K-tile, epilogue staged through the even buffer
#ifdef CH_EXPERIMENTS
        ch_pp_choose_split(p, tiles);
        return launch_pp_sched<EPI, 2>(p, tiles, s);
#else
        return ch_experiments_not_built();
#endif
    }
    ch_pp_choose_split(p, tiles);
    return launch_pp_sched<EPI, 0>(p, tiles, s);
}

}  // namespace

bool ch_gemm_pp_supported(const GemmParams &p) {
    return p.N % BN == 0 && p.K % (2 * BK) == 0 && p.X_rows_alloc >= round_up64(p.M, BM) &&
           (size_t)round_up64(p.M, BM) * p.K * 2 < (1ull << 32) && (size_t)p.N * p.K * 2 < (1ull << 32);
}

int ch_gemm_bf16_pp_dbg(const GemmParams &p0, int dbg, hipStream_t s) {
    GemmParams p = p0;
    p.group_n = ch_gemm_group_n(p.M, p.N, p.K, BM, BN);
    const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
    if (!ch_gemm_pp_supported(p)) return 2;
    p.split_full = tiles;
    p.split_s = 1;
#define PP_DBG_CASE(D)                                                                                                   \
    case D:                                                                                                              \
        (void)hipFuncSetAttribute((const void *)gemm_pp_kernel<EPI_BIAS, D>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                  2 * BUF_BYTES);                                                                        \
        hipLaunchKernelGGL((gemm_pp_kernel<EPI_BIAS, D>), dim3(tiles), dim3(NTHREADS), 2 * BUF_BYTES, s, p);             \
        break;
    switch (dbg) {
        PP_DBG_CASE(1)
        PP_DBG_CASE(2)
        PP_DBG_CASE(3)
        PP_DBG_CASE(4)
        PP_DBG_CASE(5)
        PP_DBG_CASE(6)
        PP_DBG_CASE(7)
        case 8:   // stamped, valid results; EPI_BIAS
        case 9: { // stamped, EPI_BIAS_QUICKGELU
            if (!p.resid) return 2;
            constexpr int lds = PP_LDS_BYTES;
            if (dbg == 8) {
                (void)hipFuncSetAttribute((const void *)gemm_pp_kernel<EPI_BIAS, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                hipLaunchKernelGGL((gemm_pp_kernel<EPI_BIAS, 8>), dim3(tiles), dim3(NTHREADS), lds, s, p);
            } else {
                (void)hipFuncSetAttribute((const void *)gemm_pp_kernel<EPI_BIAS_QUICKGELU, 8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          lds);
                hipLaunchKernelGGL((gemm_pp_kernel<EPI_BIAS_QUICKGELU, 8>), dim3(tiles), dim3(NTHREADS), lds, s, p);
            }
            break;
        }
        default: return 2;
    }
    CH_LAUNCH_CHECK();
    return 0;
}

int ch_gemm_bf16_pp(const GemmParams &p, int epi, hipStream_t s) {
    CH_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem");
    CH_REQUIRE(epi == EPI_PATCH || p.bias != nullptr, "gemm: bias is required");
    CH_REQUIRE(ch_gemm_pp_supported(p), "gemm_pp: needs N % 256 == 0, K % 128 == 0, X padded to 256 rows, operands < 4 GiB");
    switch (epi) {
        case EPI_BIAS: return launch_pp<EPI_BIAS>(p, s);
        case EPI_BIAS_QUICKGELU: return launch_pp<EPI_BIAS_QUICKGELU>(p, s);
        case EPI_BIAS_GELU: return launch_pp<EPI_BIAS_GELU>(p, s);
        case EPI_BIAS_RESID: return launch_pp<EPI_BIAS_RESID>(p, s);
        case EPI_SCALE_RESID: return launch_pp<EPI_SCALE_RESID>(p, s);
        case EPI_PATCH: return launch_pp<EPI_PATCH>(p, s);
        case EPI_BIAS_STATS: return launch_pp<EPI_BIAS_STATS>(p, s);
        case EPI_SCALE_RESID_STATS: return launch_pp<EPI_SCALE_RESID_STATS>(p, s);
        case EPI_FOLD_BIAS: return launch_pp<EPI_FOLD_BIAS>(p, s);
        case EPI_FOLD_QUICKGELU: return launch_pp<EPI_FOLD_QUICKGELU>(p, s);
        case EPI_FOLD_GELU: return launch_pp<EPI_FOLD_GELU>(p, s);
        case EPI_BIAS_DACT_QUICK: return launch_pp<EPI_BIAS_DACT_QUICK>(p, s);
        case EPI_BIAS_DACT_GELU: return launch_pp<EPI_BIAS_DACT_GELU>(p, s);
        case EPI_FOLD_ACT2_QUICK: return launch_pp<EPI_FOLD_ACT2_QUICK>(p, s);
        case EPI_FOLD_ACT2_GELU: return launch_pp<EPI_FOLD_ACT2_GELU>(p, s);
    }
    ch_set_error("gemm: unknown epilogue");
    return 2;
}
