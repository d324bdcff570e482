// bf16 MFMA GEMM with fused epilogues for the ConceptHash encoder (gfx950 / CDNA4).
//
//   C[m][n] = sum_k X[m][k] * W[n][k]        X: activations [M,K] bf16 row-major, W: nn.Linear weight [N,K] bf16
//
// This is the arithmetic of every Linear on the reference's hot path (HF CLIPAttention q/k/v/out_proj, CLIPMLP fc1/fc2,
// models/layers/adapter.py:46-60 down/up_proj, and the patch-embed Conv2d restated as im2col GEMM,
// models/arch/coop.py:452-466).  The reference runs them as fp32 torch ops; here operands are bf16, accumulation fp32.
//
// Structure (v1): 128x128x64 block tile, 4 waves (2x2), each wave a 64x64 sub-tile = 4x4 v_mfma_f32_16x16x32_bf16
// tiles.  Both operands are staged global->LDS with global_load_lds_dwordx4 (no VGPR round trip), double buffered,
// one barrier per K-step.  LDS rows are 128 B (64 bf16); the 16-B chunk index is XOR-swizzled with (row & 7), applied
// on the per-lane *source* address (the LDS-DMA destination is lane-linear) and again on the ds_read_b128 address,
// which makes every fragment read conflict free.
// The MFMA A operand is the WEIGHT tile and the B operand the ACTIVATION tile, so the accumulator holds D[n][m]:
// a lane owns 4 consecutive n for one m, i.e. 4 contiguous output elements -> 8-B bf16 / 16-B fp32 stores.
#include <cstdlib>

#include "ch_common.h"
#include "kernels.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 32 KiB
constexpr int NTHREADS = 256;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;


// XCD-aware, bijective block remap: blocks b and b+8 share an XCD (observed round-robin dispatch; speed only).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

// NT: the scale+residual epilogues' read-modify-write of the fp32 residual is non-temporal (gemm_epilogue.h: ld_resid).  A parameter of
// the KERNEL template on purpose: wrapping the body in an inlined function shared by two __global__ symbols changed the compiler's
// operand order of commutative instructions in every instance -- and one of the reordered forms is the packed-fp32 op_sel form
// that MI355X executes wrongly next to MFMA waves of another kernel.
template <int EPI, bool NT = false>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid & 1, wn = wid >> 1;

    const int tiles_n = p.N / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int wg = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    // tile order inside an XCD's contiguous chunk: n-tiles are taken in groups of p.group_n whose weight panels stay
    // L2-resident (<= ~2.0 MB) while the m-tiles sweep past; inside a group n is fastest so the co-resident workgroups of
    // an XCD share activation panels too.  (host: gemm_group_n)
    const int per_group = tiles_m * p.group_n;
    const int g = wg / per_group, rem = wg - g * per_group;
    const int gn = min(p.group_n, tiles_n - g * p.group_n);
    const int tm_fwd = rem / gn, tn = g * p.group_n + (rem - tm_fwd * gn);
    const int tm = p.rev ? tiles_m - 1 - tm_fwd : tm_fwd;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging addresses: the stage image is [X rows 0..127 ; W rows 0..127] x 128 B, 8 rows per wave-instruction.
    // wave w issues instructions i = 8w .. 8w+7; lane l -> row 8i + (l>>3), LDS chunk (l&7) holding source chunk
    // (l&7) ^ (row&7).
    const int lrow = lane >> 3;
    const int src_chunk = (lane & 7) ^ lrow;  // (8i + lrow) & 7 == lrow
    const char *gsrc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = (wid * 8 + j) * 8 + lrow;  // 0..255
        const bf16_t *base = row < BM ? p.X + (size_t)(m0 + row) * p.K : p.W + (size_t)(n0 + row - BM) * p.K;
        gsrc[j] = (const char *)(base + src_chunk * 8);
    }
    auto stage = [&](int buf, int kt) {
        char *dst = smem + buf * STAGE_BYTES + wid * 8 * 1024;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(gsrc[j] + (size_t)kt * BK * 2), (lds_void_t *)(dst + j * 1024),
                                             16, 0, 0);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (bytes) inside a stage: row = sub*16 + (lane&15), chunk = (kk*4 + (lane>>4)) ^ (row&7)
    const int frow = lane & 15, fq = lane >> 4;
    const int nk = p.K / BK;

    f32x4 fold_v[5];
    if constexpr (ch_epi::traits<EPI>::fold) ch_epi::fold_stats_issue(p, m0, tid, fold_v);
    stage(0, 0);
    // read-modify-write epilogues: fetch the residual tile now, so that it arrives under the K loop
    constexpr bool PREF = ch_epi::traits<EPI>::scale_resid;
    ch_epi::ResidPrefetch rp;
    if constexpr (PREF) ch_epi::resid_prefetch<EPI, NT>(p, m0 + wm * 64, n0 + wn * 64, lane, rp);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ch_epi::traits<EPI>::fold)  // per-row (mean, rstd) of the LN-folded input
        ch_epi::fold_stats_finish(p, tid, fold_v, (float *)(smem + 2 * STAGE_BYTES));
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char *xs = smem + cur * STAGE_BYTES;
        const char *ws = xs + BM * BK * 2;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 wf[4], xf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int wr = wn * 64 + t * 16 + frow;
                wf[t] = *(const bf16x8 *)(ws + wr * 128 + (((kk * 4 + fq) ^ (wr & 7)) << 4));
                const int xr = wm * 64 + t * 16 + frow;
                xf[t] = *(const bf16x8 *)(xs + xr * 128 + (((kk * 4 + fq) ^ (xr & 7)) << 4));
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue (gemm_epilogue.h): transpose through this wave's 16 KB of the idle staging LDS, full-line stores.
    // The loop's last __syncthreads() guarantees no wave still reads staged operands.
    ch_epi::store_tile<EPI, 4, PREF, NT>(p, acc, smem + wid * 16384, m0 + wm * 64, n0 + wn * 64, lane,
                                         (const float *)(smem + 2 * STAGE_BYTES) + 2 * (wm * 64), &rp);
}

template <int EPI>
int launch(const GemmParams &p0, hipStream_t s) {
    GemmParams p = p0;
    p.group_n = ch_gemm_group_n(p.M, p.N, p.K, BM, BN, p.group_n_opt);
    const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
    constexpr int lds = 2 * STAGE_BYTES + (ch_epi::traits<EPI>::fold ? CH_FOLD_LDS_BYTES : 0);
    static ch_once_per_device lds_once;
    if (int e = ch_func_max_lds((const void *)gemm_bf16_kernel<EPI>, lds, lds_once)) return e;
    if constexpr (ch_epi::traits<EPI>::scale_resid) {
        if (p.nt_resid) {
            static ch_once_per_device lds_once_nt;
            if (int e = ch_func_max_lds((const void *)gemm_bf16_kernel<EPI, true>, lds, lds_once_nt)) return e;
            CH_LAUNCH((gemm_bf16_kernel<EPI, true>), dim3(tiles), dim3(NTHREADS), lds, s, p);
            CH_LAUNCH_CHECK();
            ch_gemm_count_nt_launch(0);
            return 0;
        }
    }
    CH_LAUNCH(gemm_bf16_kernel<EPI>, dim3(tiles), dim3(NTHREADS), lds, s, p);
    CH_LAUNCH_CHECK();
    return 0;
}

}  // namespace

int ch_gemm_bf16_v1(const GemmParams &p, int epi, hipStream_t s) {
    CH_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem");
    CH_REQUIRE(epi == EPI_PATCH || p.bias != nullptr, "gemm: bias is required");
    CH_REQUIRE(p.N % BN == 0, "gemm: N must be a multiple of 128");
    CH_REQUIRE(p.K % BK == 0, "gemm: K must be a multiple of 64");
    CH_REQUIRE(p.X_rows_alloc >= round_up64(p.M, BM), "gemm: X must be allocated for M rounded up to 128 rows");
    switch (epi) {
        case EPI_BIAS: return launch<EPI_BIAS>(p, s);
        case EPI_BIAS_QUICKGELU: return launch<EPI_BIAS_QUICKGELU>(p, s);
        case EPI_BIAS_GELU: return launch<EPI_BIAS_GELU>(p, s);
        case EPI_BIAS_RESID: return launch<EPI_BIAS_RESID>(p, s);
        case EPI_SCALE_RESID: return launch<EPI_SCALE_RESID>(p, s);
        case EPI_PATCH: return launch<EPI_PATCH>(p, s);
        case EPI_BIAS_STATS: return launch<EPI_BIAS_STATS>(p, s);
        case EPI_SCALE_RESID_STATS: return launch<EPI_SCALE_RESID_STATS>(p, s);
        case EPI_FOLD_BIAS: return launch<EPI_FOLD_BIAS>(p, s);
        case EPI_FOLD_QUICKGELU: return launch<EPI_FOLD_QUICKGELU>(p, s);
        case EPI_FOLD_GELU: return launch<EPI_FOLD_GELU>(p, s);
        case EPI_BIAS_DACT_QUICK: return launch<EPI_BIAS_DACT_QUICK>(p, s);
        case EPI_BIAS_DACT_GELU: return launch<EPI_BIAS_DACT_GELU>(p, s);
        case EPI_FOLD_ACT2_QUICK: return launch<EPI_FOLD_ACT2_QUICK>(p, s);
        case EPI_FOLD_ACT2_GELU: return launch<EPI_FOLD_ACT2_GELU>(p, s);
    }
    ch_set_error("gemm: unknown epilogue");
    return 2;
}

// Estimated beyond-L2 read traffic for n-groups of gn tiles: every group re-streams X once; a group's weight panels are
// fetched once per XCD if they fit in half of the 4 MB L2, otherwise once per round of co-resident tiles.
int ch_gemm_group_n(int M, int N, int K, int bm, int bn, int forced) {
    const int tiles_n = N / bn, tiles_m = (M + bm - 1) / bm;
    const double xbytes = 2.0 * M * K, wpanel = 2.0 * bn * K, wbytes = 2.0 * N * K;
    const double rounds = (double)tiles_m * tiles_n / 256.0 + 1.0;
    if (forced > 0) return forced < tiles_n ? forced : tiles_n;   // model option "group_n" (tools/gemm_bench.py sweeps)
    int best = tiles_n;
    double best_cost = 1e300;
    for (int gn = 1; gn <= tiles_n; ++gn) {
        const int ng = (tiles_n + gn - 1) / gn;
        const bool fits = gn * wpanel <= 2.0e6;  // measured (fc1, K = 768): 4-5 panels (1.6-2.0 MB) 270 us, 6 panels (2.4 MB) 286 us
        const double cost = xbytes * ng + wbytes * 8.0 * (fits ? 1.0 : rounds);
        if (cost < best_cost - 1.0) {
            best_cost = cost;
            best = gn;
        }
    }
    return best;
}

// ---- dispatcher: the 256x256 ping-pong kernel when the shape allows it, this file's 128x128 kernel otherwise ---------
static std::atomic<int64_t> g_dispatch_count[5];   // 128x128 | 256x256 | non-temporal residual instance | non-temporal output instance | 256x384
void ch_gemm_count_nt_launch(int kind) { g_dispatch_count[2 + (kind != 0)].fetch_add(1, std::memory_order_relaxed); }
static int g_gemm_variant = 0;  // 0 auto, 1 force v1 (128x128 two-phase), 2 force pp (256x256 ping-pong), 3 force dp
void ch_gemm_set_variant(int v) { g_gemm_variant = v; }
// residual tensors from this size up are streamed past the caches by the scale+residual epilogues (gemm_epilogue.h: ld_resid);
// the model option "resid_nt" (1 / -1) forces the choice
static int resid_nt_choice(const GemmParams &p, int epi) {
    if (epi != EPI_SCALE_RESID && epi != EPI_SCALE_RESID_STATS) return 0;
    if (p.nt_resid_opt) return p.nt_resid_opt > 0;
    return std::max<int64_t>(p.M, p.footprint_rows) * p.N * 4 >= (48ll << 20);
}
// bf16 outputs of the plain / LN-folded epilogues from this size up (fc1's 316 MB and qkv's 237 MB at batch 256: read once, by the
// next kernel, out of HBM whatever the policy) are stored non-temporally by the 256x256 kernel, which leaves the caches to the
// operands: 12.15 -> 11.99 ms per encode step (profiles/r03_cache_policy_ab.txt).  The model option "nt_out" (1 / -1) forces the choice.
static int out_nt_choice(const GemmParams &p, int epi) {
    if (epi != EPI_BIAS && epi != EPI_FOLD_BIAS && epi != EPI_FOLD_QUICKGELU && epi != EPI_FOLD_GELU && epi != EPI_FOLD_ACT2_QUICK &&
        epi != EPI_FOLD_ACT2_GELU && epi != EPI_BIAS_DACT_QUICK && epi != EPI_BIAS_DACT_GELU)
        return 0;
    if (p.nt_out_opt) return p.nt_out_opt > 0;
    return std::max<int64_t>(p.M, p.footprint_rows) * p.N * 2 >= (128ll << 20);
}
int ch_gemm_bf16(const GemmParams &p0, int epi, hipStream_t s) {
    GemmParams p = p0;
    p.nt_resid = resid_nt_choice(p, epi);
    p.nt_out = out_nt_choice(p, epi);
    if (epi == EPI_BIAS_DACT_QUICK || epi == EPI_BIAS_DACT_GELU) CH_REQUIRE(p.aux != nullptr, "gemm: derivative epilogue needs aux (the pre-activation)");
    if (epi == EPI_FOLD_ACT2_QUICK || epi == EPI_FOLD_ACT2_GELU) CH_REQUIRE(p.hb_out != nullptr, "gemm: two-output epilogue needs hb_out");
    if (epi == EPI_FOLD_BIAS || epi == EPI_FOLD_QUICKGELU || epi == EPI_FOLD_GELU || epi == EPI_FOLD_ACT2_QUICK || epi == EPI_FOLD_ACT2_GELU)
        CH_REQUIRE(p.stats_in && p.fold_c && p.K % 128 == 0 && p.K <= 1280 && p.ln_eps > 0.f,
                   "gemm: LN-folded epilogue needs stats_in, fold_c, ln_eps and K % 128 == 0, K <= 1280");
    if (epi == EPI_BIAS_STATS || epi == EPI_SCALE_RESID_STATS)
        CH_REQUIRE(p.stats_out && (epi == EPI_BIAS_STATS || p.hb_out), "gemm: statistics epilogue needs stats_out (and hb_out)");
    if (g_gemm_variant == 1) return ch_gemm_bf16_v1(p, epi, s);
    if (g_gemm_variant == 2) return ch_gemm_bf16_pp(p, epi, s);
    if (g_gemm_variant == 4) {  // 256x256 kernel, coarse schedule
        GemmParams q = p;
        q.pp_sched = 1;
        return ch_gemm_bf16_pp(q, epi, s);
    }
    if (g_gemm_variant == 9) return ch_gemm_bf16_rows(p, epi, s);
    if (g_gemm_variant == 8) {  // 256x256 kernel, free tail
        GemmParams q = p;
        q.pp_sched = 2;
        return ch_gemm_bf16_pp(q, epi, s);
    }
    if (g_gemm_variant == 3) return ch_gemm_bf16_dp(p, epi, s);
    if (g_gemm_variant == 10) return ch_gemm_bf16_wide(p, epi, s);
    // Short-K GEMMs (the adapter up-projection, K = 384) are epilogue/HBM bound: two 128x128 workgroups per CU overlap one's
    // read-modify-write epilogue with the other's K loop and win there (measured: 152 -> 97 us per launch in the pipeline);
    // the 256x256 ping-pong kernel wins from K = 512 up.
    const int min_k = p.pp_min_k > 0 ? p.pp_min_k : 512;  // per model (option "pp_min_k"), not per process
    // Small grids: fewer than 128 tiles of 256x256 leave more than half of the 256 CUs idle; 128x128 tiles (4x as many, two
    // workgroups per CU) win there -- measured at 6,432 rows (batch 32): out 28 vs 36 us, fc2 57 vs 73 us; at 12,864 rows and
    // N = 768 (153 tiles) the two tie.  Not applied when CH_GEMM_PP_MIN_K pins the choice (parity tests on small fixtures).
    const int64_t tiles_pp = ceil_div64(p.M, 256) * (p.N / 256);
    const bool pp = ch_gemm_pp_supported(p) && p.K >= min_k && (p.pp_min_k > 0 || tiles_pp >= 128);
    // N = 384 (the adapter bottleneck of ViT-B/16) does not fit the 256x256 kernel; as 128x128 tiles it is LDS-port bound (64x64 wave
    // tiles).  experiments/gemm_wide.hip (256x384 tiles, 64x192 wave tiles, X read once, one round of the chip) is 9-14 % faster as an
    // isolated launch and neutral end to end (profiles/r04_gemm_wide_ab.txt): opt-in, model option "wide_kernel" = 1, experiments build.
    const bool wide = !pp && p.wide_opt > 0 && ch_gemm_wide_supported(p, epi);
    g_dispatch_count[pp ? 1 : wide ? 4 : 0].fetch_add(1, std::memory_order_relaxed);
    if (pp) return ch_gemm_bf16_pp(p, epi, s);
    if (wide) return ch_gemm_bf16_wide(p, epi, s);
    // N = 384 (the adapter bottleneck) as whole-row workgroups (experiments/gemm_rows.hip): one round of the chip instead of 2.36
    // rounds of 128x128 tiles, bit-identical -- and no faster (61 vs 59 us).  Opt-in: model option "gemm_rows", experiments build.
    if (p.rows_opt && p.small_kernel == 0 && p.M >= 128 * 128 && ch_gemm_rows_supported(p, epi)) return ch_gemm_bf16_rows(p, epi, s);
    // Small grids (batch <= 40 of ViT-B/16): every launch is one workgroup per CU walking a chain of L2 round trips; the four-stage ring
    // (gemm_r4.hip, bit-identical) keeps three K-steps in flight: -7 % per step at batch 8, -2.5 % at batch 32; slower at 51,456 rows.
    // (a non-temporal residual instance, chosen by size or forced by the option "resid_nt", exists in the two-phase kernel only and wins)
    const bool ring = (p.small_kernel == 2 || g_gemm_variant == 7 || (p.small_kernel == 0 && g_gemm_variant == 0 && p.M <= CH_RING_MAX_ROWS && !p.nt_resid)) &&
                      ch_gemm_r4_supported(p, epi);
    return ring ? ch_gemm_bf16_r4(p, epi, s) : ch_gemm_bf16_v1(p, epi, s);
}
// test tap: how many GEMMs the dispatcher has sent to the 128x128 (which = 0) / 256x256 ping-pong (which = 1) kernel, and how many
// launches ran the instance with the non-temporal residual read-modify-write (2) / the non-temporal bf16 output store (3); 4 = GEMMs sent
// to the 256x384 kernel
extern "C" int64_t ch_debug_gemm_dispatch_count(int32_t which) {
    return g_dispatch_count[which < 0 || which > 4 ? 0 : which].load(std::memory_order_relaxed);
}
