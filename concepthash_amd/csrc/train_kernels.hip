// Row / elementwise / weight-gradient kernels of the ConceptHash TRAINING step (adapters + head; reference
// trainers/coop.py:107-131 `train_one_batch` -> loss.backward() through models/layers/adapter.py:46-60,127-177 and the frozen HF
// CLIP blocks).  The big products of the backward pass run on the forward's GEMM kernels (gemm_pp.hip / gemm_bf16.hip with
// transposed weight copies, train.hip); this file holds what those kernels cannot express:
//   * hb_stats:        fp32 residual rows -> bf16 copy + the per-64-column (sum, sum of squares) partials the LN-folded GEMMs read
//   * act_fwd/act_bwd: activation on a saved bf16 pre-activation; dpre = scale * g * act'(pre)
//   * normalize:       x_hat = (x - mean) * rstd as a bf16 GEMM operand (input of the adapter down-projection's weight gradient)
//   * ln_bwd:          LayerNorm backward of one row given dy*gamma (the dgrad GEMM used W*gamma) -- added into the fp32
//                      gradient of the residual stream, which is also re-emitted as the bf16 operand of the next dgrad GEMM
//   * wgrad_tn:        dW[n][k] = sum_m A[m][n] B[m][k] over ~50k rows: both operands are row-major with the REDUCTION index slow,
//                      so both MFMA fragments come through the hardware transpose read ds_read_b64_tr_b16; split over row
//                      chunks into fp32 partial slabs, summed in chunk order by reduce_partials (deterministic, no atomics)
//   * colsum:          bias gradients, same two-stage reduction
//   * transposes:      [N, K] weights -> [K, N] bf16 (optionally scaled per input column by the LayerNorm gamma) for the dgrad GEMMs
//   * adapter_grads:   assembles the gradients of one adapter's parameters from the two weight-gradient products
#include "ch_common.h"
#include "kernels.h"
#include "gemm_epilogue.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s lds_v4s;
typedef int v2i_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char lds_char_t;
// ds_read_b64_tr_b16 at an LDS byte address, invisible to the compiler's LDS-DMA alias analysis (see wgrad_tn_kernel)
__device__ __forceinline__ v2i_t tr_read(uint32_t lds_addr) {
    v2i_t r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(lds_addr) : "memory");
    return r;
}

constexpr int MAXP = 10;  // D <= 1280, D % 128 == 0 (row kernels: one wave per row, lane holds elements (j*64 + lane)*2 + {0,1})

// (mean, rstd) of a row from the slice partials [D/64][2] written by hb_stats / the *_STATS GEMM epilogues; same formula as
// ch_epi::fold_stats_finish so that backward normalises exactly what the forward normalised
__device__ __forceinline__ void row_mean_rstd(const float *st, int D, float eps, int lane, float &mean, float &rstd) {
    const int ns = D >> 6;
    float sm = 0.f, sq = 0.f;
    if (lane < ns) {
        const ch_f32x2_t v = *(const ch_f32x2_t *)(st + 2 * lane);
        sm = v[0];
        sq = v[1];
    }
    sm = wave_sum(sm);
    sq = wave_sum(sq);
    mean = sm / (float)D;
    const float var = fmaxf(sq / (float)D - mean * mean, 0.f);
    rstd = rsqrtf(var + eps);
}

__global__ __launch_bounds__(256) void hb_stats_kernel(const float *__restrict__ H, int64_t rows, int D, bf16_t *__restrict__ hb,
                                                       float *__restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int npass = D >> 7;
    const float *x = H + row * D;
#pragma unroll
    for (int j = 0; j < MAXP; ++j)
        if (j < npass) {
            const float2 v = *(const float2 *)(x + (j * 64 + lane) * 2);
            const uint32_t u = pack_bf16x2(v.x, v.y);
            *(uint32_t *)(hb + row * D + (j * 64 + lane) * 2) = u;
            const float a = bf2f((bf16_t)(u & 0xffff)), b = bf2f((bf16_t)(u >> 16));
            float sm = a + b, sq = a * a + b * b;
            // slice 2j + (lane >> 5): sum over the 32 lanes of this half wave
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
                sm += __shfl_xor(sm, o, 64);
                sq += __shfl_xor(sq, o, 64);
            }
            if ((lane & 31) == 0) *(ch_f32x2_t *)(stats + (row * (D >> 6) + 2 * j + (lane >> 5)) * 2) = ch_f32x2_t{sm, sq};
        }
}

// ---- activations on saved pre-activations ------------------------------------------------------------------------------------
__device__ __forceinline__ float act_f(float x, int act) { return act == 0 ? ch_epi::quick_gelu_f(x) : ch_epi::gelu_erf_f(x); }
__device__ __forceinline__ float dact_f(float x, int act) { return act == 0 ? ch_epi::dquick_gelu_f(x) : ch_epi::dgelu_erf_f(x); }

template <bool BWD>
__global__ __launch_bounds__(256) void act_kernel(const bf16_t *__restrict__ g, const bf16_t *__restrict__ pre, int64_t n8, int act,
                                                  const float *__restrict__ scale_ptr, bf16_t *__restrict__ out) {
    const float scale = (BWD && scale_ptr) ? *scale_ptr : 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 p = *(const uint4 *)(pre + i * 8);
        uint4 gg = make_uint4(0, 0, 0, 0);
        if constexpr (BWD) gg = *(const uint4 *)(g + i * 8);
        const uint32_t pw[4] = {p.x, p.y, p.z, p.w}, gw[4] = {gg.x, gg.y, gg.z, gg.w};
        uint32_t ow[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x0 = bf2f((bf16_t)(pw[j] & 0xffff)), x1 = bf2f((bf16_t)(pw[j] >> 16));
            float y0, y1;
            if constexpr (BWD) {
                y0 = scale * bf2f((bf16_t)(gw[j] & 0xffff)) * dact_f(x0, act);
                y1 = scale * bf2f((bf16_t)(gw[j] >> 16)) * dact_f(x1, act);
            } else {
                y0 = act_f(x0, act);
                y1 = act_f(x1, act);
            }
            ow[j] = pack_bf16x2(y0, y1);
        }
        *(uint4 *)(out + i * 8) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
}

// ---- x_hat = (x - mean) * rstd ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void normalize_kernel(const bf16_t *__restrict__ x, const float *__restrict__ stats, int64_t rows,
                                                        int D, float eps, bf16_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float mean, rstd;
    row_mean_rstd(stats + row * (D >> 6) * 2, D, eps, lane, mean, rstd);
    const int npass = D >> 7;
#pragma unroll
    for (int j = 0; j < MAXP; ++j)
        if (j < npass) {
            const uint32_t u = *(const uint32_t *)(x + row * D + (j * 64 + lane) * 2);
            const float a = (bf2f((bf16_t)(u & 0xffff)) - mean) * rstd, b = (bf2f((bf16_t)(u >> 16)) - mean) * rstd;
            *(uint32_t *)(out + row * D + (j * 64 + lane) * 2) = pack_bf16x2(a, b);
        }
}

// ---- LayerNorm backward of a row: result = dres_in + rstd * (dyg - mean(dyg) - x_hat * mean(dyg * x_hat)) ---------------------
// dres_in: the gradient that bypasses the LayerNorm'd branch (the residual path); the result goes to dres_out (fp32, may alias
// dres_in, may be null) and / or out_b (bf16 GEMM operand, may be null)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t *__restrict__ dyg, const bf16_t *__restrict__ x,
                                                     const float *__restrict__ stats, int64_t rows, int D, float eps,
                                                     const float *dres_in, float *dres_out, bf16_t *__restrict__ out_b,
                                                     bf16_t *__restrict__ xhat_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float mean, rstd;
    row_mean_rstd(stats + row * (D >> 6) * 2, D, eps, lane, mean, rstd);
    const int npass = D >> 7;
    float2 g[MAXP], xh[MAXP];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < MAXP; ++j)
        if (j < npass) {
            const uint32_t ug = *(const uint32_t *)(dyg + row * D + (j * 64 + lane) * 2);
            const uint32_t ux = *(const uint32_t *)(x + row * D + (j * 64 + lane) * 2);
            g[j] = make_float2(bf2f((bf16_t)(ug & 0xffff)), bf2f((bf16_t)(ug >> 16)));
            xh[j] = make_float2((bf2f((bf16_t)(ux & 0xffff)) - mean) * rstd, (bf2f((bf16_t)(ux >> 16)) - mean) * rstd);
            // x_hat as a bf16 GEMM operand (the adapter's weight-gradient product consumes it): a by-product here, a separate
            // read of x otherwise (normalize_kernel)
            if (xhat_out) *(uint32_t *)(xhat_out + row * D + (j * 64 + lane) * 2) = pack_bf16x2(xh[j].x, xh[j].y);
            s1 += g[j].x + g[j].y;
            s2 += g[j].x * xh[j].x + g[j].y * xh[j].y;
        }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int j = 0; j < MAXP; ++j)
        if (j < npass) {
            float2 d = *(const float2 *)(dres_in + row * D + (j * 64 + lane) * 2);   // (non-temporal here: measured, no gain)
            d.x += rstd * (g[j].x - s1 - xh[j].x * s2);
            d.y += rstd * (g[j].y - s1 - xh[j].y * s2);
            if (dres_out) *(float2 *)(dres_out + row * D + (j * 64 + lane) * 2) = d;
            if (out_b) *(uint32_t *)(out_b + row * D + (j * 64 + lane) * 2) = pack_bf16x2(d.x, d.y);
        }
}

// ---- weight gradient: out[n][k] = sum_m A[m][n] * B[m][k] ---------------------------------------------------------------------
// Workgroup: 128 (n) x 128 (k) output tile over one chunk of rows; 4 waves = 2 x 2, wave tile 64 x 64 = acc[4][4] of
// v_mfma_f32_16x16x32_bf16.  A K-step is 32 rows: A[32][128] and B[32][128] (8 KB each) by LDS-DMA, four-stage ring.  LDS rows
// are 256 B (128 bf16); the 32-B column-tile index (16 columns) is XOR-swizzled with (row & 7) on the per-lane source address
// and on the transpose-read address: the 32 lanes served together by ds_read_b64_tr_b16 address 8 different rows (r & 7 all
// distinct) x 32 B -> all 64 banks once.
// Fragment by transpose read: lane i = 4*qq + pp of a 16-lane group (fq) addresses row 4*fq + qq, columns 4*pp..4*pp+3 of the
// 16-column tile and receives column i of rows 4*fq .. 4*fq+3; a second read 16 rows further down completes the 8 reduction
// indices of the lane.  Both operands use the same construction, so the (permuted) reduction order matches.
// The MFMA A operand is the B-matrix tile (rows i = k column), the B operand the A-matrix tile (j = n): the accumulator lane
// (n = lane & 15, fq) holds out[n][4*fq .. 4*fq+3] of the tile -> 16-byte fp32 stores into the chunk's partial slab.
constexpr int WG_TILE = 128, WG_KSTEP = 32, WG_STAGE = 2 * WG_KSTEP * WG_TILE * 2;  // 16 KB per stage (A + B)

__global__ __launch_bounds__(256, 2) void wgrad_tn_kernel(const bf16_t *__restrict__ A, int lda, const bf16_t *__restrict__ B, int ldb,
                                                          int N, int K, int steps_per_chunk, int total_steps, int nchunks,
                                                          float *__restrict__ partial) {
    __shared__ __attribute__((aligned(16))) char smem[4 * WG_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid & 1, wk = wid >> 1;
    // XCD-aware order: workgroup ids are dealt round-robin to the 8 XCDs, so ids congruent mod 8 walk the output tiles of ONE
    // row chunk back to back on one XCD -- the 18 tiles of a chunk read the same rows (each A column block 3x, each B column block
    // 6x) and find them in that XCD's L2 instead of re-fetching them (474 MB per call re-read against 118 MB of operands).
    const int tiles_k = K / WG_TILE, tiles = (N / WG_TILE) * tiles_k;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int chunk = (j / tiles) * 8 + xcd, tile = j % tiles;
    if (chunk >= nchunks) return;
    const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
    const int n0 = tn * WG_TILE, k0 = tk * WG_TILE;
    const int64_t m_begin = (int64_t)chunk * steps_per_chunk * WG_KSTEP;
    const int nsteps = max(0, min(steps_per_chunk, total_steps - chunk * steps_per_chunk));  // the last chunks may be short / empty

    // staging: a stage image is [A rows 0..31 ; B rows 0..31] x 256 B = 16 wave-instructions of 4 rows; wave w issues 4w..4w+3
    const char *gsrc[4];
    size_t gstep[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int inst = wid * 4 + j;
        const int row = (inst & 7) * 4 + (lane >> 4);  // 0..31 within A or B
        const int pc = lane & 15;                       // physical 16-B chunk of the 256-B LDS row
        const int col = (((pc >> 1) ^ (row & 7)) << 4) + ((pc & 1) << 3);
        const bool isA = inst < 8;
        const bf16_t *base = isA ? A + (m_begin + row) * (int64_t)lda + n0 + col : B + (m_begin + row) * (int64_t)ldb + k0 + col;
        gsrc[j] = (const char *)base;
        gstep[j] = (size_t)WG_KSTEP * (isA ? lda : ldb) * 2;
    }
    auto stage = [&](int buf, int st) {
        char *dst = smem + buf * WG_STAGE + wid * 4 * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(gsrc[j] + st * gstep[j]), (lds_void_t *)(dst + j * 1024), 16, 0, 0);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4, tq = fr >> 2, tp = fr & 3;
    const int trow = fq * 4 + tq, r7 = ((fq & 1) << 2) | tq;
    int aoff[4], boff[4];  // byte offsets inside a stage for sub-tile t of this wave (first 16 rows; + 16 * 256 for the second read)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        aoff[t] = trow * 256 + ((((wn * 4 + t) ^ r7)) << 5) + tp * 8;
        boff[t] = WG_KSTEP * 256 + trow * 256 + ((((wk * 4 + t) ^ r7)) << 5) + tp * 8;
    }
    union Frag {
        bf16x8 v;
        v2i_t d[2];
    };

    // four-stage ring, three K-steps of loads in flight (counted vmcnt + raw barrier, as gemm_r4.hip): with two stages and a
    // vmcnt(0) per step the kernel ran at the HBM round-trip time per 32-row step (420 TFLOP/s, ~2 workgroups per CU)
    if (nsteps > 0) stage(0, 0);
    if (nsteps > 1) stage(1, 1);
    if (nsteps > 2) stage(2, 2);
    int cur = 0;
    for (int st = 0; st < nsteps; ++st) {
        const int ahead = nsteps - 1 - st;
        if (ahead >= 2)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();   // stage st landed for every wave; every wave finished reading stage st - 1
        __builtin_amdgcn_sched_barrier(0);
        if (st + 3 < nsteps) stage((cur + 3) & 3, st + 3);
        // The transpose reads are issued as inline asm: for the builtin the compiler assumes the read may alias the LDS-DMA in
        // flight and puts `s_waitcnt vmcnt(0)` in front of it, i.e. it waits for the stages just requested -- which made the
        // two-stage version of this kernel run at one HBM round trip per K-step.  The barrier above is what orders this stage's
        // DMA before these reads; their completion is awaited by hand (lgkmcnt) before the first MFMA.
        const uint32_t sbase = (uint32_t)(size_t)(lds_char_t *)(smem) + cur * WG_STAGE;
        Frag af[4], bfm[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            af[t].d[0] = tr_read(sbase + aoff[t]);
            af[t].d[1] = tr_read(sbase + aoff[t] + 16 * 256);
            bfm[t].d[0] = tr_read(sbase + boff[t]);
            bfm[t].d[1] = tr_read(sbase + boff[t] + 16 * 256);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[kt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfm[kt].v, af[nt].v, acc[kt][nt], 0, 0, 0);
        cur = (cur + 1) & 3;
    }
    float *slab = partial + (size_t)chunk * N * K;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + wn * 64 + nt * 16 + fr, k = k0 + wk * 64 + kt * 16 + fq * 4;
            *(f32x4 *)(slab + (size_t)n * K + k) = acc[kt][nt];
        }
}

// out[i] = sum over chunks of partial[c][i]; n4 = elements / 4.  256 threads = 32 float4 columns x 8 chunk lanes: lane l sums
// chunks l, l + 8, ... in order, the 8 lane sums are added in lane order -- a fixed association, so run-to-run identical -- and the
// serial chain is chunks / 8 loads instead of chunks (the one-block column-sum reduction was a 15 us latency chain of 101 loads)
__device__ __forceinline__ void reduce_partials_body(const float *__restrict__ partial, int nchunks, int64_t n4, float *__restrict__ out,
                                                     int bx) {
    __shared__ f32x4 red[8][32];
    const int col = threadIdx.x & 31, cl = threadIdx.x >> 5;
    const int64_t i = (int64_t)bx * 32 + col;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n4)
        for (int c = cl; c < nchunks; c += 8) s += *(const f32x4 *)(partial + ((size_t)c * n4 + i) * 4);
    red[cl][col] = s;
    __syncthreads();
    if (cl == 0 && i < n4) {
        f32x4 t = red[0][col];
#pragma unroll
        for (int j = 1; j < 8; ++j) t += red[j][col];
        *(f32x4 *)(out + i * 4) = t;
    }
}
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float *__restrict__ partial, int nchunks, int64_t n4,
                                                              float *__restrict__ out) {
    reduce_partials_body(partial, nchunks, n4, out, blockIdx.x);
}
// up to four reductions in ONE launch (an adapter's two weight-gradient products and two column sums: four 5 us launches otherwise,
// 96 per step -- at batch 32 the step is a chain of ~600 dependent launches and each tiny one costs its start-up latency)
struct ReduceJobsArg {
    ChReduceJob job[4];
    int first_block[5];
};
__global__ __launch_bounds__(256) void reduce_partials_multi_kernel(ReduceJobsArg a) {
    int j = 0;
    while (j < 3 && (int)blockIdx.x >= a.first_block[j + 1]) ++j;
    reduce_partials_body(a.job[j].partial, a.job[j].nchunks, a.job[j].n4, a.job[j].out, blockIdx.x - a.first_block[j]);
}

// column sums of a [rows, N] matrix over a chunk of rows -> partial[chunk][N].  256 threads = 32 column groups x 8 row lanes; a thread
// owns 16 bytes of a row (4 fp32 / 8 bf16 columns) and keeps four independent row loads in flight; the 8 row lanes are added in
// lane order through LDS (fixed association).  The first version (4-byte / 8-byte accesses, one load in flight) ran at 1.3-2.8 TB/s.
template <bool F32>
__global__ __launch_bounds__(256) void colsum_kernel(const void *__restrict__ Av, int lda, int64_t rows, int N, int chunk_rows,
                                                     float *__restrict__ partial) {
    constexpr int CPT = F32 ? 4 : 8;                 // columns per thread
    __shared__ float red[8][32][CPT];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int col = (blockIdx.x * 32 + cg) * CPT;
    const int64_t r0 = (int64_t)blockIdx.y * chunk_rows, r1 = min(rows, r0 + chunk_rows);
    float acc[4][CPT];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < CPT; ++c) acc[u][c] = 0.f;
    auto add = [&](int u, int64_t r) {
        if constexpr (F32) {
            const f32x4 v = *(const f32x4 *)((const float *)Av + r * lda + col);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[u][c] += v[c];
        } else {
            const uint4 v = *(const uint4 *)((const bf16_t *)Av + r * lda + col);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[u][2 * c] += bf2f((bf16_t)(w[c] & 0xffff));
                acc[u][2 * c + 1] += bf2f((bf16_t)(w[c] >> 16));
            }
        }
    };
    if (col < N) {
        int64_t r = r0 + rl;
        for (; r + 24 < r1; r += 32) {
            add(0, r);
            add(1, r + 8);
            add(2, r + 16);
            add(3, r + 24);
        }
        for (; r < r1; r += 8) add(0, r);
    }
#pragma unroll
    for (int c = 0; c < CPT; ++c) red[rl][cg][c] = (acc[0][c] + acc[1][c]) + (acc[2][c] + acc[3][c]);
    __syncthreads();
    if (rl == 0 && col < N) {
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            float t = red[0][cg][c];
#pragma unroll
            for (int j = 1; j < 8; ++j) t += red[j][cg][c];
            partial[(size_t)blockIdx.y * N + col + c] = t;
        }
    }
}

// ---- transposes ----------------------------------------------------------------------------------------------------------------
// dst[c][r] = bf16(src[r][c] * (colscale ? colscale[c] : 1)), src [R, C] (ld ld_src), dst [C, ld_dst]; 32 x 32 tiles through LDS
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T *__restrict__ src, int R, int C, int ld_src,
                                                        const float *__restrict__ colscale, bf16_t *__restrict__ dst, int ld_dst) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        float v = 0.f;
        if (r < R && c < C) {
            if constexpr (sizeof(T) == 2)
                v = bf2f((bf16_t)src[(size_t)r * ld_src + c]);
            else
                v = (float)src[(size_t)r * ld_src + c];
            if (colscale) v *= colscale[c];
        }
        tile[j][tx] = v;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (c < C && r < R) dst[(size_t)c * ld_dst + r] = f2bf(tile[tx][j]);
    }
}

// ---- working copies of ALL adapters from the parameter arena in four launches (blockIdx.y = adapter): the per-adapter form
// (fold_ln + convert + two transposes, 96 launches for B/16) was 0.5 ms of every optimizer step.
// Arena block of one adapter: [ln_w D][ln_b D][down_w b*D][down_b b][up_w D*b][up_b D][scale 1], `stride` floats apart.
// (1) LayerNorm fold of the down projection: Wdf = bf16(W * gamma) [bpad, D] (rows >= b zero), c[n] = sum_k Wdf[n][k], d[n] = b[n] + sum_k W[n][k] beta[k]
__global__ __launch_bounds__(256) void refresh_fold_kernel(const float *__restrict__ P, int64_t stride, int D, int b, int bpad,
                                                           bf16_t *__restrict__ Wdf, float *__restrict__ c, float *__restrict__ d) {
    const float *base = P + blockIdx.y * stride, *gamma = base, *beta = base + D, *W = base + 2 * D, *bd = W + (size_t)b * D;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= bpad) return;
    bf16_t *out = Wdf + ((size_t)blockIdx.y * bpad + n) * D;
    float cs = 0.f, ds = 0.f;
    for (int k = lane; k < D; k += 64) {
        const float w = n < b ? W[(size_t)n * D + k] : 0.f;
        const bf16_t wb = f2bf(w * gamma[k]);
        out[k] = wb;
        cs += bf2f(wb);
        ds += beta[k] * w;
    }
    cs = wave_sum(cs);
    ds = wave_sum(ds);
    if (lane == 0) {
        c[(size_t)blockIdx.y * bpad + n] = cs;
        d[(size_t)blockIdx.y * bpad + n] = ds + (n < b ? bd[n] : 0.f);
    }
}
// (2) up_w [D, b] fp32 -> bf16 [D, bpad] (columns >= b zero)
__global__ __launch_bounds__(256) void refresh_convert_kernel(const float *__restrict__ P, int64_t stride, int D, int b, int bpad,
                                                              bf16_t *__restrict__ out) {
    const float *up_w = P + blockIdx.y * stride + 2 * D + (size_t)b * D + b;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= D * bpad) return;
    const int r = i / bpad, cc = i - r * bpad;
    out[(size_t)blockIdx.y * D * bpad + i] = cc < b ? f2bf(up_w[(size_t)r * b + cc]) : (bf16_t)0;
}
// (3, 4) transposes through LDS: which = 0: up_w [D, b] -> up_wT [bpad, D] (rows >= b stay zero); which = 1: down_w [b, D] * gamma ->
// down_wgT [D, bpad] (columns >= b stay zero)
__global__ __launch_bounds__(256) void refresh_transpose_kernel(const float *__restrict__ P, int64_t stride, int D, int b, int bpad, int which,
                                                                bf16_t *__restrict__ dst_all) {
    __shared__ float tile[32][33];
    const float *base = P + blockIdx.z * stride;
    const float *src = which == 0 ? base + 2 * D + (size_t)b * D + b : base + 2 * D;
    const float *colscale = which == 0 ? nullptr : base;
    const int R = which == 0 ? D : b, C = which == 0 ? b : D, ld_src = C, ld_dst = which == 0 ? D : bpad;
    bf16_t *dst = dst_all + (size_t)blockIdx.z * bpad * D;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, cc = c0 + tx;
        float v = 0.f;
        if (r < R && cc < C) {
            v = src[(size_t)r * ld_src + cc];
            if (colscale) v *= colscale[cc];
        }
        tile[j][tx] = v;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int cc = c0 + j, r = r0 + tx;
        if (cc < C && r < R) dst[(size_t)cc * ld_dst + r] = f2bf(tile[tx][j]);
    }
}

// ---- gradients of one adapter's parameters (models/layers/adapter.py:46-60) from the two weight-gradient products -------------
//   G [D, bpad]  = dH^T g            (dH: gradient of the block output, g = GELU(down(LN(a))); unscaled)
//   cu [D]       = column sums of dH
//   T [bpad, D]  = dpre^T x_hat      (dpre: gradient of the down-projection output, x_hat: normalised adapter input)
//   cd [bpad]    = column sums of dpre
// up:    dW_up = s G, db_up = s cu, ds = <G, W_up> + <cu, b_up>      (out = s (g W_up^T + b_up))
// down:  dW_dn = T * gamma + cd (x) beta, db_dn = cd
// LN:    dgamma[k] = sum_j T[j][k] W_dn[j][k], dbeta[k] = sum_j cd[j] W_dn[j][k]
// Parameter / gradient arena layout of one adapter (fp32): [ln_w D][ln_b D][down_w b*D][down_b b][up_w D*b][up_b D][scale 1]
// kernel 1 (grid of AG_BLOCKS blocks): the elementwise parts + this block's partial of <G, W_up> into part[blockIdx.x]
constexpr int AG_BLOCKS = 256;
__global__ __launch_bounds__(256) void adapter_grads_elem_kernel(const float *__restrict__ G, const float *__restrict__ cu,
                                                                 const float *__restrict__ T, const float *__restrict__ cd,
                                                                 const float *__restrict__ P, int D, int b, int bpad, float *__restrict__ gr,
                                                                 float *__restrict__ part, int64_t pstride) {
    // blockIdx.y: the adapter (slot arrays: G / T stride D * bpad, cu D, cd bpad, parameters and gradients pstride, partials AG_BLOCKS)
    G += (size_t)blockIdx.y * D * bpad, T += (size_t)blockIdx.y * bpad * D, cu += (size_t)blockIdx.y * D, cd += (size_t)blockIdx.y * bpad;
    P += blockIdx.y * pstride, gr += blockIdx.y * pstride, part += blockIdx.y * AG_BLOCKS;
    const float *ln_w = P, *ln_b = P + D, *down_w = P + 2 * D, *up_w = down_w + (size_t)b * D + b, *up_b = up_w + (size_t)D * b;
    const float s = up_b[D];
    float *g_down_w = gr + 2 * D, *g_down_b = g_down_w + (size_t)b * D, *g_up_w = g_down_b + b, *g_up_b = g_up_w + (size_t)D * b;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    for (int i = tid; i < b * D; i += nth) {
        const int j = i / D, k = i - j * D;
        g_down_w[i] = T[(size_t)j * D + k] * ln_w[k] + cd[j] * ln_b[k];
    }
    for (int i = tid; i < b; i += nth) g_down_b[i] = cd[i];
    float a = 0.f;
    for (int i = tid; i < D * b; i += nth) {
        const int n = i / b, j = i - n * b;
        const float g = G[(size_t)n * bpad + j];
        g_up_w[i] = s * g;
        a += g * up_w[i];
    }
    for (int i = tid; i < D; i += nth) {
        g_up_b[i] = s * cu[i];
        a += cu[i] * up_b[i];
    }
    __shared__ float red[256];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
// kernel 2: block 0 sums the AG_BLOCKS partials in a fixed order -> d(scale); blocks 1.. : 16 columns k each, 64 row lanes over j
// (6 dependent loads per thread for b = 384; with 64 columns x 16 lanes it was a 24-load chain on 13 blocks):
// dgamma[k] = sum_j T[j][k] W_dn[j][k], dbeta[k] = sum_j cd[j] W_dn[j][k]
__global__ __launch_bounds__(1024) void adapter_grads_red_kernel(const float *__restrict__ T, const float *__restrict__ cd,
                                                                 const float *__restrict__ P, int D, int b, int bpad, float *__restrict__ gr,
                                                                 const float *__restrict__ part, int64_t pstride) {
    T += (size_t)blockIdx.y * bpad * D, cd += (size_t)blockIdx.y * bpad;
    P += blockIdx.y * pstride, gr += blockIdx.y * pstride, part += blockIdx.y * AG_BLOCKS;
    __shared__ float ra[64][17], rc[64][17];
    if (blockIdx.x == 0) {
        __shared__ float red[256];
        if (threadIdx.x < 256) red[threadIdx.x] = threadIdx.x < AG_BLOCKS ? part[threadIdx.x] : 0.f;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) gr[2 * D + (size_t)b * D + b + (size_t)D * b + D] = red[0];
        return;
    }
    const float *down_w = P + 2 * D;
    const int kl = threadIdx.x & 15, jl = threadIdx.x >> 4;
    const int k = (blockIdx.x - 1) * 16 + kl;
    float a = 0.f, c = 0.f;
    if (k < D)
        for (int j = jl; j < b; j += 64) {
            const float w = down_w[(size_t)j * D + k];
            a += T[(size_t)j * D + k] * w;
            c += cd[j] * w;
        }
    ra[jl][kl] = a;
    rc[jl][kl] = c;
    __syncthreads();
    if (jl == 0 && k < D) {
        float sa = 0.f, sc = 0.f;
        for (int j = 0; j < 64; ++j) {   // fixed order
            sa += ra[j][kl];
            sc += rc[j][kl];
        }
        gr[k] = sa;
        gr[D + k] = sc;
    }
}

// torch.optim.SGD (maximize False) over one flat fp32 array: d = g + wd * p; buf = d (first step) or momentum * buf + (1 - dampening) * d;
// p -= lr * (nesterov ? d + momentum * buf : buf).  One launch over the 14 M-parameter adapter arena instead of the per-tensor
// foreach kernels over its 168 views.
__global__ __launch_bounds__(256) void sgd_step_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ buf, int64_t n4,
                                                       float lr, float momentum, float wd, float dampening, int nesterov, int first) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 pv = *(const f32x4 *)(p + i * 4);
        f32x4 d = *(const f32x4 *)(g + i * 4) + pv * wd;
        f32x4 upd = d;
        if (momentum != 0.f) {
            f32x4 b = first ? d : *(const f32x4 *)(buf + i * 4) * momentum + d * (1.0f - dampening);
            *(f32x4 *)(buf + i * 4) = b;
            upd = nesterov ? d + b * momentum : b;
        }
        *(f32x4 *)(p + i * 4) = pv - upd * lr;
    }
}

// rows of the concept tokens: out[q][:] = sum_b dH[b*ntok + ntok - Q + q][:]   (one block per (q, 256-column slab))
__global__ __launch_bounds__(256) void concept_rows_sum_kernel(const float *__restrict__ dH, int B, int ntok, int Q, int D,
                                                               float *__restrict__ out) {
    const int q = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= D) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dH[((size_t)b * ntok + ntok - Q + q) * D + k];
    out[(size_t)q * D + k] = s;
}
// dH = 0 except the concept-token rows, which take d_hash_features [B, Q, D]
__global__ __launch_bounds__(256) void scatter_concept_rows_kernel(const float *__restrict__ dhf, int B, int ntok, int Q, int D,
                                                                   float *__restrict__ dH, bf16_t *__restrict__ dHb) {
    const int64_t row = blockIdx.x;
    const int t = (int)(row % ntok), bimg = (int)(row / ntok);
    const bool con = t >= ntok - Q;
    for (int k = threadIdx.x; k < D; k += 256) {
        const float v = con ? dhf[((size_t)bimg * Q + (t - (ntok - Q))) * D + k] : 0.f;
        dH[row * D + k] = v;
        dHb[row * D + k] = f2bf(v);
    }
}
// compact head rows [B * (1 + Q), D] (slot 0 = CLS, slots 1.. = the concept tokens) -> full token rows [B * ntok, D], zero elsewhere
template <typename T>
__global__ __launch_bounds__(256) void expand_head_rows_kernel(const T *__restrict__ src, int ntok, int Q, int D, T *__restrict__ dst) {
    const int64_t row = blockIdx.x;
    const int t = (int)(row % ntok), bimg = (int)(row / ntok);
    const int slot = t == 0 ? 0 : (t >= ntok - Q ? 1 + t - (ntok - Q) : -1);
    for (int k = threadIdx.x; k < D; k += 256) dst[row * D + k] = slot >= 0 ? src[((size_t)bimg * (1 + Q) + slot) * D + k] : (T)0;
}
// LayerNorm backward of fp32 rows (the pre_layrnorm of the concept-token rows): dx = rstd * (dy*g - mean(dy*g) - x_hat * mean(dy*g*x_hat))
__global__ __launch_bounds__(256) void small_ln_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                           const float *__restrict__ gamma, int D, float eps, float *__restrict__ dx) {
    __shared__ float red[3][256];
    const int r = blockIdx.x;
    const float *xr = x + (size_t)r * D, *dyr = dy + (size_t)r * D;
    float s = 0.f;
    for (int k = threadIdx.x; k < D; k += 256) s += xr[k];
    red[0][threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[0][threadIdx.x] += red[0][threadIdx.x + o];
        __syncthreads();
    }
    const float mean = red[0][0] / D;
    __syncthreads();
    float q = 0.f;
    for (int k = threadIdx.x; k < D; k += 256) q += (xr[k] - mean) * (xr[k] - mean);
    red[0][threadIdx.x] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[0][threadIdx.x] += red[0][threadIdx.x + o];
        __syncthreads();
    }
    const float rstd = rsqrtf(red[0][0] / D + eps);
    __syncthreads();
    float s1 = 0.f, s2 = 0.f;
    for (int k = threadIdx.x; k < D; k += 256) {
        const float g = dyr[k] * gamma[k];
        s1 += g;
        s2 += g * (xr[k] - mean) * rstd;
    }
    red[1][threadIdx.x] = s1;
    red[2][threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
            red[2][threadIdx.x] += red[2][threadIdx.x + o];
        }
        __syncthreads();
    }
    s1 = red[1][0] / D;
    s2 = red[2][0] / D;
    for (int k = threadIdx.x; k < D; k += 256)
        dx[(size_t)r * D + k] = rstd * (dyr[k] * gamma[k] - s1 - (xr[k] - mean) * rstd * s2);
}
// out[b][q][:] = H[b*ntok + ntok - Q + q][:]
__global__ __launch_bounds__(256) void gather_concept_rows_kernel(const float *__restrict__ H, int ntok, int Q, int D, float *__restrict__ out) {
    const int r = blockIdx.x, bimg = r / Q, q = r - bimg * Q;
    for (int k = threadIdx.x; k < D; k += 256) out[(size_t)r * D + k] = H[((size_t)bimg * ntok + ntok - Q + q) * D + k];
}

}  // namespace

int ch_hb_stats(const float *H, int64_t rows, int D, bf16_t *hb, float *stats, hipStream_t s) {
    CH_REQUIRE(D % 128 == 0 && D <= 128 * MAXP, "hb_stats: D must be a multiple of 128, <= 1280");
    hipLaunchKernelGGL(hb_stats_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, s, H, rows, D, hb, stats);
    CH_LAUNCH_CHECK();
    return 0;
}
int ch_act_fwd(const bf16_t *pre, int64_t n, int act, bf16_t *out, hipStream_t s) {
    CH_REQUIRE(n % 8 == 0, "act_fwd: element count must be a multiple of 8");
    const int64_t n8 = n / 8;
    hipLaunchKernelGGL(act_kernel<false>, dim3((unsigned)std::min<int64_t>(ceil_div64(n8, 256), 8192)), dim3(256), 0, s,
                       (const bf16_t *)nullptr, pre, n8, act, (const float *)nullptr, out);
    CH_LAUNCH_CHECK();
    return 0;
}
int ch_act_bwd(const bf16_t *g, const bf16_t *pre, int64_t n, int act, const float *scale_ptr, bf16_t *out, hipStream_t s) {
    CH_REQUIRE(n % 8 == 0, "act_bwd: element count must be a multiple of 8");
    const int64_t n8 = n / 8;
    hipLaunchKernelGGL(act_kernel<true>, dim3((unsigned)std::min<int64_t>(ceil_div64(n8, 256), 8192)), dim3(256), 0, s, g, pre, n8, act,
                       scale_ptr, out);
    CH_LAUNCH_CHECK();
    return 0;
}
int ch_normalize_bf16(const bf16_t *x, const float *stats, int64_t rows, int D, float eps, bf16_t *out, hipStream_t s) {
    CH_REQUIRE(D % 128 == 0 && D <= 128 * MAXP, "normalize: D must be a multiple of 128, <= 1280");
    hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, s, x, stats, rows, D, eps, out);
    CH_LAUNCH_CHECK();
    return 0;
}
int ch_ln_bwd(const bf16_t *dyg, const bf16_t *x, const float *stats, int64_t rows, int D, float eps, const float *dres_in,
              float *dres_out, bf16_t *out_b, hipStream_t s, bf16_t *xhat_out) {
    CH_REQUIRE(D % 128 == 0 && D <= 128 * MAXP, "ln_bwd: D must be a multiple of 128, <= 1280");
    hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, s, dyg, x, stats, rows, D, eps, dres_in, dres_out,
                       out_b, xhat_out);
    CH_LAUNCH_CHECK();
    return 0;
}

// number of row chunks of the two-stage reductions and the workspace they need (floats)
int ch_wgrad_chunks(int64_t rows, int N, int K) {
    const int tiles = (N / WG_TILE) * (K / WG_TILE);
    // ONE round of workgroups: the kernel pins chunk c to XCD c % 8 (so that a chunk's tiles share that XCD's L2), and an XCD has
    // 32 CUs x 2 resident workgroups = 64 slots -> at most floor(64 / tiles) chunks per XCD.  (The first version took 512 / tiles
    // chunks over the whole chip: 28 chunks of 18 tiles put 72 workgroups on XCDs 0-3 -- 8 of them waited for a second round and
    // the launch took two workgroup lifetimes, 56 us instead of 32.)
    int chunks = 8 * std::max(1, 64 / std::max(tiles, 1));
    const int64_t steps = ceil_div64(rows, WG_KSTEP);
    // at least 16 K-steps (512 rows) per chunk: below that the fp32 partial slabs (chunks * N * K * 4 bytes, written and re-read)
    // cost more than the idle CUs
    chunks = (int)std::max<int64_t>(1, std::min<int64_t>(chunks, steps / 16));
    return chunks;
}
size_t ch_wgrad_ws_floats(int64_t rows, int N, int K) { return (size_t)ch_wgrad_chunks(rows, N, K) * N * K; }

int ch_wgrad_tn(bf16_t *A, int lda, const bf16_t *B, int ldb, int64_t rows, int64_t rows_alloc, int N, int K, float *out,
                float *ws, hipStream_t s, int *chunks_out) {
    CH_REQUIRE(N % WG_TILE == 0 && K % WG_TILE == 0, "wgrad: N and K must be multiples of 128");
    CH_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, "wgrad: leading dimensions must be multiples of 8");
    const int chunks = ch_wgrad_chunks(rows, N, K);
    const int total_steps = (int)ceil_div64(rows, WG_KSTEP);
    const int spc = (total_steps + chunks - 1) / chunks;
    // rows past `rows` are read up to the next multiple of 32: A's padding rows are zeroed here (they may hold a LARGER earlier
    // batch's values: the buffers are reused across calls), B's only need to be finite, which any earlier contents are
    CH_REQUIRE((int64_t)total_steps * WG_KSTEP <= rows_alloc, "wgrad: operands must be allocated to a multiple of 32 rows");
    if ((int64_t)total_steps * WG_KSTEP > rows)
        CH_CHECK_HIP(hipMemsetAsync(A + rows * lda, 0, sizeof(bf16_t) * ((int64_t)total_steps * WG_KSTEP - rows) * lda, s));
    const int tiles = (N / WG_TILE) * (K / WG_TILE);
    hipLaunchKernelGGL(wgrad_tn_kernel, dim3(8 * ((chunks + 7) / 8) * tiles), dim3(256), 0, s, A, lda, B, ldb, N, K, spc, total_steps, chunks,
                       ws);
    CH_LAUNCH_CHECK();
    if (chunks_out) {   // the caller reduces the slabs itself (ch_reduce_partials_multi)
        *chunks_out = chunks;
        return 0;
    }
    const int64_t n4 = (int64_t)N * K / 4;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)ceil_div64(n4, 32)), dim3(256), 0, s, ws, chunks, n4, out);
    CH_LAUNCH_CHECK();
    return 0;
}
int ch_colsum(const void *A, int is_f32, int lda, int64_t rows, int N, float *out, float *ws, hipStream_t s, int *chunks_out) {
    CH_REQUIRE(N % 8 == 0 && lda % 8 == 0, "colsum: N and the leading dimension must be multiples of 8");
    const int chunks = (int)std::min<int64_t>(128, ceil_div64(rows, 256));
    const int chunk_rows = (int)ceil_div64(rows, chunks);
    if (is_f32)
        hipLaunchKernelGGL(colsum_kernel<true>, dim3((N / 4 + 31) / 32, chunks), dim3(256), 0, s, A, lda, rows, N, chunk_rows, ws);
    else
        hipLaunchKernelGGL(colsum_kernel<false>, dim3((N / 8 + 31) / 32, chunks), dim3(256), 0, s, A, lda, rows, N, chunk_rows, ws);
    CH_LAUNCH_CHECK();
    if (chunks_out) {
        *chunks_out = chunks;
        return 0;
    }
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)ceil_div64(N / 4, 32)), dim3(256), 0, s, ws, chunks, (int64_t)N / 4, out);
    CH_LAUNCH_CHECK();
    return 0;
}
size_t ch_colsum_ws_floats(int N) { return (size_t)128 * N; }

int ch_transpose_f32_to_bf16(const float *src, int R, int C, int ld_src, const float *colscale, bf16_t *dst, int ld_dst, hipStream_t s) {
    hipLaunchKernelGGL(transpose_kernel<float>, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, s, src, R, C, ld_src, colscale, dst, ld_dst);
    CH_LAUNCH_CHECK();
    return 0;
}
int ch_transpose_bf16(const bf16_t *src, int R, int C, int ld_src, bf16_t *dst, int ld_dst, hipStream_t s) {
    hipLaunchKernelGGL(transpose_kernel<bf16_t>, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, s, src, R, C, ld_src,
                       (const float *)nullptr, dst, ld_dst);
    CH_LAUNCH_CHECK();
    return 0;
}
// working copies of `nad` adapters (contiguous arrays, one slot per adapter): down_wf [bpad, D], fold_c / fold_d [bpad], up_w [D, bpad],
// up_wT [bpad, D], down_wgT [D, bpad]
int ch_adapter_refresh(const float *params, int64_t stride, int nad, int D, int b, int bpad, bf16_t *down_wf, float *fold_c, float *fold_d,
                       bf16_t *up_w, bf16_t *up_wT, bf16_t *down_wgT, hipStream_t s) {
    hipLaunchKernelGGL(refresh_fold_kernel, dim3((bpad + 3) / 4, nad), dim3(256), 0, s, params, stride, D, b, bpad, down_wf, fold_c, fold_d);
    CH_LAUNCH_CHECK();
    hipLaunchKernelGGL(refresh_convert_kernel, dim3((D * bpad + 255) / 256, nad), dim3(256), 0, s, params, stride, D, b, bpad, up_w);
    CH_LAUNCH_CHECK();
    hipLaunchKernelGGL(refresh_transpose_kernel, dim3((b + 31) / 32, (D + 31) / 32, nad), dim3(256), 0, s, params, stride, D, b, bpad, 0, up_wT);
    CH_LAUNCH_CHECK();
    hipLaunchKernelGGL(refresh_transpose_kernel, dim3((D + 31) / 32, (b + 31) / 32, nad), dim3(256), 0, s, params, stride, D, b, bpad, 1, down_wgT);
    CH_LAUNCH_CHECK();
    return 0;
}
int ch_adapter_grads(const float *G, const float *cu, const float *T, const float *cd, const float *params, int D, int b, int bpad,
                     float *grads, float *ws, hipStream_t s, int nad, int64_t stride) {
    hipLaunchKernelGGL(adapter_grads_elem_kernel, dim3(AG_BLOCKS, nad), dim3(256), 0, s, G, cu, T, cd, params, D, b, bpad, grads, ws, stride);
    CH_LAUNCH_CHECK();
    hipLaunchKernelGGL(adapter_grads_red_kernel, dim3(1 + (D + 15) / 16, nad), dim3(1024), 0, s, T, cd, params, D, b, bpad, grads, ws, stride);
    CH_LAUNCH_CHECK();
    return 0;
}
int ch_reduce_partials_multi(const ChReduceJob *jobs, int njobs, hipStream_t s) {
    CH_REQUIRE(njobs >= 1 && njobs <= 4, "reduce_partials_multi: 1..4 jobs");
    ReduceJobsArg a{};
    int blocks = 0;
    for (int j = 0; j < 4; ++j) {
        a.first_block[j] = blocks;
        if (j < njobs) {
            a.job[j] = jobs[j];
            blocks += (int)ceil_div64(jobs[j].n4, 32);
        }
    }
    a.first_block[4] = blocks;
    for (int j = njobs; j < 4; ++j) a.first_block[j] = blocks;
    hipLaunchKernelGGL(reduce_partials_multi_kernel, dim3(blocks), dim3(256), 0, s, a);
    CH_LAUNCH_CHECK();
    return 0;
}
int ch_sgd_step_launch(float *p, const float *g, float *buf, int64_t n, float lr, float momentum, float wd, float dampening, int nesterov,
                       int first, hipStream_t s) {
    CH_REQUIRE(n % 4 == 0, "sgd_step: element count must be a multiple of 4 (pad the arena)");
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(sgd_step_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64(n4, 256), 4096)), dim3(256), 0, s, p, g, buf, n4, lr,
                       momentum, wd, dampening, nesterov, first);
    CH_LAUNCH_CHECK();
    return 0;
}
int ch_concept_rows_sum(const float *dH, int B, int ntok, int Q, int D, float *out, hipStream_t s) {
    hipLaunchKernelGGL(concept_rows_sum_kernel, dim3((D + 255) / 256, Q), dim3(256), 0, s, dH, B, ntok, Q, D, out);
    CH_LAUNCH_CHECK();
    return 0;
}
int ch_scatter_concept_rows(const float *dhf, int B, int ntok, int Q, int D, float *dH, bf16_t *dHb, hipStream_t s) {
    hipLaunchKernelGGL(scatter_concept_rows_kernel, dim3((unsigned)((int64_t)B * ntok)), dim3(256), 0, s, dhf, B, ntok, Q, D, dH, dHb);
    CH_LAUNCH_CHECK();
    return 0;
}
int ch_expand_head_rows(const void *src, int is_f32, int B, int ntok, int Q, int D, void *dst, hipStream_t s) {
    if (is_f32)
        hipLaunchKernelGGL(expand_head_rows_kernel<float>, dim3((unsigned)((int64_t)B * ntok)), dim3(256), 0, s, (const float *)src, ntok, Q, D, (float *)dst);
    else
        hipLaunchKernelGGL(expand_head_rows_kernel<bf16_t>, dim3((unsigned)((int64_t)B * ntok)), dim3(256), 0, s, (const bf16_t *)src, ntok, Q, D,
                           (bf16_t *)dst);
    CH_LAUNCH_CHECK();
    return 0;
}
int ch_small_ln_bwd(const float *dy, const float *x, const float *gamma, int rows, int D, float eps, float *dx, hipStream_t s) {
    hipLaunchKernelGGL(small_ln_bwd_kernel, dim3(rows), dim3(256), 0, s, dy, x, gamma, D, eps, dx);
    CH_LAUNCH_CHECK();
    return 0;
}
int ch_gather_concept_rows(const float *H, int B, int ntok, int Q, int D, float *out, hipStream_t s) {
    hipLaunchKernelGGL(gather_concept_rows_kernel, dim3(B * Q), dim3(256), 0, s, H, ntok, Q, D, out);
    CH_LAUNCH_CHECK();
    return 0;
}
