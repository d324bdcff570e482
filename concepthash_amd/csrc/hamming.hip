// Packed Hamming retrieval kernels (gfx950): XOR + popcount distance, exact top-k, and the two passes of mAP.
//
// Replaces the un-vendored utils.hashing.{calculate_mAP, calculate_pr_curve, get_hamm_dist} (call sites
// experiments/test_hashing.py:106-119,153-162; trainers/orthohash.py:362; in-repo twin get_hd trainers/orthohash.py:263-264,
// which builds a float (Qn,G) matrix with a matmul and argsorts it).  Definition: SURVEY.md section 8c / oracle/hamming_oracle.c.
//
// Work layout for all three scan kernels: one LANE per QUERY (its code words and its selection state live in that
// lane's registers / LDS column), the GALLERY segment is walked sequentially and is wave-uniform, so gallery words and
// labels arrive through the scalar data path (s_load_dwordx{4,8,16}) and every XOR uses an SGPR operand.  Nothing of
// size Qn x G is ever written.  grid = (query tiles, gallery segments) so small galleries still fill the chip.
//
//  * top-k:   per-lane sorted list of the KREG smallest keys, key = dist << 23 | row-in-segment (unique, so "k smallest
//             keys" == ascending (distance, gallery index)).  A wave-uniform branch skips the insertion network unless
//             some lane beats its current threshold; the network itself is branch free (min/max chain).
//  * hist:    per-lane per-distance counters in LDS, column = lane -> conflict-free ds_add; count and relevant count
//             share one 32-bit word (16 + 16 bits, segments <= 65535 rows).
//  * AP:      the same LDS counters, read-modify-write with return: the returned value is the row's position inside its
//             (query, distance) bucket in gallery order, which with the prefix "base" gives its exact global rank and
//             relevant-rank; AP numerators are accumulated in 2^-32 fixed point (integer, order independent).
#include <cstdlib>
#include <type_traits>

#include "../../include/concepthash_hip.h"
#include "../../include/concepthash_hip_debug.h"
#include "ch_common.h"
#include "kernels.h"

// test tap (include/concepthash_hip_debug.h): 1 = the mAP passes take the gallery through scalar loads instead of VMEM blocks + DPP
static std::atomic<int> g_scalar_loads{0};
extern "C" void ch_debug_set_hamming_scalar_loads(int32_t on) { g_scalar_loads.store(on != 0, std::memory_order_relaxed); }

namespace {

constexpr int KEY_SHIFT = 23;
constexpr uint32_t KEY_MASK = (1u << KEY_SHIFT) - 1;

// The scan is VALU-issue bound, so its inner loop is written down to the instruction:
// v_bcnt_u32_b32 d, x, acc = popcount(x) + acc: chaining the accumulate operand keeps a distance at 2 instructions per 32-bit
// word (left to itself the compiler re-associates into separate counts + v_add3: one more instruction per row);
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
    uint32_t d;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(acc));
    return d;
}
// Two words of a distance against gallery words held by lane K of the lane's own row of 16 lanes: the broadcast is the DPP operand
// of the xor (`row_newbcast:K`), so it costs no instruction.  Written as one asm block because of the DPP read hazard (a VGPR
// written by a VALU instruction must not be read by a DPP instruction within the next two): the temporaries are distinct
// early-clobber registers and each is written >= 2 instructions before the block ends, so back-to-back blocks are safe whatever
// registers the allocator picks, without the s_nop the compiler puts between its own DPP instructions.  g0 / g1 are only ever
// written by loads.
template <int K>
__device__ __forceinline__ uint32_t xor_bcnt2_row_bcast(uint32_t g0, uint32_t g1, uint32_t q0, uint32_t q1, uint32_t acc) {
    uint32_t t0, t1, d;
    asm("v_xor_b32_dpp %0, %3, %5 row_newbcast:%8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_xor_b32_dpp %1, %4, %6 row_newbcast:%8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_bcnt_u32_b32 %2, %0, %7\n\t"
        "v_bcnt_u32_b32 %2, %1, %2"
        : "=&v"(t0), "=&v"(t1), "=&v"(d)
        : "v"(g0), "v"(g1), "v"(q0), "v"(q1), "v"(acc), "n"(K));
    return d;
}
// x ^ (lane K's v), same DPP form; the result is not read by a DPP instruction
template <int K>
__device__ __forceinline__ uint32_t xor_row_bcast(uint32_t v, uint32_t x) {
    uint32_t t;
    asm("v_xor_b32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(t) : "v"(v), "v"(x), "n"(K));
    return t;
}
// key = dist << KEY_SHIFT | row with the (wave-uniform) row number taken from an SGPR: one v_lshl_or_b32 (the compiler's own
// form is a shift plus v_or3 with the row's low bits as a literal).
__device__ __forceinline__ uint32_t make_key(uint32_t d, uint32_t row_uniform) {
    uint32_t key;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(d), "n"(KEY_SHIFT), "s"(row_uniform));
    return key;
}
template <int W>
__device__ __forceinline__ int hamming(const uint32_t (&q)[2 * W], const uint64_t *__restrict__ g) {
    uint32_t d = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const uint64_t gw = g[w];
        d = bcnt_acc(q[2 * w] ^ (uint32_t)gw, d);
        d = bcnt_acc(q[2 * w + 1] ^ (uint32_t)(gw >> 32), d);
    }
    return (int)d;
}

template <int W>
__device__ __forceinline__ void load_query(uint32_t (&q)[2 * W], const uint64_t *qp, int64_t qi, int64_t Qn) {
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const uint64_t v = qi < Qn ? qp[qi * W + w] : 0ull;
        q[2 * w] = (uint32_t)v;
        q[2 * w + 1] = (uint32_t)(v >> 32);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// full distance matrix (small problems)
// ---------------------------------------------------------------------------------------------------------------
__global__ void dist_kernel(const uint64_t *q, int64_t Qn, const uint64_t *g, int64_t G, int W, int32_t *out) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= Qn * G) return;
    const int64_t i = gid / G, j = gid - i * G;
    int d = 0;
    for (int w = 0; w < W; ++w) d += __builtin_popcountll(q[i * W + w] ^ g[j * W + w]);
    out[gid] = d;
}

// ---------------------------------------------------------------------------------------------------------------
// top-k, per (query tile, gallery segment) partial lists
// ---------------------------------------------------------------------------------------------------------------
template <int W, int KREG>
__global__ __launch_bounds__(256) void topk_partial_kernel(const uint64_t *__restrict__ q, int64_t Qn,
                                                           const uint64_t *__restrict__ g, int64_t G, int seg_rows, int k,
                                                           uint32_t *__restrict__ part) {
    const int64_t qi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int seg = blockIdx.y;
    const int64_t g0 = (int64_t)seg * seg_rows;
    const int n = (int)min((int64_t)seg_rows, G - g0);
    uint32_t qw[2 * W];
    load_query<W>(qw, q, qi, Qn);
    uint32_t list[KREG];
#pragma unroll
    for (int i = 0; i < KREG; ++i) list[i] = 0xFFFFFFFFu;
    const uint64_t *gp = g + g0 * W;
    auto insert = [&](uint32_t key) {
        if (__builtin_amdgcn_ballot_w64(key < list[KREG - 1]) != 0ull) {
#pragma unroll
            for (int i = 0; i < KREG; ++i) {
                const uint32_t lo = min(list[i], key);
                key = max(list[i], key);
                list[i] = lo;
            }
        }
    };
    // four gallery rows per trip: one wide scalar load (the next block is requested before this one is consumed), four
    // XOR/popcount keys, ONE threshold test on their minimum; the insertion network runs only if some lane beats its list.
    constexpr int UB = 4;
    uint64_t bufA[UB * W], bufB[UB * W];
    auto load_block = [&](uint64_t (&dst)[UB * W], int row) {
#pragma unroll
        for (int t = 0; t < UB * W; ++t) dst[t] = gp[(size_t)row * W + t];
    };
    auto scan_block = [&](const uint64_t (&blk)[UB * W], int row) {
        uint32_t key[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) key[u] = make_key((uint32_t)hamming<W>(qw, blk + u * W), (uint32_t)(row + u));
        const uint32_t kmin = min(min(key[0], key[1]), min(key[2], key[3]));
        if (__builtin_amdgcn_ballot_w64(kmin < list[KREG - 1]) != 0ull) {
#pragma unroll
            for (int u = 0; u < UB; ++u) insert(key[u]);
        }
    };
    // two blocks per iteration with the two SGPR buffers taking turns: no register copies between trips
    int j = 0;
    if (n >= UB) load_block(bufA, 0);
    for (; j + 2 * UB <= n; j += 2 * UB) {
        load_block(bufB, j + UB);
        scan_block(bufA, j);
        if (j + 3 * UB <= n) load_block(bufA, j + 2 * UB);
        scan_block(bufB, j + UB);
    }
    if (j + UB <= n) {  // an odd number of whole blocks: the last one is already in bufA
        scan_block(bufA, j);
        j += UB;
    }
    for (; j < n; ++j) insert(((uint32_t)hamming<W>(qw, gp + (size_t)j * W) << KEY_SHIFT) | (uint32_t)j);
    if (qi < Qn) {
        uint32_t *o = part + ((size_t)seg * Qn + qi) * k;
#pragma unroll
        for (int i = 0; i < KREG; ++i)
            if (i < k) o[i] = list[i];
    }
}

// one wave per query: repeatedly extract the smallest composite (dist, global row) above the previous one
__global__ __launch_bounds__(256) void topk_merge_keys_kernel(const uint32_t *__restrict__ part, int nseg, int64_t Qn, int k,
                                                              int seg_rows, int64_t g_index_base, int64_t *out_idx,
                                                              int32_t *out_dist) {
    const int64_t qi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (qi >= Qn) return;
    const int ncand = nseg * k;
    unsigned long long prev = 0ull;  // composite + 1 of the last output (0 = none yet)
    for (int r = 0; r < k; ++r) {
        unsigned long long best = ~0ull;
        for (int c = lane; c < ncand; c += 64) {
            const int s = c / k, i = c - s * k;
            const uint32_t key = part[((size_t)s * Qn + qi) * k + i];
            if (key == 0xFFFFFFFFu) continue;
            const unsigned long long comp =
                ((unsigned long long)(key >> KEY_SHIFT) << 40) | ((unsigned long long)s * seg_rows + (key & KEY_MASK));
            if (comp + 1 > prev && comp < best) best = comp;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(best, o, 64);
            best = other < best ? other : best;
        }
        if (lane == 0) {
            if (best == ~0ull) {
                out_idx[qi * k + r] = -1;
                out_dist[qi * k + r] = -1;
            } else {
                out_idx[qi * k + r] = g_index_base + (int64_t)(best & ((1ull << 40) - 1));
                out_dist[qi * k + r] = (int32_t)(best >> 40);
            }
        }
        if (best == ~0ull) {
            // nothing left: fill the rest
            for (int rr = r + 1; rr < k; ++rr)
                if (lane == 0) {
                    out_idx[qi * k + rr] = -1;
                    out_dist[qi * k + rr] = -1;
                }
            return;
        }
        prev = best + 1;
    }
}

// merge already-final lists (idx,dist) from several shards: same extraction on composite (dist, idx)
__global__ __launch_bounds__(256) void topk_merge_lists_kernel(const int64_t *__restrict__ idx_lists,
                                                               const int32_t *__restrict__ dist_lists, int nlists, int64_t Qn,
                                                               int k, int64_t *out_idx, int32_t *out_dist) {
    const int64_t qi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (qi >= Qn) return;
    const int ncand = nlists * k;
    unsigned long long prev = 0ull;
    for (int r = 0; r < k; ++r) {
        unsigned long long best = ~0ull;
        for (int c = lane; c < ncand; c += 64) {
            const int s = c / k, i = c - s * k;
            const int32_t d = dist_lists[((size_t)s * Qn + qi) * k + i];
            if (d < 0) continue;
            const unsigned long long comp =
                ((unsigned long long)d << 48) | (unsigned long long)idx_lists[((size_t)s * Qn + qi) * k + i];
            if (comp + 1 > prev && comp < best) best = comp;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(best, o, 64);
            best = other < best ? other : best;
        }
        if (lane == 0) {
            out_idx[qi * k + r] = best == ~0ull ? -1 : (int64_t)(best & ((1ull << 48) - 1));
            out_dist[qi * k + r] = best == ~0ull ? -1 : (int32_t)(best >> 48);
        }
        if (best != ~0ull) prev = best + 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// mAP passes.  MODE 0 = histogram, MODE 1 = AP accumulation (NR rank limits in one pass)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool relevant_multi(const uint64_t *qm, const uint64_t *gm, int LW) {
    bool r = false;
    for (int w = 0; w < LW; ++w) r |= (qm[w] & gm[w]) != 0ull;
    return r;
}

// floor(a * 2^32 / b) for 1 <= a <= b < 2^32, exact: double estimate (hardware reciprocal + one Newton step, relative error
// ~2^-51, i.e. < 2^-18 absolute on a quotient <= 2^32) + integer correction.  No IEEE division sequence.
__device__ __forceinline__ unsigned long long fixdiv32(uint32_t a, uint32_t b) {
    const unsigned long long num = (unsigned long long)a << 32;
    const double bd = (double)b;
    double r = __builtin_amdgcn_rcp(bd);
    r = __builtin_fma(r, __builtin_fma(-bd, r, 1.0), r);
    unsigned long long qq = (unsigned long long)((double)a * 4294967296.0 * r);
    long long rem = (long long)(num - qq * (unsigned long long)b);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (rem < 0) {
            qq -= 1;
            rem += b;
        }
        if (rem >= (long long)b) {
            qq += 1;
            rem -= b;
        }
    }
    return qq;
}

constexpr int MAX_LIMITS = 16;
struct RankLimits {
    uint32_t lim[MAX_LIMITS];  // mAP@R cut-offs of one AP pass; 0xFFFFFFFF = the whole ranking
};

// one relevant row: exact global rank / relevant-rank from (base, position in bucket), AP term into every limit it is inside
template <int NR>
__device__ __forceinline__ void ap_account(uint2 b, uint32_t old, int skip, int frel, const RankLimits &lims,
                                           unsigned long long (&S)[NR], uint32_t (&nrel)[NR]) {
    uint32_t rank = b.x + (old & 0xFFFFu) + 1u;     // 1-based global rank
    uint32_t relrank = b.y + (old >> 16) + 1u;      // 1-based rank among relevant rows
    if (skip) {
        if (rank == 1u) return;
        rank -= 1u;
        relrank -= (uint32_t)frel;
    }
    if (rank > lims.lim[NR - 1]) return;            // limits ascend: outside the largest = outside all
    const unsigned long long term = fixdiv32(relrank, rank);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const bool in = rank <= lims.lim[r];
        S[r] += in ? term : 0ull;
        nrel[r] += in ? 1u : 0u;
    }
}

// One lane per query; the gallery segment is wave-uniform and walked four rows per trip: codes and labels of a trip come
// through ONE scalar load each (the next trip's block is requested before this one is consumed), distances are chained
// v_bcnt, every row is one LDS counter update in the lane's own column (conflict free).
//   MODE 0: ds_add (no return); at the end the [bucket][lane] counters leave through a coalesced transposed write.
//   MODE 1: ds_add_rtn: the returned value is the row's position inside its (query, distance) bucket in gallery order; only
//           when some lane of the wave holds a relevant row in the trip is the "ranked before" base of each lane's relevant row
//           gathered (`account_trip`) and its AP term added -- to NR accumulators, one per rank limit, so that mAP@R for a list
//           of R, P@k and R@k all come out of a single pass.
//   MODE 2: MODE 0 and MODE 1 in ONE scan (the record form): the histogram is built with ds_add_rtn, and every relevant row leaves
//           a record (distance, returned counter) in a per-lane list in global memory -- what the AP term needs once the "ranked
//           before" bases exist.  `ap_records_kernel` then walks the lists (G / C records per query instead of G rows): the
//           second distance scan disappears.  Lists are bounded (`cap` records per lane and segment); a workgroup with a lane
//           that overflows raises its flag and is redone by the MODE 1 kernel (which then skips every other workgroup).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
struct RecordArgs {
    uint2 *rec;           // [workgroup][cap][BLK] (distance, returned counter); workgroup = segment * tiles + tile
    uint32_t *rec_cnt;    // [nseg][Qn] records of the lane (may exceed cap: then the workgroup is flagged)
    uint32_t *wg_flags;   // [nseg * tiles] 1 = some list overflowed (zeroed by the caller); MODE 1: run only flagged workgroups
    int cap;
};

// VM: the gallery arrives through VMEM in blocks of 16 rows spread over the lanes (in-order vmcnt, four blocks in flight) and is
// broadcast by DPP operands -- instead of scalar loads.  SMEM shares lgkmcnt with the LDS atomics and returns out of order, so the
// scalar form's per-trip `s_waitcnt lgkmcnt(0)` also drains the trip's four atomics; with VMEM the row loop of the histogram pass
// never waits for an atomic, and the AP pass only for counters issued a whole trip earlier.
template <int W, int BLK, int MODE, int NR, bool VM>
__global__ __launch_bounds__(BLK) void map_scan_kernel(const uint64_t *__restrict__ q, int64_t Qn,
                                                       const uint64_t *__restrict__ g, int64_t G, const void *q_labels,
                                                       const void *g_labels, int LW, int seg_rows,
                                                       uint32_t *__restrict__ out_hist, const uint32_t *__restrict__ base,
                                                       RankLimits lims, int nlim, const int32_t *__restrict__ first_rel,
                                                       unsigned long long *out_S, uint32_t *out_nrel, RecordArgs ra) {
    extern __shared__ __attribute__((aligned(16))) uint32_t cnt[];  // [nb][BLK]
    const uint32_t wg_index = blockIdx.y * gridDim.x + blockIdx.x;
    if (MODE == 1 && ra.wg_flags != nullptr && ra.wg_flags[wg_index] == 0u) return;   // fallback launch: only the workgroups whose
                                                                                       // record lists overflowed (wave-uniform)
    constexpr int NB = 64 * W + 1;
    constexpr int UB = 4;
    const int tid = threadIdx.x;
    const int64_t q0 = (int64_t)blockIdx.x * BLK;
    const int64_t qi = q0 + tid;
    const int seg = blockIdx.y;
    const int64_t g0 = (int64_t)seg * seg_rows;
    const int n = (int)min((int64_t)seg_rows, G - g0);
    for (int i = tid; i < NB * BLK; i += BLK) cnt[i] = 0;
    uint32_t qw[2 * W];
    load_query<W>(qw, q, qi, Qn);
    const bool valid = qi < Qn;
    const int32_t qlab = (LW == 0 && valid) ? ((const int32_t *)q_labels)[qi] : -1;
    const uint64_t *qm = (LW > 0 && valid) ? (const uint64_t *)q_labels + qi * LW : nullptr;
    const int32_t *__restrict__ gl32 = (const int32_t *)g_labels + g0;
    const uint64_t *glm = (const uint64_t *)g_labels + g0 * (LW > 0 ? LW : 0);
    __syncthreads();  // counters zeroed (each lane only touches its own column afterwards)

    unsigned long long S[NR];
    uint32_t nrel[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        S[r] = 0ull;
        nrel[r] = 0;
    }
    int skip = 0, frel = 0;
    if (MODE == 1 && first_rel != nullptr) {
        skip = 1;
        frel = valid ? first_rel[qi] : 0;
    }
    const uint64_t *__restrict__ gp = g + g0 * W;
    const size_t brow = ((size_t)seg * Qn + (valid ? qi : 0)) * NB * 2;
    uint32_t *col = cnt + tid;

    auto account = [&](uint2 b, uint32_t old) { ap_account<NR>(b, old, skip, frel, lims, S, nrel); };
    // The relevant rows of the four-row trips (MODE 1).  With C classes a lane holds a relevant row in 4 / C of its trips, so SOME
    // lane of the wave does in most trips while nearly every lane has none; accounting inside the trip (a gather of the "ranked
    // before" base + an exact division, ~750 clocks with one wave per SIMD to hide nothing behind) is what bounded this pass.
    // Instead every lane parks its relevant row -- (distance, returned counter) -- in a one-entry slot, and the wave accounts all
    // parked rows at once only when some lane needs its slot a second time (about every 9 trips at 200 classes: 64 lanes, 2 % each
    // per trip).  The parking of trip t runs in trip t + 1, in front of that trip's atomics, so it never waits for a counter to
    // come back.  The sums are integers: the order of accumulation does not matter.
    // (Tried and slower, 23 vs 16 ms at 16,384 x 1M: the same bookkeeping on the rows' 64-bit lane masks in SGPRs with a scalar
    // test per row -- four more branches per trip cost more than the eight v_cndmask they skip.)
    constexpr uint32_t EMPTY = 0xFFFFFFFFu;
    uint32_t pend_d = EMPTY, pend_old = 0;     // the parked row
    // the previous trip, not yet parked (scalars, not arrays: an array the slow path below indexes ends up in scratch memory)
    uint32_t pd0 = 0, pd1 = 0, pd2 = 0, pd3 = 0, po0 = 0, po1 = 0, po2 = 0, po3 = 0, prev_m = 0;
    static_assert(UB == 4, "park_trip selects among four rows");
    uint32_t nrec = 0;   // MODE 2: records this lane has produced
    auto gather_account = [&](bool has, uint32_t dd, uint32_t oo) {
        if (MODE == 2) {
            if (has) {
                if (nrec < (uint32_t)ra.cap) ra.rec[((size_t)wg_index * ra.cap + nrec) * BLK + tid] = make_uint2(dd, oo);
                nrec += 1u;
            }
            return;
        }
        // lanes without a row read the wave-uniform head of `base`: one extra cache line instead of a divergent branch
        const uint2 b = *(const uint2 *)(has ? base + brow + 2 * dd : base);
        if (has) account(b, oo);
    };
    auto drain = [&]() {
        const bool has = pend_d != EMPTY;
        if (__builtin_amdgcn_ballot_w64(has) != 0ull) {
            gather_account(has, pend_d, pend_old);
            pend_d = EMPTY;
        }
    };
    // the operand of the row `low` (one bit of a trip's relevance mask) names: three v_cndmask
    auto pick = [](uint32_t low, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3) {
        uint32_t r = a3;
        r = (low & 4u) ? a2 : r;
        r = (low & 2u) ? a1 : r;
        r = (low & 1u) ? a0 : r;
        return r;
    };
    auto park_trip = [&]() {
        const uint32_t m = prev_m;
        const bool conflict = m != 0u && (pend_d != EMPTY || (m & (m - 1u)) != 0u);
        if (__builtin_amdgcn_ballot_w64(conflict) != 0ull) {
            drain();
            uint32_t mm = m & (m - 1u);         // a lane's rows beyond its first: accounted directly (rare)
            while (__builtin_amdgcn_ballot_w64(mm != 0u) != 0ull) {
                const uint32_t low = mm & (0u - mm);
                gather_account(mm != 0u, pick(low, pd0, pd1, pd2, pd3), pick(low, po0, po1, po2, po3));
                mm ^= low;
            }
        }
        const uint32_t low = m & (0u - m);
        const uint32_t nd = pick(low, pd0, pd1, pd2, pd3), no = pick(low, po0, po1, po2, po3);
        pend_d = m ? nd : pend_d;
        pend_old = m ? no : pend_old;
    };
    auto keep_trip = [&](const uint32_t (&d)[UB], const uint32_t (&old)[UB], const bool (&rel)[UB]) {
        pd0 = d[0]; pd1 = d[1]; pd2 = d[2]; pd3 = d[3];
        po0 = old[0]; po1 = old[1]; po2 = old[2]; po3 = old[3];
        prev_m = (uint32_t)rel[0] | ((uint32_t)rel[1] << 1) | ((uint32_t)rel[2] << 2) | ((uint32_t)rel[3] << 3);
    };
    auto row = [&](const uint64_t *gw, bool rel) {   // single-row form (segment tail, multi-hot labels)
        const uint32_t d = (uint32_t)hamming<W>(qw, gw);
        const uint32_t inc = 1u | ((uint32_t)rel << 16);
        if (MODE == 0) {
            atomicAdd(col + d * BLK, inc);
        } else if (MODE == 2) {
            const uint32_t old = atomicAdd(col + d * BLK, inc);
            gather_account(rel, d, old);
        } else {
            const uint32_t old = atomicAdd(col + d * BLK, inc);
            if (rel) account(*(const uint2 *)(base + brow + 2 * d), old);
        }
    };

    int j = 0;
    if (LW == 0 && VM) {
        // Blocks of 16 gallery rows, lane l holding row (l & 15) of the block -- its 2 W code words and its label -- so a block
        // costs the texture path TWO OR THREE load instructions per wave instead of the 20 that a same-address ("broadcast")
        // load per row and word needs (those are not free: the address unit walks all 64 lanes of each; at four waves per CU it
        // was the bound of this loop).  Row k of the block reaches every lane through the DPP operand of the xor itself
        // (`v_xor_b32_dpp ... row_newbcast:k`: lane k of the lane's own row of 16), i.e. at no instruction at all.
        // Four blocks in flight behind counted vmcnt waits (inline-asm loads: in order, and invisible to the compiler's own waits).
        constexpr int RB = 16, NBUF = 4;
        constexpr int NLD = (W <= 2) ? 2 : 3;   // load instructions per block
        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
        struct VBlk {
            u32x4_t p4, q4;
            u32x2_t p2;
            uint32_t lab;
        };
        VBlk b0, b1, b2, b3;
        const char *pc = (const char *)(gp + (size_t)(tid & 15) * W);   // per-lane running pointers
        const char *pl = (const char *)(gl32 + (tid & 15));
#define CH_GLD(inst, dst, ptr, off) asm volatile(inst " %0, %1, off offset:" #off : "=v"(dst) : "v"(ptr) : "memory")
        auto issue = [&](VBlk &b) {
            if constexpr (W == 1) CH_GLD("global_load_dwordx2", b.p2, pc, 0);
            if constexpr (W == 2) CH_GLD("global_load_dwordx4", b.p4, pc, 0);
            if constexpr (W == 3) {
                CH_GLD("global_load_dwordx4", b.p4, pc, 0);
                CH_GLD("global_load_dwordx2", b.p2, pc, 16);
            }
            if constexpr (W == 4) {
                CH_GLD("global_load_dwordx4", b.p4, pc, 0);
                CH_GLD("global_load_dwordx4", b.q4, pc, 16);
            }
            CH_GLD("global_load_dword", b.lab, pl, 0);
            pc += RB * W * 8;
            pl += RB * 4;
        };
#undef CH_GLD
        // wait until all but the `younger` blocks issued after `b` have landed; the "+v" operands pin every use of the block
        // behind the wait
        auto landed = [&](VBlk &b, auto younger) {
            constexpr int N = decltype(younger)::value * NLD;
            if constexpr (W == 1) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b.p2), "+v"(b.lab) : "n"(N));
            if constexpr (W == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b.p4), "+v"(b.lab) : "n"(N));
            if constexpr (W == 3) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(b.p4), "+v"(b.p2), "+v"(b.lab) : "n"(N));
            if constexpr (W == 4) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(b.p4), "+v"(b.q4), "+v"(b.lab) : "n"(N));
        };
        auto word = [&](const VBlk &b, int i) -> uint32_t {
            if constexpr (W == 1) return b.p2[i];
            if constexpr (W == 2) return b.p4[i];
            if constexpr (W == 3) return i < 4 ? b.p4[i] : b.p2[i - 4];
            if constexpr (W == 4) return i < 4 ? b.p4[i] : b.q4[i - 4];
        };
        // rows 4 G .. 4 G + 3 of a block: one "trip" of the parking logic
        auto group = [&](const VBlk &b, auto gc) {
            constexpr int G = decltype(gc)::value;
            uint32_t d[UB], old[UB];
            bool rel[UB];
            auto one = [&](auto uc) {
                constexpr int U = decltype(uc)::value, K = 4 * G + U;
                uint32_t acc = 0;
#pragma unroll
                for (int w = 0; w < W; ++w)
                    acc = xor_bcnt2_row_bcast<K>(word(b, 2 * w), word(b, 2 * w + 1), qw[2 * w], qw[2 * w + 1], acc);
                d[U] = acc;
                bool r = xor_row_bcast<K>(b.lab, (uint32_t)qlab) == 0u;   // all lanes take part in the DPP read: no `valid &&` in front
                r &= valid;
                rel[U] = r;
            };
            one(std::integral_constant<int, 0>{});
            one(std::integral_constant<int, 1>{});
            one(std::integral_constant<int, 2>{});
            one(std::integral_constant<int, 3>{});
            if (MODE != 0) park_trip();     // the previous trip's rows; its counters came back long ago
#pragma unroll
            for (int u = 0; u < UB; ++u) {  // in row order: two rows of a trip may share a bucket
                const uint32_t inc = 1u | ((uint32_t)rel[u] << 16);
                if (MODE == 0)
                    atomicAdd(col + d[u] * BLK, inc);
                else
                    old[u] = atomicAdd(col + d[u] * BLK, inc);
            }
            if (MODE != 0) keep_trip(d, old, rel);
        };
        auto scan_vblk = [&](const VBlk &b) {
            group(b, std::integral_constant<int, 0>{});
            group(b, std::integral_constant<int, 1>{});
            group(b, std::integral_constant<int, 2>{});
            group(b, std::integral_constant<int, 3>{});
        };
        {   // the query words come from a compiler-managed load: consume them once HERE, or the compiler's own wait for them
            // (vmcnt(0): it cannot see the asm loads) lands inside the loop and drains the prefetch every iteration
            uint32_t use = 0;
#pragma unroll
            for (int w = 0; w < 2 * W; ++w) use |= qw[w];
            asm volatile("" ::"v"(use), "v"(qlab));
        }
        const std::integral_constant<int, 0> y0{};
        const std::integral_constant<int, 1> y1{};
        const std::integral_constant<int, 2> y2{};
        const std::integral_constant<int, 3> y3{};
        const int nring = n / (RB * NBUF);   // rounds of the four-block ring
        if (nring > 0) {   // ONE region: every path that issues ring loads also runs the last round's waits (tools/check_dpp_hazards.py
                           // follows the control-flow graph and would otherwise see an infeasible "loop, then skip the drain" path)
            issue(b0);
            issue(b1);
            issue(b2);
            issue(b3);
            for (int it = 0; it + 1 < nring; ++it) {   // steady state: the three blocks issued after a slot's are still in flight
                landed(b0, y3); scan_vblk(b0); issue(b0);
                landed(b1, y3); scan_vblk(b1); issue(b1);
                landed(b2, y3); scan_vblk(b2); issue(b2);
                landed(b3, y3); scan_vblk(b3); issue(b3);
            }
            landed(b0, y3); scan_vblk(b0);             // last round: nothing left to issue
            landed(b1, y2); scan_vblk(b1);
            landed(b2, y1); scan_vblk(b2);
            landed(b3, y0); scan_vblk(b3);
        }
        j = nring * RB * NBUF;
        for (; j < n; ++j) row(gp + (size_t)j * W, valid && (gl32[j] == qlab));
    } else if (LW == 0) {
        uint64_t bufA[UB * W], bufB[UB * W];
        int32_t labA[UB], labB[UB];
        auto load_block = [&](uint64_t (&dst)[UB * W], int32_t (&lab)[UB], int r0) {
#pragma unroll
            for (int t = 0; t < UB * W; ++t) dst[t] = gp[(size_t)r0 * W + t];
#pragma unroll
            for (int u = 0; u < UB; ++u) lab[u] = gl32[r0 + u];
        };
        auto scan_block = [&](const uint64_t (&blk)[UB * W], const int32_t (&lab)[UB]) {
            uint32_t d[UB], old[UB];
            bool rel[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                d[u] = (uint32_t)hamming<W>(qw, blk + u * W);
                rel[u] = valid && (lab[u] == qlab);
            }
            if (MODE != 0) park_trip();     // the previous trip's rows; its counters came back long ago
#pragma unroll
            for (int u = 0; u < UB; ++u) {  // in row order: two rows of a trip may share a bucket
                const uint32_t inc = 1u | ((uint32_t)rel[u] << 16);
                if (MODE == 0)
                    atomicAdd(col + d[u] * BLK, inc);
                else
                    old[u] = atomicAdd(col + d[u] * BLK, inc);
            }
            if (MODE != 0) keep_trip(d, old, rel);
        };
        if (n >= UB) load_block(bufA, labA, 0);
        for (; j + 2 * UB <= n; j += 2 * UB) {
            load_block(bufB, labB, j + UB);
            scan_block(bufA, labA);
            if (j + 3 * UB <= n) load_block(bufA, labA, j + 2 * UB);
            scan_block(bufB, labB);
        }
        if (j + UB <= n) {  // an odd number of whole blocks: the last one is already in bufA
            scan_block(bufA, labA);
            j += UB;
        }
        for (; j < n; ++j) row(gp + (size_t)j * W, valid && (gl32[j] == qlab));
    } else {
        for (; j < n; ++j) row(gp + (size_t)j * W, valid && relevant_multi(qm, glm + (size_t)j * LW, LW));
    }

    if (MODE != 0) {
        park_trip();
        drain();
    }
    if (MODE == 2) {
        if (valid) ra.rec_cnt[(size_t)seg * Qn + qi] = nrec;
        if (__builtin_amdgcn_ballot_w64(nrec > (uint32_t)ra.cap) != 0ull && (tid & 63) == 0) ra.wg_flags[wg_index] = 1u;
    }
    if (MODE == 0 || MODE == 2) {
        // out_hist[seg][q][bucket] = (count, relevant count): element e of this tile's [BLK][NB] block is written by thread
        // e % BLK -> consecutive lanes write consecutive 8-byte elements
        __syncthreads();
        const int nq = (int)min((int64_t)BLK, Qn - q0);
        uint2 *o = (uint2 *)(out_hist + ((size_t)seg * Qn + q0) * NB * 2);
        for (int e = tid; e < nq * NB; e += BLK) {
            const int ql = e / NB, d = e - ql * NB;
            const uint32_t v = cnt[d * BLK + ql];
            o[e] = make_uint2(v & 0xFFFFu, v >> 16);
        }
    } else {
        if (valid) {
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if (r < nlim && nrel[r]) {  // out_S / out_nrel are [nlim, Qn]; limits past nlim are padding
                    atomicAdd(out_S + (size_t)r * Qn + qi, S[r]);
                    atomicAdd(out_nrel + (size_t)r * Qn + qi, nrel[r]);
                }
        }
    }
}

// The AP pass of the record form: one lane per query (same grid as the scan), walking the records its MODE 2 workgroup left.
// Workgroups whose lists overflowed are skipped here and redone by the MODE 1 scan.
template <int NR>
__global__ __launch_bounds__(256) void ap_records_kernel(int64_t Qn, int NB, RecordArgs ra, const uint32_t *__restrict__ base,
                                                         RankLimits lims, int nlim, const int32_t *__restrict__ first_rel,
                                                         unsigned long long *out_S, uint32_t *out_nrel) {
    const int BLK = blockDim.x, tid = threadIdx.x;
    const uint32_t wg_index = blockIdx.y * gridDim.x + blockIdx.x;
    if (ra.wg_flags[wg_index] != 0u) return;
    const int seg = blockIdx.y;
    const int64_t qi = (int64_t)blockIdx.x * BLK + tid;
    const bool valid = qi < Qn;
    const uint32_t n = valid ? ra.rec_cnt[(size_t)seg * Qn + qi] : 0u;
    unsigned long long S[NR];
    uint32_t nrel[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        S[r] = 0ull;
        nrel[r] = 0;
    }
    int skip = 0, frel = 0;
    if (first_rel != nullptr) {
        skip = 1;
        frel = valid ? first_rel[qi] : 0;
    }
    const uint32_t *brow = base + ((size_t)seg * Qn + (valid ? qi : 0)) * NB * 2;
    const uint2 *rp = ra.rec + (size_t)wg_index * ra.cap * BLK + tid;
    // two records per trip: both record loads, then both base gathers, in flight together
    for (uint32_t k = 0; __builtin_amdgcn_ballot_w64(k < n) != 0ull; k += 2) {
        const bool h0 = k < n, h1 = k + 1 < n;
        const uint2 r0 = h0 ? rp[(size_t)k * BLK] : make_uint2(0u, 0u);
        const uint2 r1 = h1 ? rp[(size_t)(k + 1) * BLK] : make_uint2(0u, 0u);
        const uint2 b0 = *(const uint2 *)(brow + 2 * r0.x), b1 = *(const uint2 *)(brow + 2 * r1.x);
        if (h0) ap_account<NR>(b0, r0.y, skip, frel, lims, S, nrel);
        if (h1) ap_account<NR>(b1, r1.y, skip, frel, lims, S, nrel);
    }
    if (valid) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (r < nlim && nrel[r]) {
                atomicAdd(out_S + (size_t)r * Qn + qi, S[r]);
                atomicAdd(out_nrel + (size_t)r * Qn + qi, nrel[r]);
            }
    }
}

// hist [nseg,Qn,nb,2] -> base: exclusive prefix in ranking order (bucket ascending, then segment ascending).
// One wave per query, a lane per bucket (coalesced 8-byte accesses): per-bucket totals over the segments, a wave scan across
// buckets, then the per-segment bases.
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, int lane, uint32_t &total) {
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    total = __shfl(inc, 63, 64);
    return inc - v;
}
__global__ __launch_bounds__(256) void hist_prefix_kernel(const uint32_t *__restrict__ hist, int nseg, int64_t Qn, int nb,
                                                          uint32_t *__restrict__ base, uint32_t *__restrict__ totals) {
    const int64_t qi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (qi >= Qn) return;  // wave-uniform
    const uint2 *h = (const uint2 *)hist;
    uint2 *b = (uint2 *)base;
    uint32_t carry_a = 0, carry_r = 0;
    for (int d0 = 0; d0 < nb; d0 += 64) {
        const int d = d0 + lane;
        const bool act = d < nb;
        uint32_t ta = 0, tr = 0;
        for (int s = 0; s < nseg; ++s)
            if (act) {
                const uint2 v = h[((size_t)s * Qn + qi) * nb + d];
                ta += v.x;
                tr += v.y;
            }
        uint32_t sum_a, sum_r;
        uint32_t ba = carry_a + wave_excl_scan(ta, lane, sum_a), br = carry_r + wave_excl_scan(tr, lane, sum_r);
        for (int s = 0; s < nseg; ++s)
            if (act) {
                const size_t off = ((size_t)s * Qn + qi) * nb + d;
                const uint2 v = h[off];
                b[off] = make_uint2(ba, br);
                ba += v.x;
                br += v.y;
            }
        carry_a += sum_a;
        carry_r += sum_r;
    }
    if (totals && lane == 0) {
        totals[qi * 2] = carry_a;
        totals[qi * 2 + 1] = carry_r;
    }
}

struct ScanArgs {   // everything a map_scan_kernel launch takes
    const uint64_t *q;
    int64_t Qn;
    const uint64_t *g;
    int64_t G;
    const void *ql, *gl;
    int LW, seg_rows;
    uint32_t *out_hist;
    const uint32_t *base;
    RankLimits lims;
    int nlim;
    const int32_t *first_rel;
    unsigned long long *out_S;
    uint32_t *out_nrel;
    RecordArgs ra;
};

template <int W, int BLK, int MODE, int NR, bool VM>
int launch_scan_vm(const ScanArgs &a, hipStream_t s) {
    const int nseg = (int)ceil_div64(a.G, a.seg_rows);
    const size_t lds = sizeof(uint32_t) * (64 * W + 1) * BLK;
    dim3 grid((unsigned)ceil_div64(a.Qn, BLK), (unsigned)nseg);
    static ch_once_per_device lds_once;
    if (int e = ch_func_max_lds((const void *)map_scan_kernel<W, BLK, MODE, NR, VM>, (int)lds, lds_once)) return e;
    hipLaunchKernelGGL((map_scan_kernel<W, BLK, MODE, NR, VM>), grid, dim3(BLK), lds, s, a.q, a.Qn, a.g, a.G, a.ql, a.gl, a.LW, a.seg_rows,
                       a.out_hist, a.base, a.lims, a.nlim, a.first_rel, a.out_S, a.out_nrel, a.ra);
    CH_LAUNCH_CHECK();
    return 0;
}

template <int W, int BLK, int MODE, int NR>
int launch_scan_nr(const ScanArgs &a, hipStream_t s) {
    // All passes take the gallery through VMEM in 16-row blocks spread over the lanes + DPP broadcast (at 16,384 x 1M x 128 bit:
    // histogram pass 18.9 ms with scalar loads -> 14.4 ms with same-address VMEM loads -> 8.0 ms; AP pass 39 -> 16 ms together
    // with the parked accounting).  ch_debug_set_hamming_scalar_loads(1) = the scalar-load form (same results; kept as the cross-check).
    if (!g_scalar_loads.load(std::memory_order_relaxed)) return launch_scan_vm<W, BLK, MODE, NR, true>(a, s);
    return launch_scan_vm<W, BLK, MODE, NR, false>(a, s);
}

// mode 0: histogram, 1: AP pass (all workgroups, or only the flagged ones when a.ra.wg_flags is set), 2: histogram + records
template <int W, int BLK>
int launch_scan(int mode, const ScanArgs &a, hipStream_t s) {
    if (mode == 0) return launch_scan_nr<W, BLK, 0, 1>(a, s);
    if (mode == 2) return launch_scan_nr<W, BLK, 2, 1>(a, s);
    if (a.nlim == 1) return launch_scan_nr<W, BLK, 1, 1>(a, s);
    if (a.nlim <= 4) return launch_scan_nr<W, BLK, 1, 4>(a, s);
    return launch_scan_nr<W, BLK, 1, MAX_LIMITS>(a, s);
}

// lims: nlim ascending rank limits, padded to MAX_LIMITS with copies of the last one (the kernel instance accumulates
// NR = 1 / 4 / 16 of them and stores rows [0, nlim) of out_S / out_nrel)
int scan_dispatch(int mode, const ScanArgs &a, int W, hipStream_t s) {
    CH_REQUIRE(W >= 1 && W <= 4, "hamming: 1 <= W <= 4 (nbit <= 256)");
    CH_REQUIRE(a.seg_rows >= 1 && a.seg_rows <= 65535, "hamming: seg_rows must be in [1, 65535]");
    CH_REQUIRE(a.LW >= 0, "hamming: LW must be >= 0");
    CH_REQUIRE(a.Qn >= 0 && a.G >= 0, "hamming: negative sizes");
    if (a.Qn == 0 || a.G == 0) return 0;
    CH_REQUIRE(a.q && a.g && a.ql && a.gl, "hamming: null pointer");
    CH_REQUIRE(ceil_div64(a.G, a.seg_rows) <= 65535, "hamming: too many gallery segments (raise seg_rows)");
    switch (W) {
        case 1: return launch_scan<1, 256>(mode, a, s);
        case 2: return launch_scan<2, 256>(mode, a, s);
        case 3: return launch_scan<3, 128>(mode, a, s);
        default: return launch_scan<4, 128>(mode, a, s);
    }
}

bool fill_limits(const int64_t *rank_limits, int nlimits, RankLimits &lims) {
    for (int i = 0; i < MAX_LIMITS; ++i) {
        const int64_t r = rank_limits[i < nlimits ? i : nlimits - 1];
        lims.lim[i] = (r <= 0 || r >= 0xFFFFFFFFll) ? 0xFFFFFFFFu : (uint32_t)r;
        if (i > 0 && lims.lim[i] < lims.lim[i - 1]) return false;
    }
    return true;
}

int topk_seg_rows(int64_t Qn, int64_t G, int k) {
    // (tile, segment) workgroups for ONE full round and never a few more: the scan kernel has no LDS, so residency is set by its
    // registers -- 24-42 VGPRs for lists of <= 16 keys (8 workgroups of four waves per CU), 68-74 for 32 (7), ~136 for 64 (3),
    // the whole file for 128 (1).  ceil(2048 / tiles) segments put 2,134 workgroups on the 2,048 slots at the NABirds size
    // (97 tiles): a second round for 86 of them doubled the launch (0.38 -> 0.2 ms).  Segments not shorter than 256 rows.
    const int per_cu = k <= 16 ? 8 : k <= 32 ? 7 : k <= 64 ? 3 : 1;
    const int64_t slots = 256 * per_cu;
    const int64_t tiles = ceil_div64(Qn, 256);
    int64_t nseg = std::max<int64_t>(1, slots / tiles);
    // ... and not more segments than needed: every segment starts with empty lists, so its first ~640 rows run the insertion
    // network for some lane of the wave almost every row, and the merge cost grows with the segment count -- segments of >= 4,096
    // rows as long as two workgroups per CU remain (NABirds size: 6 segments instead of 21, 0.45 -> 0.37 ms; the 1M-row scan keeps 32)
    nseg = std::min(nseg, std::max<int64_t>(std::max<int64_t>(1, ceil_div64(512, tiles)), G / 4096));
    int64_t rows = ceil_div64(G, nseg);
    if (rows < 256) rows = 256;
    if (rows > (int64_t)KEY_MASK) rows = KEY_MASK;
    return (int)rows;
}

template <int W, int KREG>
int launch_topk_partial(const uint64_t *q, int64_t Qn, const uint64_t *g, int64_t G, int seg_rows, int k, uint32_t *part,
                        hipStream_t s) {
    const int nseg = (int)ceil_div64(G, seg_rows);
    dim3 grid((unsigned)ceil_div64(Qn, 256), (unsigned)nseg);
    hipLaunchKernelGGL((topk_partial_kernel<W, KREG>), grid, dim3(256), 0, s, q, Qn, g, G, seg_rows, k, part);
    CH_LAUNCH_CHECK();
    return 0;
}

template <int W>
int topk_dispatch_k(const uint64_t *q, int64_t Qn, const uint64_t *g, int64_t G, int seg_rows, int k, uint32_t *part,
                    hipStream_t s) {
    if (k <= 10) return launch_topk_partial<W, 10>(q, Qn, g, G, seg_rows, k, part, s);  // PRs = [1, 5, 10]: exactly k entries, so the
                                                                                         // insertion threshold is the k-th key
    if (k <= 16) return launch_topk_partial<W, 16>(q, Qn, g, G, seg_rows, k, part, s);
    if (k <= 32) return launch_topk_partial<W, 32>(q, Qn, g, G, seg_rows, k, part, s);
    if (k <= 64) return launch_topk_partial<W, 64>(q, Qn, g, G, seg_rows, k, part, s);
    return launch_topk_partial<W, 128>(q, Qn, g, G, seg_rows, k, part, s);
}

}  // namespace

extern "C" int ch_hamming_dist(const uint64_t *q, int64_t Qn, const uint64_t *g, int64_t G, int32_t W, int32_t *out,
                               void *stream) {
    CH_REQUIRE(W >= 1 && Qn >= 0 && G >= 0, "hamming_dist: bad sizes");
    if (Qn == 0 || G == 0) return 0;
    CH_REQUIRE(q && g && out, "hamming_dist: null pointer");
    hipLaunchKernelGGL(dist_kernel, dim3((unsigned)ceil_div64(Qn * G, 256)), dim3(256), 0, (hipStream_t)stream, q, Qn, g, G, W,
                       out);
    CH_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t ch_hamming_topk_workspace(int64_t Qn, int64_t G, int32_t W, int32_t k) {
    (void)W;
    if (Qn <= 0 || G <= 0 || k <= 0) return 16;
    const int seg_rows = topk_seg_rows(Qn, G, k);
    const int64_t nseg = ceil_div64(G, seg_rows);
    return (size_t)(nseg * Qn * k) * sizeof(uint32_t) + 16;
}

extern "C" int ch_hamming_topk(const uint64_t *q, int64_t Qn, const uint64_t *g, int64_t G, int32_t W, int32_t k,
                               int64_t g_index_base, int64_t *out_idx, int32_t *out_dist, void *workspace,
                               size_t workspace_bytes, void *stream) {
    CH_REQUIRE(W >= 1 && W <= 4, "hamming_topk: 1 <= W <= 4 (nbit <= 256)");
    CH_REQUIRE(k >= 1 && k <= 128, "hamming_topk: 1 <= k <= 128");
    CH_REQUIRE(Qn >= 0 && G >= 0, "hamming_topk: negative sizes");
    if (Qn == 0) return 0;
    CH_REQUIRE(q && out_idx && out_dist, "hamming_topk: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (G == 0) {
        CH_CHECK_HIP(hipMemsetAsync(out_idx, 0xFF, sizeof(int64_t) * Qn * k, s));
        CH_CHECK_HIP(hipMemsetAsync(out_dist, 0xFF, sizeof(int32_t) * Qn * k, s));
        return 0;
    }
    CH_REQUIRE(g != nullptr, "hamming_topk: null gallery");
    CH_REQUIRE(workspace && workspace_bytes >= ch_hamming_topk_workspace(Qn, G, W, k), "hamming_topk: workspace too small");
    const int seg_rows = topk_seg_rows(Qn, G, k);
    const int nseg = (int)ceil_div64(G, seg_rows);
    CH_REQUIRE(nseg <= 65535, "hamming_topk: gallery too large for one call (shard it)");
    uint32_t *part = (uint32_t *)workspace;
    int e;
    switch (W) {
        case 1: e = topk_dispatch_k<1>(q, Qn, g, G, seg_rows, k, part, s); break;
        case 2: e = topk_dispatch_k<2>(q, Qn, g, G, seg_rows, k, part, s); break;
        case 3: e = topk_dispatch_k<3>(q, Qn, g, G, seg_rows, k, part, s); break;
        default: e = topk_dispatch_k<4>(q, Qn, g, G, seg_rows, k, part, s); break;
    }
    if (e) return e;
    hipLaunchKernelGGL(topk_merge_keys_kernel, dim3((unsigned)ceil_div64(Qn, 4)), dim3(256), 0, s, part, nseg, Qn, k, seg_rows,
                       g_index_base, out_idx, out_dist);
    CH_LAUNCH_CHECK();
    return 0;
}

extern "C" int ch_topk_merge(const int64_t *idx_lists, const int32_t *dist_lists, int32_t nlists, int64_t Qn, int32_t k,
                             int64_t *out_idx, int32_t *out_dist, void *stream) {
    CH_REQUIRE(nlists >= 1 && k >= 1 && Qn >= 0, "topk_merge: bad sizes");
    if (Qn == 0) return 0;
    CH_REQUIRE(idx_lists && dist_lists && out_idx && out_dist, "topk_merge: null pointer");
    hipLaunchKernelGGL(topk_merge_lists_kernel, dim3((unsigned)ceil_div64(Qn, 4)), dim3(256), 0, (hipStream_t)stream, idx_lists,
                       dist_lists, nlists, Qn, k, out_idx, out_dist);
    CH_LAUNCH_CHECK();
    return 0;
}

extern "C" int ch_hamming_hist(const uint64_t *q, int64_t Qn, const uint64_t *g, int64_t G, int32_t W, const void *q_labels,
                               const void *g_labels, int32_t LW, int32_t seg_rows, uint32_t *out_hist, void *stream) {
    CH_REQUIRE(Qn == 0 || G == 0 || out_hist != nullptr, "hamming_hist: null output");
    ScanArgs a{q, Qn, g, G, q_labels, g_labels, LW, seg_rows, out_hist, nullptr, RankLimits{}, 1, nullptr, nullptr, nullptr, RecordArgs{}};
    return scan_dispatch(0, a, W, (hipStream_t)stream);
}

extern "C" int ch_hamming_ap_multi(const uint64_t *q, int64_t Qn, const uint64_t *g, int64_t G, int32_t W, const void *q_labels,
                                   const void *g_labels, int32_t LW, int32_t seg_rows, const uint32_t *base,
                                   const int64_t *rank_limits, int32_t nlimits, const int32_t *first_rel,
                                   unsigned long long *out_S, uint32_t *out_nrel, void *stream) {
    CH_REQUIRE(nlimits >= 1 && nlimits <= MAX_LIMITS && rank_limits != nullptr, "hamming_ap_multi: 1 <= nlimits <= 16");
    CH_REQUIRE(Qn == 0 || G == 0 || (base && out_S && out_nrel), "hamming_ap_multi: null pointer");
    RankLimits lims;
    CH_REQUIRE(fill_limits(rank_limits, nlimits, lims), "hamming_ap_multi: rank limits must ascend (<= 0 = unlimited, last)");
    ScanArgs a{q, Qn, g, G, q_labels, g_labels, LW, seg_rows, nullptr, base, lims, nlimits, first_rel, out_S, out_nrel, RecordArgs{}};
    return scan_dispatch(1, a, W, (hipStream_t)stream);
}

extern "C" int ch_hamming_ap(const uint64_t *q, int64_t Qn, const uint64_t *g, int64_t G, int32_t W, const void *q_labels,
                             const void *g_labels, int32_t LW, int32_t seg_rows, const uint32_t *base, int64_t rank_limit,
                             const int32_t *first_rel, unsigned long long *out_S, uint32_t *out_nrel, void *stream) {
    return ch_hamming_ap_multi(q, Qn, g, G, W, q_labels, g_labels, LW, seg_rows, base, &rank_limit, 1, first_rel, out_S, out_nrel,
                               stream);
}

// ---- record form: ONE distance scan (csrc/hamming.hip, MODE 2) --------------------------------------------------------------
extern "C" size_t ch_hamming_rec_workgroups(int64_t Qn, int64_t G, int32_t W, int32_t seg_rows) {
    if (Qn <= 0 || G <= 0 || seg_rows <= 0) return 0;
    return (size_t)(ceil_div64(Qn, W <= 2 ? 256 : 128) * ceil_div64(G, seg_rows));
}
extern "C" int32_t ch_hamming_rec_block(int32_t W) { return W <= 2 ? 256 : 128; }

extern "C" int ch_hamming_hist_rec(const uint64_t *q, int64_t Qn, const uint64_t *g, int64_t G, int32_t W, const void *q_labels,
                                   const void *g_labels, int32_t LW, int32_t seg_rows, uint32_t *out_hist, void *rec,
                                   int32_t rec_cap, uint32_t *rec_cnt, uint32_t *wg_flags, void *stream) {
    CH_REQUIRE(Qn == 0 || G == 0 || (out_hist && rec && rec_cnt && wg_flags), "hamming_hist_rec: null pointer");
    CH_REQUIRE(rec_cap >= 1, "hamming_hist_rec: rec_cap must be >= 1");
    ScanArgs a{q, Qn, g, G, q_labels, g_labels, LW, seg_rows, out_hist, nullptr, RankLimits{}, 1, nullptr, nullptr, nullptr,
               RecordArgs{(uint2 *)rec, rec_cnt, wg_flags, rec_cap}};
    return scan_dispatch(2, a, W, (hipStream_t)stream);
}

extern "C" int ch_hamming_ap_rec(const uint64_t *q, int64_t Qn, const uint64_t *g, int64_t G, int32_t W, const void *q_labels,
                                 const void *g_labels, int32_t LW, int32_t seg_rows, const uint32_t *base, const void *rec,
                                 int32_t rec_cap, const uint32_t *rec_cnt, const uint32_t *wg_flags, const int64_t *rank_limits,
                                 int32_t nlimits, const int32_t *first_rel, unsigned long long *out_S, uint32_t *out_nrel,
                                 void *stream) {
    CH_REQUIRE(nlimits >= 1 && nlimits <= MAX_LIMITS && rank_limits != nullptr, "hamming_ap_rec: 1 <= nlimits <= 16");
    CH_REQUIRE(W >= 1 && W <= 4, "hamming_ap_rec: 1 <= W <= 4");
    CH_REQUIRE(seg_rows >= 1 && seg_rows <= 65535 && rec_cap >= 1, "hamming_ap_rec: bad seg_rows / rec_cap");
    if (Qn <= 0 || G <= 0) return 0;
    CH_REQUIRE(base && rec && rec_cnt && wg_flags && out_S && out_nrel, "hamming_ap_rec: null pointer");
    RankLimits lims;
    CH_REQUIRE(fill_limits(rank_limits, nlimits, lims), "hamming_ap_rec: rank limits must ascend (<= 0 = unlimited, last)");
    hipStream_t s = (hipStream_t)stream;
    const int BLK = W <= 2 ? 256 : 128, NB = 64 * W + 1;
    dim3 grid((unsigned)ceil_div64(Qn, BLK), (unsigned)ceil_div64(G, seg_rows));
    RecordArgs ra{(uint2 *)rec, (uint32_t *)rec_cnt, (uint32_t *)wg_flags, rec_cap};
    if (nlimits == 1)
        hipLaunchKernelGGL(ap_records_kernel<1>, grid, dim3(BLK), 0, s, Qn, NB, ra, base, lims, nlimits, first_rel, out_S, out_nrel);
    else if (nlimits <= 4)
        hipLaunchKernelGGL(ap_records_kernel<4>, grid, dim3(BLK), 0, s, Qn, NB, ra, base, lims, nlimits, first_rel, out_S, out_nrel);
    else
        hipLaunchKernelGGL(ap_records_kernel<MAX_LIMITS>, grid, dim3(BLK), 0, s, Qn, NB, ra, base, lims, nlimits, first_rel, out_S, out_nrel);
    CH_LAUNCH_CHECK();
    // the workgroups whose lists overflowed: the two-scan form, which skips every unflagged workgroup at its first instruction
    ScanArgs a{q, Qn, g, G, q_labels, g_labels, LW, seg_rows, nullptr, base, lims, nlimits, first_rel, out_S, out_nrel, ra};
    return scan_dispatch(1, a, W, s);
}

extern "C" int ch_hamming_hist_prefix(const uint32_t *hist, int32_t nseg, int64_t Qn, int32_t nb, uint32_t *out_base,
                                      uint32_t *out_totals, void *stream) {
    CH_REQUIRE(nseg >= 1 && nb >= 1 && Qn >= 0, "hist_prefix: bad sizes");
    if (Qn == 0) return 0;
    CH_REQUIRE(hist && out_base, "hist_prefix: null pointer");
    hipLaunchKernelGGL(hist_prefix_kernel, dim3((unsigned)ceil_div64(Qn, 4)), dim3(256), 0, (hipStream_t)stream, hist, nseg, Qn, nb,
                       out_base, out_totals);
    CH_LAUNCH_CHECK();
    return 0;
}
