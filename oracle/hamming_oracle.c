/*
 * CPU oracle for the ConceptHash *retrieve* path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * PARITY UNPINNED: the reference's retrieval arithmetic lives in `utils.hashing`
 * ({calculate_mAP, calculate_pr_curve, get_hamm_dist}), an un-vendored module of another repository
 * (kamwoh/sdc `utils` package, no version pin; pointer: /root/reference/README.md:11).  Its source is not
 * in the snapshot and no reference test pins its results.  This file therefore restates the NORMATIVE
 * DEFINITION published in SURVEY.md section 8c, anchored on the reference's call sites:
 *   experiments/test_hashing.py:106-119   calculate_mAP(db_codes, db_labels, test_codes, test_labels, R,
 *                                          threshold=ternary_threshold, dist_metric, PRs, remove_first_retrieved)
 *   experiments/test_hashing.py:153-162   calculate_pr_curve(...)
 *   trainers/orthohash.py:263-264         get_hd(a,b) = 0.5*(nbit - a.b^T)/nbit on +-1 codes  (in-repo twin of
 *                                          get_hamm_dist; equals popcount(xor)/nbit, checked in tests)
 *   utils/metrics.py:18-29                argmin / 5-smallest consumer semantics
 *
 * Definition:
 *   bit_i      = (code_i - threshold) > 0, packed little-endian: bit i -> word i/64, bit position i%64
 *   dist(q,g)  = popcount(q xor g) summed over W = ceil(nbit/64) words
 *   ranking    = ascending (dist, gallery index)            [stable]
 *   relevant   = labels share >= 1 class
 *   AP@R       = sum_{r<=R, rel(r)} relrank(r)/r  /  #rel in top-R   (0 when no relevant in top-R); R<=0 => R=G
 *   fixed-point AP numerator  S = sum floor(relrank * 2^32 / r)  (order-independent integer sum; the HIP path and
 *                               the multi-GPU reduction reproduce S bit-for-bit), AP = S / (nrel * 2^32)
 *   P@k = #rel in top-k / k ;  R@k = #rel in top-k / #rel in gallery (0 if none)
 *   remove_first_retrieved: rank 1 is dropped before everything else (self-match when test set == database)
 *
 * The algorithm here (full per-query stable counting sort, then a sequential walk) is deliberately different
 * from the HIP kernels (bucket histograms + prefix bases), so agreement is a real check.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef int64_t i64;
typedef uint32_t u32;
typedef int32_t i32;

static inline int popc64(u64 x) { return __builtin_popcountll(x); }

void ho_pack(const float *codes, i64 rows, int nbit, float threshold, u64 *out) {
    int W = (nbit + 63) / 64;
    for (i64 r = 0; r < rows; ++r) {
        for (int w = 0; w < W; ++w) out[r * W + w] = 0;
        for (int i = 0; i < nbit; ++i)
            if ((codes[r * nbit + i] - threshold) > 0.0f) out[r * W + (i >> 6)] |= (u64)1 << (i & 63);
    }
}

void ho_dist(const u64 *q, const u64 *g, i64 Qn, i64 G, int W, i32 *out) {
    for (i64 i = 0; i < Qn; ++i)
        for (i64 j = 0; j < G; ++j) {
            int d = 0;
            for (int w = 0; w < W; ++w) d += popc64(q[i * W + w] ^ g[j * W + w]);
            out[i * G + j] = d;
        }
}

/* stable counting sort of gallery indices by distance: order[] = ranking of one query */
static void rank_one(const u64 *q, const u64 *g, i64 G, int W, int nb, i32 *dist, i32 *order, i64 *cnt) {
    memset(cnt, 0, sizeof(i64) * (nb + 1));
    for (i64 j = 0; j < G; ++j) {
        int d = 0;
        for (int w = 0; w < W; ++w) d += popc64(q[w] ^ g[j * W + w]);
        dist[j] = d;
        cnt[d + 1]++;
    }
    for (int d = 0; d < nb; ++d) cnt[d + 1] += cnt[d];
    for (i64 j = 0; j < G; ++j) order[cnt[dist[j]]++] = (i32)j;
}

/* top-k: idx/dist [Qn,k]; entries beyond G are (-1, -1) */
void ho_topk(const u64 *q, const u64 *g, i64 Qn, i64 G, int W, int k, i32 *out_idx, i32 *out_dist) {
    int nb = 64 * W + 1;
    i32 *dist = (i32 *)malloc(sizeof(i32) * (G > 0 ? G : 1));
    i32 *order = (i32 *)malloc(sizeof(i32) * (G > 0 ? G : 1));
    i64 *cnt = (i64 *)malloc(sizeof(i64) * (nb + 1));
    for (i64 i = 0; i < Qn; ++i) {
        rank_one(q + i * W, g, G, W, nb, dist, order, cnt);
        for (int r = 0; r < k; ++r) {
            if (r < G) {
                out_idx[i * k + r] = order[r];
                out_dist[i * k + r] = dist[order[r]];
            } else {
                out_idx[i * k + r] = -1;
                out_dist[i * k + r] = -1;
            }
        }
    }
    free(dist);
    free(order);
    free(cnt);
}

/* relevance: single-label (LW == 0: labels are int32 class ids stored in the low half of each u64 slot is NOT
 * used -- pass lab32 arrays) or multi-hot bitmasks of LW words. */
static inline int relevant(const i32 *ql32, const i32 *gl32, const u64 *qlm, const u64 *glm, int LW, i64 qi, i64 gj) {
    if (LW == 0) return ql32[qi] == gl32[gj];
    for (int w = 0; w < LW; ++w)
        if (qlm[qi * LW + w] & glm[gj * LW + w]) return 1;
    return 0;
}

/*
 * mAP@R with P@k / R@k.
 *   out_S[Qn]      fixed-point AP numerator (u64)
 *   out_nrel[Qn]   #relevant within top-R (u32)
 *   out_ap[Qn]     float64 AP, reference-style sequential mean(count / tindex)
 *   out_hits[Qn,nk] #relevant within top-k for each k in ks
 *   out_total[Qn]  #relevant in the whole (possibly first-removed) ranking
 *   out_hist[Qn,nb,2] per-distance-bucket (count, relevant) -- what HIP pass 1 must reproduce (over full gallery,
 *                     before remove_first)
 */
void ho_map(const u64 *q, const u64 *g, const i32 *ql32, const i32 *gl32, const u64 *qlm, const u64 *glm, int LW,
            i64 Qn, i64 G, int W, i64 R, int remove_first, const i32 *ks, int nk, u64 *out_S, u32 *out_nrel,
            double *out_ap, u32 *out_hits, u32 *out_total, u32 *out_hist) {
    int nb = 64 * W + 1;
    i32 *dist = (i32 *)malloc(sizeof(i32) * (G > 0 ? G : 1));
    i32 *order = (i32 *)malloc(sizeof(i32) * (G > 0 ? G : 1));
    i64 *cnt = (i64 *)malloc(sizeof(i64) * (nb + 1));
    for (i64 i = 0; i < Qn; ++i) {
        rank_one(q + i * W, g, G, W, nb, dist, order, cnt);
        if (out_hist) {
            u32 *h = out_hist + i * nb * 2;
            memset(h, 0, sizeof(u32) * nb * 2);
            for (i64 j = 0; j < G; ++j) {
                h[dist[j] * 2]++;
                if (relevant(ql32, gl32, qlm, glm, LW, i, j)) h[dist[j] * 2 + 1]++;
            }
        }
        i64 start = remove_first ? 1 : 0;
        i64 n = G - start;
        if (n < 0) n = 0;
        i64 Rr = (R <= 0 || R > n) ? n : R;
        u64 S = 0;
        u32 nrel = 0, total = 0;
        double apsum = 0.0;
        for (int t = 0; t < nk; ++t) out_hits[i * nk + t] = 0;
        for (i64 r = 1; r <= n; ++r) {
            i64 j = order[start + r - 1];
            int rel = relevant(ql32, gl32, qlm, glm, LW, i, j);
            if (!rel) continue;
            total++;
            if (r <= Rr) {
                nrel++;
                S += (((u64)nrel) << 32) / (u64)r;
                apsum += (double)nrel / (double)r;
            }
            for (int t = 0; t < nk; ++t)
                if (r <= ks[t]) out_hits[i * nk + t]++;
        }
        out_S[i] = S;
        out_nrel[i] = nrel;
        out_total[i] = total;
        out_ap[i] = nrel ? apsum / (double)nrel : 0.0;
    }
    free(dist);
    free(order);
    free(cnt);
}

/* Bounded-work CPU baseline leg: distances + top-k for a sample of queries; returns a checksum so the
 * compiler cannot drop the work. */
u64 ho_bench_topk(const u64 *q, const u64 *g, i64 Qn, i64 G, int W, int k) {
    i32 *idx = (i32 *)malloc(sizeof(i32) * Qn * k);
    i32 *dst = (i32 *)malloc(sizeof(i32) * Qn * k);
    ho_topk(q, g, Qn, G, W, k, idx, dst);
    u64 s = 0;
    for (i64 i = 0; i < Qn * k; ++i) s += (u64)(u32)idx[i] * 31u + (u64)(u32)dst[i];
    free(idx);
    free(dst);
    return s;
}
