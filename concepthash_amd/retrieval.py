"""Packed-code Hamming retrieval on the GPU through the C-ABI: pack, distance, top-k, mAP / P@k / R@k, and the
gallery-sharded multi-GPU variants (one process per GPU, RCCL via ``torch.distributed``).

Replaces the reference's un-vendored ``utils.hashing`` arithmetic (call sites experiments/test_hashing.py:106-119,
153-162).  PyTorch supplies device buffers, streams and collectives; all per-pair arithmetic runs in
``csrc/hamming.hip``.  Packed codes are ``int64`` tensors holding the uint64 bit patterns ([rows, W]).
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence, Tuple

import torch

from . import _lib

TWO32 = 4294967296.0


def _dev_guard(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("retrieval kernels need GPU tensors (MI355X); there is no CPU fallback")
    return torch.cuda.device(t.device)


def pack_sign(codes: torch.Tensor, threshold: float = 0.0, stream=None) -> torch.Tensor:
    """[rows, nbit] fp32 -> [rows, ceil(nbit/64)] int64; bit i = (codes[:, i] - threshold) > 0 (little endian)."""
    lib = _lib.load()
    if codes.dim() != 2:
        raise ValueError("codes must be 2-D")
    codes = codes.to(torch.float32).contiguous()
    rows, nbit = codes.shape
    out = torch.empty(rows, (nbit + 63) // 64, dtype=torch.int64, device=codes.device)
    with _dev_guard(codes):
        _lib.check(lib.ch_pack_sign(_lib.ptr(codes), rows, nbit, float(threshold), _lib.ptr(out), _lib.stream_ptr(stream)),
                   "ch_pack_sign")
    return out


def _check_packed(q: torch.Tensor, g: torch.Tensor):
    if q.dtype != torch.int64 or g.dtype != torch.int64 or q.dim() != 2 or g.dim() != 2:
        raise TypeError("packed codes must be 2-D int64 tensors")
    if q.shape[1] != g.shape[1]:
        raise ValueError(f"query has {q.shape[1]} words per code, gallery {g.shape[1]}")
    if q.device != g.device:
        raise ValueError("query and gallery codes must be on the same device")
    return q.contiguous(), g.contiguous()


def hamming_dist(q: torch.Tensor, g: torch.Tensor, stream=None) -> torch.Tensor:
    """Full [Qn, G] int32 distance matrix (small problems; ``get_hamm_dist`` semantics without normalisation)."""
    lib = _lib.load()
    q, g = _check_packed(q, g)
    out = torch.empty(q.shape[0], g.shape[0], dtype=torch.int32, device=q.device)
    with _dev_guard(q):
        _lib.check(lib.ch_hamming_dist(_lib.ptr(q), q.shape[0], _lib.ptr(g), g.shape[0], q.shape[1], _lib.ptr(out),
                                       _lib.stream_ptr(stream)), "ch_hamming_dist")
    return out


def hamming_topk(q: torch.Tensor, g: torch.Tensor, k: int, g_index_base: int = 0, stream=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Top-k by ascending (distance, gallery index): (idx int64 [Qn,k], dist int32 [Qn,k]); -1 where k > G."""
    lib = _lib.load()
    q, g = _check_packed(q, g)
    Qn, W = q.shape
    G = g.shape[0]
    idx = torch.empty(Qn, k, dtype=torch.int64, device=q.device)
    dist = torch.empty(Qn, k, dtype=torch.int32, device=q.device)
    wsb = int(lib.ch_hamming_topk_workspace(Qn, G, W, k))
    ws = torch.empty(wsb, dtype=torch.uint8, device=q.device)
    with _dev_guard(q):
        _lib.check(lib.ch_hamming_topk(_lib.ptr(q), Qn, _lib.ptr(g), G, W, k, int(g_index_base), _lib.ptr(idx),
                                       _lib.ptr(dist), _lib.ptr(ws), wsb, _lib.stream_ptr(stream)), "ch_hamming_topk")
    return idx, dist


def topk_merge(idx_lists: torch.Tensor, dist_lists: torch.Tensor, stream=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """[nlists, Qn, k] per-shard lists -> global [Qn, k]."""
    lib = _lib.load()
    idx_lists = idx_lists.contiguous()
    dist_lists = dist_lists.contiguous()
    n, Qn, k = idx_lists.shape
    idx = torch.empty(Qn, k, dtype=torch.int64, device=idx_lists.device)
    dist = torch.empty(Qn, k, dtype=torch.int32, device=idx_lists.device)
    with _dev_guard(idx_lists):
        _lib.check(lib.ch_topk_merge(_lib.ptr(idx_lists), _lib.ptr(dist_lists), n, Qn, k, _lib.ptr(idx), _lib.ptr(dist),
                                     _lib.stream_ptr(stream)), "ch_topk_merge")
    return idx, dist


# ---------------------------------------------------------------------------------------------------------------
# labels
# ---------------------------------------------------------------------------------------------------------------
def labels_single(q_labels: torch.Tensor, g_labels: torch.Tensor) -> bool:
    """2-D indicator matrices whose rows all hold exactly one class (a gallery SHARD decides for its own rows only: the ranks of a
    sharded evaluation combine their answers, concepthash_amd.distributed)."""
    return bool(((q_labels != 0).sum(1) == 1).all().item()) and bool(((g_labels != 0).sum(1) == 1).all().item())


def prepare_labels(q_labels: torch.Tensor, g_labels: torch.Tensor, single=None):
    """Returns (q_lab, g_lab, LW).  1-D integer labels, or one-hot rows with exactly one class each -> int32 ids
    (LW = 0); otherwise multi-hot -> int64 bitmasks [rows, LW].  `single`: the decision, when the caller has made it (globally)."""
    if q_labels.dim() == 1 and g_labels.dim() == 1:
        return q_labels.to(torch.int32).contiguous(), g_labels.to(torch.int32).contiguous(), 0
    if q_labels.dim() != 2 or g_labels.dim() != 2 or q_labels.shape[1] != g_labels.shape[1]:
        raise ValueError("labels must both be 1-D class ids or 2-D [rows, C] indicator matrices")
    qb, gb = q_labels != 0, g_labels != 0
    if single is None:
        single = labels_single(q_labels, g_labels)
    if single:
        return qb.int().argmax(1).to(torch.int32).contiguous(), gb.int().argmax(1).to(torch.int32).contiguous(), 0
    C = qb.shape[1]
    LW = (C + 63) // 64

    def pack(b):
        pad = LW * 64 - C
        if pad:
            b = torch.cat([b, torch.zeros(b.shape[0], pad, dtype=torch.bool, device=b.device)], dim=1)
        w = torch.ones(64, dtype=torch.int64, device=b.device) << torch.arange(64, dtype=torch.int64, device=b.device)
        return (b.view(b.shape[0], LW, 64).to(torch.int64) * w).sum(-1).contiguous()  # wraps mod 2^64: bit 63 ok

    return pack(qb), pack(gb), LW


def map_seg_rows(Qn: int, G: int, W: int) -> int:
    """Gallery rows per segment of the two mAP passes (grid = query tiles x segments).  A workgroup holds its tile's counters in
    LDS -- (64 W + 1) x BLK x 4 B -- so an MI355X has 256 (x 2 at 64 bit) workgroup slots, and a launch takes
    ceil(workgroups / slots) rounds of one segment each: the segment count is chosen to fill whole rounds (e.g. NABirds size,
    97 tiles: 10 segments = 1.9 rounds instead of 11 = 2.1 -> 3) at about 1024 workgroups, segments of 256 .. 65,535 rows."""
    blk = 256 if W <= 2 else 128
    tiles = max(1, -(-Qn // blk))
    slots = 256 * max(1, (160 * 1024) // ((64 * W + 1) * blk * 4))
    nseg_max = max(1, min(65535, G // 256))          # segments of at least 256 rows
    nseg_min = max(1, -(-G // 65535))                # ... and at most 65,535 (16-bit counters)
    want = max(1, -(-1024 // tiles))
    hi = max(nseg_min, min(nseg_max, 2 * want))
    lo = min(hi, max(nseg_min, min(want, hi) // 2))
    best, best_cost = None, None
    for nseg in range(lo, hi + 1):
        rows = -(-G // nseg)
        cost = -(-(tiles * nseg) // slots) * (rows + 128)     # rounds x (rows of a segment + a workgroup's fixed cost in row units)
        if best_cost is None or cost < best_cost:
            best, best_cost = nseg, cost
    rows = -(-G // best)
    return int(min(65535, max(256, rows)))


def hamming_hist(q, g, q_lab, g_lab, LW: int, seg_rows: int, stream=None) -> torch.Tensor:
    """mAP pass 1: [nseg, Qn, 64W+1, 2] int32 (uint32 bit patterns)."""
    lib = _lib.load()
    q, g = _check_packed(q, g)
    Qn, W = q.shape
    G = g.shape[0]
    nseg = max(1, -(-G // seg_rows))
    # the pass writes every counter of every (segment, query); an empty gallery launches nothing
    hist = (torch.empty if G > 0 else torch.zeros)(nseg, Qn, 64 * W + 1, 2, dtype=torch.int32, device=q.device)
    with _dev_guard(q):
        _lib.check(lib.ch_hamming_hist(_lib.ptr(q), Qn, _lib.ptr(g), G, W, _lib.ptr(q_lab), _lib.ptr(g_lab), LW, seg_rows,
                                       _lib.ptr(hist), _lib.stream_ptr(stream)), "ch_hamming_hist")
    return hist


def hist_prefix(hist: torch.Tensor, stream=None) -> Tuple[torch.Tensor, torch.Tensor]:
    lib = _lib.load()
    hist = hist.contiguous()
    nseg, Qn, nb, _ = hist.shape
    base = torch.empty_like(hist)
    totals = torch.empty(Qn, 2, dtype=torch.int32, device=hist.device)
    with _dev_guard(hist):
        _lib.check(lib.ch_hamming_hist_prefix(_lib.ptr(hist), nseg, Qn, nb, _lib.ptr(base), _lib.ptr(totals),
                                              _lib.stream_ptr(stream)), "ch_hamming_hist_prefix")
    return base, totals


def hamming_ap(q, g, q_lab, g_lab, LW: int, seg_rows: int, base: torch.Tensor, rank_limit: int = -1,
               first_rel: Optional[torch.Tensor] = None, out_S: Optional[torch.Tensor] = None,
               out_nrel: Optional[torch.Tensor] = None, stream=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """mAP pass 2 -> (S int64 [Qn] fixed-point numerators, nrel int32 [Qn])."""
    lib = _lib.load()
    q, g = _check_packed(q, g)
    Qn, W = q.shape
    G = g.shape[0]
    S = out_S if out_S is not None else torch.zeros(Qn, dtype=torch.int64, device=q.device)
    nrel = out_nrel if out_nrel is not None else torch.zeros(Qn, dtype=torch.int32, device=q.device)
    if first_rel is not None:
        first_rel = first_rel.to(torch.int32).contiguous()
    with _dev_guard(q):
        _lib.check(lib.ch_hamming_ap(_lib.ptr(q), Qn, _lib.ptr(g), G, W, _lib.ptr(q_lab), _lib.ptr(g_lab), LW, seg_rows,
                                     _lib.ptr(base.contiguous()), int(rank_limit), _lib.ptr(first_rel), _lib.ptr(S),
                                     _lib.ptr(nrel), _lib.stream_ptr(stream)), "ch_hamming_ap")
    return S, nrel


def _relevance_of(idx: torch.Tensor, q_lab: torch.Tensor, g_lab_all: torch.Tensor, LW: int) -> torch.Tensor:
    """[Qn,k] bool relevance of retrieved rows (label gather; idx -1 -> False)."""
    safe = idx.clamp_min(0)
    if LW == 0:
        rel = g_lab_all[safe] == q_lab[:, None]
    else:
        rel = (g_lab_all[safe] & q_lab[:, None, :]).ne(0).any(-1)
    return rel & (idx >= 0)


MAX_LIMITS = 16   # rank limits one AP pass accumulates (csrc/hamming.hip)


def normalize_limits(limits: Sequence[int]):
    """-> (ascending unique limits with "unlimited" (<= 0) last as 0, index of every input limit in that list)"""
    norm = [int(r) if int(r) > 0 else 0 for r in limits]
    uniq = sorted({r for r in norm if r > 0}) + ([0] if 0 in norm else [])
    pos = {r: i for i, r in enumerate(uniq)}
    return uniq, [pos[r] for r in norm]


def hamming_ap_multi(q, g, q_lab, g_lab, LW: int, seg_rows: int, base: torch.Tensor, rank_limits: Sequence[int],
                     first_rel: Optional[torch.Tensor] = None, stream=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """mAP pass 2 for a list of ascending rank limits (<= 0 = unlimited, last) in ONE gallery scan per 16 limits
    -> (S int64 [n, Qn], nrel int32 [n, Qn]); nrel[i] = number of relevant rows inside limit i."""
    lib = _lib.load()
    q, g = _check_packed(q, g)
    Qn, W = q.shape
    G = g.shape[0]
    lim = [int(r) for r in rank_limits]
    n = len(lim)
    S = torch.zeros(n, Qn, dtype=torch.int64, device=q.device)
    nrel = torch.zeros(n, Qn, dtype=torch.int32, device=q.device)
    if first_rel is not None:
        first_rel = first_rel.to(torch.int32).contiguous()
    base = base.contiguous()
    with _dev_guard(q):
        for c0 in range(0, n, MAX_LIMITS):
            chunk = lim[c0:c0 + MAX_LIMITS]
            arr = (ctypes.c_int64 * len(chunk))(*chunk)
            _lib.check(lib.ch_hamming_ap_multi(_lib.ptr(q), Qn, _lib.ptr(g), G, W, _lib.ptr(q_lab), _lib.ptr(g_lab), LW, seg_rows,
                                               _lib.ptr(base), arr, len(chunk), _lib.ptr(first_rel), _lib.ptr(S[c0:c0 + len(chunk)]),
                                               _lib.ptr(nrel[c0:c0 + len(chunk)]), _lib.stream_ptr(stream)), "ch_hamming_ap_multi")
    return S, nrel


REC_BUDGET_BYTES = 4 << 30     # upper bound of the record buffer of one evaluation
REC_COMFORT_BYTES = 1 << 30    # ... and what is spent without need, for long lists (class-sorted galleries)


def record_cap(Qn: int, G: int, W: int, seg_rows: int) -> int:
    """Entries per (query, segment) record list of the one-scan form.  At least room for one relevant row in 64 of a segment (+ 32:
    any label distribution with no class above ~1.5 % of a segment fits), and as much more -- up to 2,048 entries -- as
    REC_COMFORT_BYTES buys: galleries listed class by class put a whole class (hundreds of rows) into one segment's lists.  Never
    above the segment length or REC_BUDGET_BYTES for the whole buffer.  A (tile, segment) workgroup whose lists overflow is redone
    by the two-scan kernel, so this only decides speed."""
    lib = _lib.load()
    wgs = int(lib.ch_hamming_rec_workgroups(Qn, G, W, seg_rows))
    unit = max(1, wgs * int(lib.ch_hamming_rec_block(W)) * 8)          # bytes per list entry over all lists
    cap = max(seg_rows // 64 + 32, min(2048, REC_COMFORT_BYTES // unit))
    return int(max(1, min(cap, seg_rows, REC_BUDGET_BYTES // unit)))


def records_fit(Qn: int, G: int, W: int, seg_rows: int) -> bool:
    """False where REC_BUDGET_BYTES cannot even hold the minimum list length (very large Qn x G): the lists would overflow nearly
    everywhere and the recording scan would be paid on top of the two-scan form -- evaluate() then runs the two-scan form."""
    return record_cap(Qn, G, W, seg_rows) >= min(seg_rows, seg_rows // 64 + 32)


OVERFLOW_SWITCH = 0.30     # predicted fraction of overflowing (query tile, segment) workgroups above which two scans are faster
OVERFLOW_MIN_PAIRS = 1 << 31   # below this many (query, row) pairs the whole evaluation is < 1 ms: not worth a prediction


def predicted_overflow(q_lab: torch.Tensor, g_lab: torch.Tensor, W: int, seg_rows: int, cap: int) -> float:
    """Fraction of the recording scan's (query tile, gallery segment) workgroups in which some query's record list would exceed
    `cap` entries -- EXACT for single-label ids, from the per-segment class histogram of the gallery labels (a few launches on
    G + Qn x nseg elements; one host read).  A list overflows when a query's class has more than `cap` rows inside one segment:
    galleries listed class by class under shuffled queries do that in nearly every workgroup, and there the one-scan form (which
    redoes every overflowed workgroup with the two-scan kernel) loses to running two scans outright: 81 vs 65 ms at 16,384 x 1M x
    128 bit."""
    lib = _lib.load()
    G, Qn = g_lab.shape[0], q_lab.shape[0]
    if G == 0 or Qn == 0 or cap >= seg_rows:
        return 0.0
    nseg = -(-G // seg_rows)
    blk = int(lib.ch_hamming_rec_block(W))
    ncls = int(torch.maximum(q_lab.max(), g_lab.max()).item()) + 1
    seg_id = torch.arange(G, device=g_lab.device, dtype=torch.int64) // seg_rows
    counts = torch.bincount(seg_id * ncls + g_lab.to(torch.int64), minlength=nseg * ncls).view(nseg, ncls)
    over = counts[:, q_lab.to(torch.int64)] > cap                               # [nseg, Qn]
    pad = (-Qn) % blk
    if pad:
        over = torch.cat([over, torch.zeros(nseg, pad, dtype=torch.bool, device=over.device)], dim=1)
    return float(over.view(nseg, -1, blk).any(dim=2).float().mean().item())


def hamming_hist_rec(q, g, q_lab, g_lab, LW: int, seg_rows: int, rec_cap: Optional[int] = None, stream=None):
    """mAP pass 1 of the one-scan form: the histogram of hamming_hist plus the per-lane record lists of the relevant rows
    -> (hist, records) where records = (rec, rec_cap, rec_cnt, wg_flags) is what hamming_ap_rec takes."""
    lib = _lib.load()
    q, g = _check_packed(q, g)
    Qn, W = q.shape
    G = g.shape[0]
    nseg = max(1, -(-G // seg_rows))
    cap = int(rec_cap) if rec_cap else record_cap(Qn, G, W, seg_rows)
    wgs = int(lib.ch_hamming_rec_workgroups(Qn, G, W, seg_rows))
    blk = int(lib.ch_hamming_rec_block(W))
    hist = (torch.empty if G > 0 else torch.zeros)(nseg, Qn, 64 * W + 1, 2, dtype=torch.int32, device=q.device)
    rec = torch.empty(max(1, wgs) * cap * blk, 2, dtype=torch.int32, device=q.device)
    rec_cnt = torch.empty(nseg, Qn, dtype=torch.int32, device=q.device)
    wg_flags = torch.zeros(max(1, wgs), dtype=torch.int32, device=q.device)
    with _dev_guard(q):
        _lib.check(lib.ch_hamming_hist_rec(_lib.ptr(q), Qn, _lib.ptr(g), G, W, _lib.ptr(q_lab), _lib.ptr(g_lab), LW, seg_rows,
                                           _lib.ptr(hist), _lib.ptr(rec), cap, _lib.ptr(rec_cnt), _lib.ptr(wg_flags),
                                           _lib.stream_ptr(stream)), "ch_hamming_hist_rec")
    return hist, (rec, cap, rec_cnt, wg_flags)


def hamming_ap_rec(q, g, q_lab, g_lab, LW: int, seg_rows: int, base: torch.Tensor, records, rank_limits: Sequence[int],
                   first_rel: Optional[torch.Tensor] = None, stream=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """hamming_ap_multi from the records of hamming_hist_rec (no second distance scan except for overflowed workgroups)."""
    lib = _lib.load()
    q, g = _check_packed(q, g)
    Qn, W = q.shape
    G = g.shape[0]
    rec, cap, rec_cnt, wg_flags = records
    lim = [int(r) for r in rank_limits]
    n = len(lim)
    S = torch.zeros(n, Qn, dtype=torch.int64, device=q.device)
    nrel = torch.zeros(n, Qn, dtype=torch.int32, device=q.device)
    if first_rel is not None:
        first_rel = first_rel.to(torch.int32).contiguous()
    base = base.contiguous()
    with _dev_guard(q):
        for c0 in range(0, n, MAX_LIMITS):
            chunk = lim[c0:c0 + MAX_LIMITS]
            arr = (ctypes.c_int64 * len(chunk))(*chunk)
            _lib.check(lib.ch_hamming_ap_rec(_lib.ptr(q), Qn, _lib.ptr(g), G, W, _lib.ptr(q_lab), _lib.ptr(g_lab), LW, seg_rows,
                                             _lib.ptr(base), _lib.ptr(rec), cap, _lib.ptr(rec_cnt), _lib.ptr(wg_flags), arr,
                                             len(chunk), _lib.ptr(first_rel), _lib.ptr(S[c0:c0 + len(chunk)]),
                                             _lib.ptr(nrel[c0:c0 + len(chunk)]), _lib.stream_ptr(stream)), "ch_hamming_ap_rec")
    return S, nrel


def ap_from_fixed(S: torch.Tensor, nrel: torch.Tensor) -> torch.Tensor:
    """AP[q] = S / (nrel * 2^32) in float64 (0 where nrel == 0); S holds uint64 bit patterns."""
    Sf = S.to(torch.float64)
    Sf = torch.where(S < 0, Sf + 18446744073709551616.0, Sf)
    n = nrel.to(torch.float64)
    return torch.where(nrel > 0, Sf / (n.clamp_min(1.0) * TWO32), torch.zeros_like(Sf))


def summarize(S, nrel, total, idx_of, Rs: Sequence[int], ks: Sequence[int], skip_queries_without_relevant: bool = False) -> dict:
    """Host-side statistics from the per-limit integers of one multi-limit AP pass: limits were Rs + ks (idx_of maps each to
    its row of S / nrel).  mAP per R; P@k = hits / k and R@k = hits / total with hits = nrel under limit k.
    skip_queries_without_relevant: the ONE convention of the AP definition that changes numbers ("Retrieval
    definition"): a query with no relevant row inside its top R has AP = 0.  False (default): it counts in the
    mean with that 0.  True (the HashNet / OrthoHash-family convention, `if tsum == 0: continue`): it is left out of the mean --
    mAP@R = mean of AP over the queries that have a relevant row in their top R (0.0 when no query has one).  Identical whenever
    every query has a relevant row inside R, e.g. mAP@all on CUB-200 / Cars196 / NABirds."""
    Qn = total.shape[0]
    dev = total.device
    nR = len(Rs)
    aps = [ap_from_fixed(S[idx_of[i]], nrel[idx_of[i]]) for i in range(nR)]
    hits = torch.zeros(Qn, len(ks), dtype=torch.int32, device=dev)
    if not Qn:
        return dict(mAPs=[0.0] * nR, aps=aps, hits=hits, precisions=[0.0] * len(ks), recalls=[0.0] * len(ks))
    if skip_queries_without_relevant:
        means = []
        for i, a in enumerate(aps):
            has = nrel[idx_of[i]] > 0
            means.append(a.sum() / has.sum().clamp_min(1).double())    # a is 0 where nrel == 0
    else:
        means = [a.mean() for a in aps]                     # 0-dim float64 tensors; ONE device->host copy for all of them below
    tot = total.clamp_min(1).double()
    for t, k in enumerate(ks):
        h = nrel[idx_of[nR + t]]
        hits[:, t] = h
        means.append((h.double() / k).mean())
    for t in range(len(ks)):
        h = hits[:, t]
        means.append(torch.where(total > 0, h.double() / tot, torch.zeros_like(tot)).mean())
    vals = torch.stack(means).tolist()
    return dict(mAPs=vals[:nR], aps=aps, hits=hits, precisions=vals[nR:nR + len(ks)], recalls=vals[nR + len(ks):])


def evaluate(q: torch.Tensor, g: torch.Tensor, q_labels: torch.Tensor, g_labels: torch.Tensor, R=-1,
             ks: Sequence[int] = (1, 5, 10), remove_first: bool = False, seg_rows: Optional[int] = None,
             records: Optional[bool] = None, rec_cap: Optional[int] = None, skip_queries_without_relevant: bool = False) -> dict:
    """Single-GPU mAP@R + P@k + R@k on packed codes: histogram pass, prefix, ONE AP pass whose rank limits are R (an int or
    a list) and every k -- the number of relevant rows inside limit k is exactly hits@k, for any k.  Returns python
    floats/lists plus the raw integer statistics (S, nrel, hits, total) that the parity tests compare bit-for-bit with the
    oracle.  With a list R: mAP, S, nrel, ap are lists (one entry per R).
    records (default on; CH_HAMMING_RECORDS=0 = off): the one-scan form -- the histogram pass also records the relevant rows and
    the AP pass walks those records instead of scanning the gallery again; rec_cap overrides the list capacity (tests).  Left to the
    default, large single-label problems first predict (exactly, from the labels) how many workgroups' lists would overflow and run
    the two-scan form where most would (`predicted_overflow`).
    skip_queries_without_relevant: which queries the mean of AP runs over (`summarize`); the integers S / nrel do not depend on it."""
    q, g = _check_packed(q, g)
    Qn, W = q.shape
    G = g.shape[0]
    q_lab, g_lab, LW = prepare_labels(q_labels.to(q.device), g_labels.to(q.device))
    seg = seg_rows or map_seg_rows(Qn, G, W)
    ks = [int(k) for k in ks]
    if any(k <= 0 for k in ks):
        raise ValueError("P@k / R@k need k >= 1")
    many = isinstance(R, (list, tuple))
    Rs = [int(r) for r in R] if many else [int(R)]
    dev = q.device
    if Qn == 0 or G == 0:
        z64 = torch.zeros(Qn, dtype=torch.int64, device=dev)
        z32 = torch.zeros(Qn, dtype=torch.int32, device=dev)
        out = dict(mAP=0.0, precisions=[0.0] * len(ks), recalls=[0.0] * len(ks), S=z64, nrel=z32,
                   hits=torch.zeros(Qn, len(ks), dtype=torch.int32, device=dev), total=z32, ap=z64.double())
        if many:
            out.update(mAP=[0.0] * len(Rs), S=[z64] * len(Rs), nrel=[z32] * len(Rs), ap=[z64.double()] * len(Rs))
        return out
    first_rel = None
    if remove_first:   # relevance of every query's rank-1 row (the self-match when the test set is the database)
        idx, _ = hamming_topk(q, g, 1)
        first_rel = _relevance_of(idx, q_lab, g_lab, LW)[:, 0].to(torch.int32)
    limits, idx_of = normalize_limits(Rs + ks)
    if records is None:
        records = os.environ.get("CH_HAMMING_RECORDS", "1") != "0" and (rec_cap is not None or records_fit(Qn, G, W, seg))
        if records and rec_cap is None and LW == 0 and Qn * G >= OVERFLOW_MIN_PAIRS:
            # the label layout decides which form is faster (never the result): lists that would overflow almost everywhere -> two scans
            records = predicted_overflow(q_lab, g_lab, W, seg, record_cap(Qn, G, W, seg)) <= OVERFLOW_SWITCH
    if records:   # one distance scan: histogram + records of the relevant rows, prefix, then the AP terms from the records
        hist, recs = hamming_hist_rec(q, g, q_lab, g_lab, LW, seg, rec_cap=rec_cap)
        base, totals = hist_prefix(hist)
        S, nrel = hamming_ap_rec(q, g, q_lab, g_lab, LW, seg, base, recs, limits, first_rel=first_rel)
    else:         # two scans (the round-1 / early round-2 form; also what the sharded evaluator runs)
        hist = hamming_hist(q, g, q_lab, g_lab, LW, seg)
        base, totals = hist_prefix(hist)
        S, nrel = hamming_ap_multi(q, g, q_lab, g_lab, LW, seg, base, limits, first_rel=first_rel)
    total = totals[:, 1].clone()
    if remove_first:
        total = total - first_rel
    sm = summarize(S, nrel, total, idx_of, Rs, ks, skip_queries_without_relevant)
    out = dict(precisions=sm["precisions"], recalls=sm["recalls"], hits=sm["hits"], total=total)
    if many:
        out.update(mAP=sm["mAPs"], S=[S[idx_of[i]] for i in range(len(Rs))], nrel=[nrel[idx_of[i]] for i in range(len(Rs))],
                   ap=sm["aps"])
    else:
        out.update(mAP=sm["mAPs"][0], S=S[idx_of[0]], nrel=nrel[idx_of[0]], ap=sm["aps"][0])
    return out
