"""CPU stand-in for `concepthash_amd.retrieval` used ONLY by the gloo tests of the multi-GPU choreography
(concepthash_amd/distributed.py): same function names and tensor contracts, computed with numpy + the C oracle."""
import numpy as np
import torch

from concepthash_amd.retrieval import (ap_from_fixed, labels_single, map_seg_rows, normalize_limits,  # pure-torch host helpers
                                       prepare_labels, summarize)
from oracle import hamming_oracle as ho

__all__ = ["hamming_topk", "topk_merge", "prepare_labels", "labels_single", "hamming_hist", "hist_prefix", "hamming_ap", "hamming_ap_multi",
           "map_seg_rows", "ap_from_fixed", "normalize_limits", "summarize"]


def _u(t):
    return np.ascontiguousarray(t.cpu().numpy()).view(np.uint64)


def hamming_topk(q, g, k, g_index_base=0, stream=None):
    idx, dst = ho.topk(_u(q), _u(g), k)
    idx = idx.astype(np.int64)
    idx[idx >= 0] += g_index_base
    return torch.from_numpy(idx), torch.from_numpy(dst)


def topk_merge(idx_lists, dist_lists, stream=None):
    n, Qn, k = idx_lists.shape
    idx = idx_lists.permute(1, 0, 2).reshape(Qn, n * k).numpy()
    dst = dist_lists.permute(1, 0, 2).reshape(Qn, n * k).numpy().astype(np.int64)
    key = np.where(dst < 0, np.iinfo(np.int64).max, dst * (1 << 40) + idx)
    order = np.argsort(key, axis=1, kind="stable")[:, :k]
    oi, od = np.take_along_axis(idx, order, 1), np.take_along_axis(dst, order, 1).astype(np.int32)
    oi[od < 0] = -1
    return torch.from_numpy(oi), torch.from_numpy(od)


def _rel(q_lab, g_lab, LW):
    if LW == 0:
        return q_lab.numpy()[:, None] == g_lab.numpy()[None, :]
    return (q_lab.numpy()[:, None, :] & g_lab.numpy()[None, :, :]).any(-1)


def hamming_hist(q, g, q_lab, g_lab, LW, seg_rows, stream=None):
    Qn, W = q.shape
    G = g.shape[0]
    nb = 64 * W + 1
    nseg = max(1, -(-G // seg_rows))
    hist = np.zeros((nseg, Qn, nb, 2), dtype=np.int32)
    if G and Qn:
        d = ho.dist(_u(q), _u(g))
        rel = _rel(q_lab, g_lab, LW)
        for s in range(nseg):
            sl = slice(s * seg_rows, min(G, (s + 1) * seg_rows))
            for i in range(Qn):
                hist[s, i, :, 0] = np.bincount(d[i, sl], minlength=nb)
                hist[s, i, :, 1] = np.bincount(d[i, sl][rel[i, sl]], minlength=nb)
    return torch.from_numpy(hist)


def hist_prefix(hist, stream=None):
    h = hist.numpy().astype(np.int64)                       # [nseg, Qn, nb, 2]
    nseg, Qn, nb, _ = h.shape
    flat = h.transpose(1, 3, 2, 0).reshape(Qn, 2, nb * nseg)   # ranking order: bucket major, segment minor
    ex = np.cumsum(flat, axis=-1) - flat
    base = ex.reshape(Qn, 2, nb, nseg).transpose(3, 0, 2, 1)
    totals = flat.sum(-1)
    return torch.from_numpy(np.ascontiguousarray(base).astype(np.int32)), torch.from_numpy(totals.astype(np.int32))


def hamming_ap(q, g, q_lab, g_lab, LW, seg_rows, base, rank_limit=-1, first_rel=None, out_S=None, out_nrel=None, stream=None):
    Qn, W = q.shape
    G = g.shape[0]
    S = np.zeros(Qn, dtype=np.uint64)
    nrel = np.zeros(Qn, dtype=np.int32)
    if G and Qn:
        d = ho.dist(_u(q), _u(g))
        rel = _rel(q_lab, g_lab, LW)
        b = base.numpy().astype(np.int64)
        for i in range(Qn):
            seen = {}
            for j in range(G):
                s, dj = j // seg_rows, int(d[i, j])
                pos, rpos = seen.get((s, dj), (0, 0))
                r = bool(rel[i, j])
                seen[(s, dj)] = (pos + 1, rpos + int(r))
                if not r:
                    continue
                rank, rr = int(b[s, i, dj, 0]) + pos + 1, int(b[s, i, dj, 1]) + rpos + 1
                if first_rel is not None:
                    if rank == 1:
                        continue
                    rank -= 1
                    rr -= int(first_rel[i])
                if rank_limit > 0 and rank > rank_limit:
                    continue
                S[i] += np.uint64((rr << 32) // rank)
                nrel[i] += 1
    return torch.from_numpy(S.view(np.int64).copy()), torch.from_numpy(nrel)


def hamming_ap_multi(q, g, q_lab, g_lab, LW, seg_rows, base, rank_limits, first_rel=None, stream=None):
    outs = [hamming_ap(q, g, q_lab, g_lab, LW, seg_rows, base, rank_limit=int(r), first_rel=first_rel) for r in rank_limits]
    return torch.stack([o[0] for o in outs]), torch.stack([o[1] for o in outs])
