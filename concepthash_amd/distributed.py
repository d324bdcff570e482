"""Gallery-sharded retrieval across the GPUs of one node (one process per GPU, ``torch.distributed``; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" is used by the CPU tests of the host logic).

SURVEY.md section 8e: the gallery is row-sharded where it was encoded (rank r owns a contiguous block of global
rows), packed query codes are all-gathered (KB-MB, latency bound), each rank scans only its shard, and the small
per-shard results are exchanged:
  * top-k:  all_gather of the per-shard (idx, dist) lists, k-way merge by (distance, global index);
  * mAP:    all_gather of per-shard histograms -> every rank computes the same global "ranked before" bases ->
            local pass 2 -> integer all_reduce of the AP numerators.  Bit-identical to the single-GPU result.

The per-shard compute is injected (``ops``): on GPUs it is ``concepthash_amd.retrieval`` (HIP kernels); the gloo
tests inject a CPU stand-in so that the collective choreography itself is covered without a GPU.
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import torch
import torch.distributed as dist


def shard_bounds(total_rows: int, world: int):
    """Contiguous row blocks: rank r owns [bounds[r], bounds[r+1])."""
    base, rem = divmod(total_rows, world)
    bounds = [0]
    for r in range(world):
        bounds.append(bounds[-1] + base + (1 if r < rem else 0))
    return bounds


def _all_gather_rows(t: torch.Tensor, group=None) -> torch.Tensor:
    """all_gather along dim 0 for equal-sized per-rank tensors."""
    world = dist.get_world_size(group)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    return out


def _all_gather_ragged(t: torch.Tensor, group=None):
    """all_gather along dim 0 when ranks hold different row counts; returns (concatenated, counts)."""
    world = dist.get_world_size(group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(counts) if counts else 0
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0), counts


class ShardedRetrieval:
    """Rank-local view of a row-sharded gallery.

    gallery:  packed codes of THIS rank's rows, int64 [G_local, W]
    labels:   labels of this rank's rows (1-D class ids or 2-D indicator matrix), optional
    ops:      module/object providing hamming_topk, topk_merge, prepare_labels, hamming_hist, hist_prefix,
              hamming_ap_multi, normalize_limits, summarize, map_seg_rows  (default: concepthash_amd.retrieval)
    """

    def __init__(self, gallery: torch.Tensor, labels: Optional[torch.Tensor] = None, group=None, ops=None):
        if not dist.is_initialized():
            raise RuntimeError("ShardedRetrieval needs an initialised torch.distributed process group")
        if ops is None:
            from . import retrieval as ops
        self.ops = ops
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.gallery = gallery.contiguous()
        self.labels = labels
        dev = gallery.device
        n = torch.tensor([gallery.shape[0]], dtype=torch.int64, device=dev)
        counts = [torch.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(counts, n, group=group)
        self.counts = [int(c.item()) for c in counts]
        self.base = sum(self.counts[: self.rank])       # global index of this shard's first row
        self.total = sum(self.counts)

    # ---- queries -------------------------------------------------------------------------------------------------
    def gather_queries(self, q_local: torch.Tensor) -> torch.Tensor:
        """Queries encoded on different ranks -> the full query set on every rank (rank-major order)."""
        allq, _ = _all_gather_ragged(q_local, self.group)
        return allq

    # ---- top-k ---------------------------------------------------------------------------------------------------
    def topk(self, q_all: torch.Tensor, k: int):
        """q_all: the SAME [Qn, W] on every rank.  Returns the global (idx int64 [Qn,k], dist int32 [Qn,k]) on every rank."""
        idx, dst = self.ops.hamming_topk(q_all, self.gallery, k, g_index_base=self.base)
        if self.world == 1:
            return idx, dst
        li = _all_gather_rows(idx.unsqueeze(0), self.group)   # [world, Qn, k]
        ld = _all_gather_rows(dst.unsqueeze(0), self.group)
        return self.ops.topk_merge(li, ld)

    # ---- mAP / P@k / R@k -----------------------------------------------------------------------------------------
    def evaluate(self, q_all: torch.Tensor, q_labels: torch.Tensor, R=-1, ks: Sequence[int] = (1, 5, 10),
                 remove_first: bool = False, seg_rows: Optional[int] = None) -> dict:
        """Same statistics as ``retrieval.evaluate`` (R an int or a list), gallery sharded by rows: per-shard histograms are
        all-gathered, every rank builds the same global bases, ONE local AP pass accumulates every rank limit (each R and each
        k), and the integer sums are all-reduced -- bit-identical to the single-GPU result for any shard count."""
        if self.labels is None:
            raise ValueError("gallery labels are required for evaluate()")
        ops = self.ops
        dev = q_all.device
        Qn, W = q_all.shape
        q_lab, g_lab, LW = ops.prepare_labels(q_labels.to(dev), self.labels.to(dev))
        ks = [int(k) for k in ks]
        if any(k <= 0 for k in ks):
            raise ValueError("P@k / R@k need k >= 1")
        many = isinstance(R, (list, tuple))
        Rs = [int(r) for r in R] if many else [int(R)]
        first_rel = None
        if remove_first:      # relevance of the global rank-1 row of every query
            idx, _ = self.topk(q_all, 1)
            g_lab_all, _ = _all_gather_ragged(g_lab, self.group) if self.world > 1 else (g_lab, None)
            safe = idx.clamp_min(0)
            if LW == 0:
                rel = (g_lab_all[safe] == q_lab[:, None]) & (idx >= 0)
            else:
                rel = (g_lab_all[safe] & q_lab[:, None, :]).ne(0).any(-1) & (idx >= 0)
            first_rel = rel[:, 0].to(torch.int32)
        # one segment size for all ranks, so the gathered histogram stack has one shape
        gmax = max(self.counts) if self.counts else 0
        seg = seg_rows or ops.map_seg_rows(Qn, max(gmax, 1), W)
        nseg = max(1, -(-gmax // seg))
        # one-scan form where the ops provide it (concepthash_amd.retrieval does): the histogram pass also records this shard's
        # relevant rows, and the AP terms come from the records once the GLOBAL bases exist (CH_HAMMING_RECORDS=0 = two scans)
        use_rec = (hasattr(ops, "hamming_hist_rec") and os.environ.get("CH_HAMMING_RECORDS", "1") != "0"
                   and ops.records_fit(Qn, max(gmax, 1), W, seg))
        if use_rec:
            hist, recs = ops.hamming_hist_rec(q_all, self.gallery, q_lab, g_lab, LW, seg)
        else:
            hist = ops.hamming_hist(q_all, self.gallery, q_lab, g_lab, LW, seg)        # [nseg_local, Qn, nb, 2]
        if hist.shape[0] < nseg:                                                       # shorter shard: pad with empty segments
            hist = torch.cat([hist, torch.zeros((nseg - hist.shape[0],) + tuple(hist.shape[1:]), dtype=hist.dtype,
                                                device=dev)], dim=0)
        hist_all = _all_gather_rows(hist, self.group) if self.world > 1 else hist      # [world*nseg, Qn, nb, 2], rank-major
        base_all, totals = ops.hist_prefix(hist_all)
        base = base_all[self.rank * nseg:(self.rank + 1) * nseg].contiguous()
        limits, idx_of = ops.normalize_limits(Rs + ks)
        if use_rec:
            S, nrel = ops.hamming_ap_rec(q_all, self.gallery, q_lab, g_lab, LW, seg, base, recs, limits, first_rel=first_rel)
        else:
            S, nrel = ops.hamming_ap_multi(q_all, self.gallery, q_lab, g_lab, LW, seg, base, limits, first_rel=first_rel)
        if self.world > 1:
            dist.all_reduce(S, op=dist.ReduceOp.SUM, group=self.group)       # int64 wrap-around sum == uint64 sum
            dist.all_reduce(nrel, op=dist.ReduceOp.SUM, group=self.group)
        total = totals[:, 1].clone()
        if remove_first:
            total = total - first_rel
        sm = ops.summarize(S, nrel, total, idx_of, Rs, ks)
        out = dict(precisions=sm["precisions"], recalls=sm["recalls"], hits=sm["hits"], total=total)
        if many:
            out.update(mAP=sm["mAPs"], S=[S[idx_of[i]] for i in range(len(Rs))], nrel=[nrel[idx_of[i]] for i in range(len(Rs))],
                       ap=sm["aps"])
        else:
            out.update(mAP=sm["mAPs"][0], S=S[idx_of[0]], nrel=nrel[idx_of[0]], ap=sm["aps"][0])
        return out
