"""Gallery-sharded retrieval across the GPUs of one node (one process per GPU, ``torch.distributed``; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" is used by the CPU tests of the host logic).

SURVEY.md section 8e: the gallery is row-sharded where it was encoded (rank r owns a contiguous block of global
rows), packed query codes are all-gathered (KB-MB, latency bound), each rank scans only its shard, and the small
per-shard results are exchanged:
  * top-k:  all_gather of the per-shard (idx, dist) lists, k-way merge by (distance, global index);
  * mAP:    all_gather of per-shard histogram TOTALS -> every rank computes the global "ranked before" bases of its own segments
            -> local AP terms (from the records of its one scan, or a second local scan) -> integer all_reduce of the AP
            numerators.  Bit-identical to the single-GPU result.

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


class RowShard:
    """This rank's block of an [N, ...] tensor whose rows are sharded over the ranks in contiguous, rank-major (= dataset) order --
    what `BaseTrainer.inference_one_epoch` returns per output in a multi-rank run: the codes STAY on the GPU that encoded them
    (SURVEY.md section 8e "shards where it was encoded, no movement"), `utils.hashing` scans them in place, and only
    `gather()` (for `save_code`) moves them.  Supports the handful of tensor operations the evaluator applies to its outputs
    (experiments/test_hashing.py:76-103): `clone`, `dim`, `size`, `shape`, column selection `x[:, cols]`, `x - row`, `mean(dim=0)`,
    and row-wise maps."""

    def __init__(self, local: torch.Tensor, counts=None, group=None):
        self.local, self.group = local, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if counts is None:
            n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
            got = [torch.zeros_like(n) for _ in range(self.world)]
            dist.all_gather(got, n, group=group)
            counts = [int(c.item()) for c in got]
        self.counts = list(counts)
        assert self.counts[self.rank] == local.shape[0]

    # ---- tensor-like surface ---------------------------------------------------------------------------------------------------
    @property
    def shape(self):
        return torch.Size((sum(self.counts),) + tuple(self.local.shape[1:]))

    @property
    def dtype(self):
        return self.local.dtype

    @property
    def device(self):
        return self.local.device

    @property
    def offset(self):
        return sum(self.counts[: self.rank])

    def size(self, d=None):
        return self.shape if d is None else self.shape[d]

    def dim(self):
        return self.local.dim()

    def __len__(self):
        return sum(self.counts)

    def map(self, fn):
        """row-wise function of the local block (must keep the row count)"""
        out = fn(self.local)
        assert out.shape[0] == self.local.shape[0]
        return RowShard(out, self.counts, self.group)

    def clone(self):
        return self.map(lambda t: t.clone())

    def __getitem__(self, idx):
        if not (isinstance(idx, tuple) and len(idx) == 2 and idx[0] == slice(None)):
            raise IndexError("RowShard supports column selection x[:, cols] only (rows live on different ranks)")
        cols = idx[1].to(self.local.device) if torch.is_tensor(idx[1]) else idx[1]
        return self.map(lambda t: t[:, cols])

    def __sub__(self, other):
        return self.map(lambda t: t - torch.as_tensor(other).to(t.device))

    def mean(self, dim=0, keepdim=False):
        """Column mean over ALL rows with the arithmetic of the single-process evaluator (torch.mean of the whole CPU tensor, the
        reference's `db_codes.mean(dim=0, keepdim=True)`): computed on rank 0 from the gathered rows and broadcast, so that
        `zero_mean_eval` gives the same bits for any rank count.  The one option that moves codes; off in the shipped configs."""
        if dim != 0:
            raise NotImplementedError("RowShard.mean: dim=0 only")
        full = self.gather(dst=0)
        box = [full.mean(dim=0, keepdim=keepdim) if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=self.group)
        return box[0]

    def gather(self, dst=0):
        """The whole tensor on the host of rank `dst` (None on the other ranks); dst=None: on every rank.  No rank other than
        `dst` copies anything to its host."""
        full, _ = _all_gather_ragged(self.local.contiguous(), self.group)
        if dst is None or self.rank == dst:
            return full.cpu()
        return None


def gather_outputs(outputs: dict, dst=0):
    """{name: RowShard | anything} -> the same dict with every RowShard gathered to rank `dst` (for `save_code`)."""
    return {k: (v.gather(dst) if isinstance(v, RowShard) else v) for k, v in outputs.items()}


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
        # CH_FORCE_COLLECTIVES=1: run every collective also in a one-rank group (they are then identities) -- lets a single GPU
        # exercise the RCCL backend with this module's dtypes and shapes (tests/test_parity_r3_gpu.py)
        self.coll = self.world > 1 or os.environ.get("CH_FORCE_COLLECTIVES") == "1"

    # ---- queries -------------------------------------------------------------------------------------------------
    def gather_queries(self, q_local: torch.Tensor) -> torch.Tensor:
        """Queries encoded on different ranks -> the full query set on every rank (rank-major order)."""
        allq, _ = _all_gather_ragged(q_local, self.group)
        return allq

    # ---- top-k ---------------------------------------------------------------------------------------------------
    def topk(self, q_all: torch.Tensor, k: int):
        """q_all: the SAME [Qn, W] on every rank.  Returns the global (idx int64 [Qn,k], dist int32 [Qn,k]) on every rank."""
        idx, dst = self.ops.hamming_topk(q_all, self.gallery, k, g_index_base=self.base)
        if not self.coll:
            return idx, dst
        li = _all_gather_rows(idx.unsqueeze(0), self.group)   # [world, Qn, k]
        ld = _all_gather_rows(dst.unsqueeze(0), self.group)
        return self.ops.topk_merge(li, ld)

    # ---- mAP / P@k / R@k -----------------------------------------------------------------------------------------
    def evaluate(self, q_all: torch.Tensor, q_labels: torch.Tensor, R=-1, ks: Sequence[int] = (1, 5, 10),
                 remove_first: bool = False, seg_rows: Optional[int] = None, skip_queries_without_relevant: bool = False) -> dict:
        """Same statistics as ``retrieval.evaluate`` (R an int or a list; `skip_queries_without_relevant` as there), gallery sharded by rows: per-shard histograms are
        all-gathered, every rank builds the same global bases, ONE local AP pass accumulates every rank limit (each R and each
        k), and the integer sums are all-reduced -- bit-identical to the single-GPU result for any shard count."""
        if self.labels is None:
            raise ValueError("gallery labels are required for evaluate()")
        ops = self.ops
        dev = q_all.device
        Qn, W = q_all.shape
        # one-hot rows -> class ids only if EVERY shard's rows (and the queries) are single-label: the ranks must agree on the form
        single = None
        ql, gl = q_labels.to(dev), self.labels.to(dev)
        if self.coll and ql.dim() == 2 and gl.dim() == 2 and hasattr(ops, "labels_single"):
            flag = torch.tensor([1 if ops.labels_single(ql, gl) else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            single = bool(flag.item())
        q_lab, g_lab, LW = ops.prepare_labels(ql, gl, single) if single is not None else ops.prepare_labels(ql, gl)
        ks = [int(k) for k in ks]
        if any(k <= 0 for k in ks):
            raise ValueError("P@k / R@k need k >= 1")
        many = isinstance(R, (list, tuple))
        Rs = [int(r) for r in R] if many else [int(R)]
        first_rel = None
        if remove_first:      # relevance of the global rank-1 row of every query
            idx, _ = self.topk(q_all, 1)
            g_lab_all, _ = _all_gather_ragged(g_lab, self.group) if self.coll else (g_lab, None)
            safe = idx.clamp_min(0)
            if LW == 0:
                rel = (g_lab_all[safe] == q_lab[:, None]) & (idx >= 0)
            else:
                rel = (g_lab_all[safe] & q_lab[:, None, :]).ne(0).any(-1) & (idx >= 0)
            first_rel = rel[:, 0].to(torch.int32)
        # one segment size for all ranks (chosen for the largest shard)
        gmax = max(self.counts) if self.counts else 0
        seg = seg_rows or ops.map_seg_rows(Qn, max(gmax, 1), W)
        # one-scan form where the ops provide it (concepthash_amd.retrieval does): the histogram pass also records this shard's
        # relevant rows, and the AP terms come from the records once the GLOBAL bases exist (CH_HAMMING_RECORDS=0 = two scans)
        use_rec = (hasattr(ops, "hamming_hist_rec") and os.environ.get("CH_HAMMING_RECORDS", "1") != "0"
                   and ops.records_fit(Qn, max(gmax, 1), W, seg))
        if use_rec and LW == 0 and hasattr(ops, "predicted_overflow") and Qn * max(gmax, 1) >= ops.OVERFLOW_MIN_PAIRS:
            # a local choice (each rank for its own shard): only speed depends on it
            use_rec = ops.predicted_overflow(q_lab, g_lab, W, seg, ops.record_cap(Qn, max(gmax, 1), W, seg)) <= ops.OVERFLOW_SWITCH
        if use_rec:
            hist, recs = ops.hamming_hist_rec(q_all, self.gallery, q_lab, g_lab, LW, seg)
        else:
            hist = ops.hamming_hist(q_all, self.gallery, q_lab, g_lab, LW, seg)        # [nseg_local, Qn, nb, 2]
        if self.coll:
            # "ranked before" bases of THIS shard's segments need, per (query, bucket), the rows of all lower buckets anywhere plus
            # the same bucket's rows in earlier shards and earlier local segments: only per-SHARD totals travel ([Qn, nb, 2] per
            # rank -- 17 MB at 16,384 queries x 128 bit -- instead of every segment's histogram, 16x that at the 1M-row size), and
            # the prefix runs over [all earlier shards | local segments | all later shards]
            tot = hist.sum(0, dtype=hist.dtype)
            tot_all = _all_gather_rows(tot.unsqueeze(0), self.group)                        # [world, Qn, nb, 2], rank-major
            before = tot_all[: self.rank].sum(0, dtype=hist.dtype)
            after = tot_all[self.rank + 1:].sum(0, dtype=hist.dtype)
            base_s, totals = ops.hist_prefix(torch.cat([before.unsqueeze(0), hist, after.unsqueeze(0)], dim=0))
            base = base_s[1:1 + hist.shape[0]].contiguous()
        else:
            base, totals = ops.hist_prefix(hist)
        limits, idx_of = ops.normalize_limits(Rs + ks)
        if use_rec:
            S, nrel = ops.hamming_ap_rec(q_all, self.gallery, q_lab, g_lab, LW, seg, base, recs, limits, first_rel=first_rel)
        else:
            S, nrel = ops.hamming_ap_multi(q_all, self.gallery, q_lab, g_lab, LW, seg, base, limits, first_rel=first_rel)
        if self.coll:
            dist.all_reduce(S, op=dist.ReduceOp.SUM, group=self.group)       # int64 wrap-around sum == uint64 sum
            dist.all_reduce(nrel, op=dist.ReduceOp.SUM, group=self.group)
        total = totals[:, 1].clone()
        if remove_first:
            total = total - first_rel
        sm = ops.summarize(S, nrel, total, idx_of, Rs, ks, skip_queries_without_relevant)
        out = dict(precisions=sm["precisions"], recalls=sm["recalls"], hits=sm["hits"], total=total)
        if many:
            out.update(mAP=sm["mAPs"], S=[S[idx_of[i]] for i in range(len(Rs))], nrel=[nrel[idx_of[i]] for i in range(len(Rs))],
                       ap=sm["aps"])
        else:
            out.update(mAP=sm["mAPs"][0], S=S[idx_of[0]], nrel=nrel[idx_of[0]], ap=sm["aps"][0])
        return out
