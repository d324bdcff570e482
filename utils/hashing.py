"""`utils.hashing` -- retrieval metrics under the reference's import path, computed by the HIP kernels.

The reference imports `calculate_mAP`, `calculate_pr_curve` (experiments/test_hashing.py:15) and `get_hamm_dist`
(trainers/orthohash.py:17) from an un-vendored package whose source is not in the snapshot; signatures and return
arity below are those of the call sites (experiments/test_hashing.py:106-131,153-167; experiments/train_helper.py:228-241;
trainers/orthohash.py:362).  Semantics are the published definition of SURVEY.md section 8c / DESIGN.md section 2.

Inputs may be CPU or GPU float tensors (the reference hands over CPU tensors, trainers/base.py:291-296); they are moved
to the current GPU, sign-packed to uint64 and never expanded to a (Qn, G) matrix.  In a multi-rank run the trainer hands over
`concepthash_amd.distributed.RowShard`s -- each rank's rows, still on the GPU that encoded them: the database is scanned in place,
only the packed query codes are all-gathered.
"""
from __future__ import annotations

from typing import List, Sequence

import torch

from concepthash_amd import retrieval as rt


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("utils.hashing needs a GPU (MI355X); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _check(dist_metric, threshold, landmark_gt):
    if dist_metric != "hamming":
        raise NotImplementedError(f"dist_metric='{dist_metric}': only 'hamming' is built (the shipped configs use it)")
    if threshold not in (0, 0.0):
        raise NotImplementedError("ternary_threshold != 0 (ternary codes) is not built; every reference call site uses 0")
    if landmark_gt is not None:
        raise NotImplementedError("landmark ground-truth evaluation is outside the ConceptHash path")


def _labels(t: torch.Tensor, dev):
    t = torch.as_tensor(t)
    return t.to(dev)


def _sharded():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _evaluate(db_codes, db_labels, test_codes, test_labels, R, ks, remove_first, skip=False):
    dev = _device()
    from concepthash_amd.distributed import RowShard
    if isinstance(db_codes, RowShard) or isinstance(test_codes, RowShard):
        # multi-rank evaluation on the outputs of BaseTrainer.inference_one_epoch: every rank's block of the DATABASE stays on the
        # GPU that encoded it and is scanned there; the QUERIES' packed codes (8-16 bytes each) and compact labels are all-gathered
        from concepthash_amd.distributed import ShardedRetrieval, _all_gather_ragged
        if not (isinstance(db_codes, RowShard) and isinstance(db_labels, RowShard)):
            raise TypeError("sharded evaluation: database codes and labels must both be RowShards")
        g = rt.pack_sign(db_codes.local.to(dev, torch.float32))
        sr = ShardedRetrieval(g, db_labels.local.to(dev))
        if isinstance(test_codes, RowShard):
            q = sr.gather_queries(rt.pack_sign(test_codes.local.to(dev, torch.float32)))
        else:
            q = rt.pack_sign(torch.as_tensor(test_codes).to(dev, torch.float32))
        if isinstance(test_labels, RowShard):
            tl = test_labels.local.to(dev)
            if tl.dim() == 2:
                tl = (tl != 0).to(torch.uint8)                      # indicator rows travel as bytes
            ql, _ = _all_gather_ragged(tl.contiguous(), sr.group)
        else:
            ql = _labels(test_labels, dev)
        return sr.evaluate(q, ql, R=R, ks=ks, remove_first=remove_first, skip_queries_without_relevant=skip)
    q = rt.pack_sign(torch.as_tensor(test_codes).to(dev, torch.float32))
    if _sharded():
        # replicated inputs under an initialised process group (a caller that gathered its codes): every rank takes its block
        import torch.distributed as dist
        from concepthash_amd.distributed import ShardedRetrieval, shard_bounds
        b = shard_bounds(db_codes.shape[0], dist.get_world_size())
        lo, hi = b[dist.get_rank()], b[dist.get_rank() + 1]
        g = rt.pack_sign(torch.as_tensor(db_codes[lo:hi]).to(dev, torch.float32))
        sr = ShardedRetrieval(g, _labels(db_labels[lo:hi], dev))
        return sr.evaluate(q, _labels(test_labels, dev), R=R, ks=ks, remove_first=remove_first, skip_queries_without_relevant=skip)
    g = rt.pack_sign(torch.as_tensor(db_codes).to(dev, torch.float32))
    return rt.evaluate(q, g, _labels(test_labels, dev), _labels(db_labels, dev), R=R, ks=ks, remove_first=remove_first,
                       skip_queries_without_relevant=skip)


def calculate_mAP(db_codes, db_labels, test_codes, test_labels, R, threshold=0., dist_metric="hamming", PRs=None,
                  remove_first_retrieved=False, landmark_gt=None, db_id=None, test_id=None, multiclass=False,
                  skip_queries_without_relevant=False):
    """-> (mAP, recalls, precisions); `mAP` is a list when `R` is a list (experiments/test_hashing.py:124-128).
    R <= 0 means the whole database.  One histogram pass + one AP pass whatever the length of R and PRs: every R and every k
    is a rank limit of the same gallery scan.
    skip_queries_without_relevant (not a reference argument; the reference's un-vendored `utils.hashing` cannot be read, so the one
    convention that changes numbers at R < database size is a switch): False (default) = a query with no relevant row inside its top
    R contributes AP = 0 to the mean; True = such queries are left out of the mean, as the HashNet /
    OrthoHash family of evaluators does.  mAP@all on CUB-200 / Cars196 / NABirds is the same under both."""
    _check(dist_metric, threshold, landmark_gt)
    ks = [int(k) for k in (PRs or [])]
    many = isinstance(R, (list, tuple)) or (hasattr(R, "__iter__") and not isinstance(R, (int, float)))
    Rs = [int(r) for r in R] if many else int(R)
    if many and not Rs:
        return [], [], []
    res = _evaluate(db_codes, db_labels, test_codes, test_labels, Rs, ks, remove_first_retrieved, bool(skip_queries_without_relevant))
    return res["mAP"], res["recalls"], res["precisions"]


def pr_curve_points(n_db: int) -> List[int]:
    """Retrieval depths of the P/R sweep: 1, 2, 5 decades up to the database size, plus the size itself."""
    pts, base = [], 1
    while base <= n_db:
        for m in (1, 2, 5):
            if base * m <= n_db:
                pts.append(base * m)
        base *= 10
    if not pts or pts[-1] != n_db:
        pts.append(n_db)
    return pts


def calculate_pr_curve(db_codes, db_labels, test_codes, test_labels, threshold=0., dist_metric="hamming",
                       remove_first_retrieved=False, Rs: Sequence[int] = None):
    """-> (recalls, precisions, Rs): mean precision / recall at each retrieval depth R (experiments/test_hashing.py:153-167).
    Every depth is a rank limit of ONE AP pass (16 limits per gallery scan): #relevant-in-top-R is that limit's `nrel`."""
    _check(dist_metric, threshold, None)
    n = db_codes.shape[0] - (1 if remove_first_retrieved else 0)
    Rs = [int(r) for r in (Rs if Rs is not None else pr_curve_points(max(n, 1)))]
    if not Rs:
        return [], [], []
    res = _evaluate(db_codes, db_labels, test_codes, test_labels, Rs, [], remove_first_retrieved)
    total = res["total"].double()
    recalls, precisions = [], []
    for r, nrel in zip(Rs, res["nrel"]):
        nrel = nrel.double()
        precisions.append(float((nrel / r).mean().item()) if nrel.numel() else 0.0)
        recalls.append(float(torch.where(total > 0, nrel / total.clamp_min(1), torch.zeros_like(nrel)).mean().item())
                       if nrel.numel() else 0.0)
    return recalls, precisions, Rs


def get_hamm_dist(codes, centroids, normalize=True):
    """(B, nbit) codes vs (C, nbit) codebook -> (B, C) Hamming distance, divided by nbit when `normalize`
    (trainers/orthohash.py:362; in-repo twin get_hd, trainers/orthohash.py:263-264).  Output stays on the GPU."""
    dev = _device()
    a = rt.pack_sign(torch.as_tensor(codes).to(dev, torch.float32))
    b = rt.pack_sign(torch.as_tensor(centroids).to(dev, torch.float32))
    d = rt.hamming_dist(a, b).to(torch.float32)
    return d / codes.shape[1] if normalize else d


def get_sim(label_a, label_b, onehot=True):
    """Ground-truth similarity matrix (labels share >= 1 class) -- small-problem helper used by some reference losses."""
    if onehot:
        return (label_a.float() @ label_b.float().t() > 0).float()
    return (label_a.reshape(-1, 1) == label_b.reshape(1, -1)).float()
