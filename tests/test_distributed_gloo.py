"""CPU, world_size 2 (gloo): the multi-GPU retrieval choreography of concepthash_amd/distributed.py -- ragged gallery
shards, all_gather of queries / lists / histograms, global prefix bases, integer all_reduce -- must reproduce the
single-process oracle bit for bit.  Per-shard arithmetic is the numpy/C stand-in of tests/cpu_ops.py; on GPUs the same
class runs with the HIP kernels (tests/test_hamming_gpu.py covers those, incl. the shard+merge data path on one GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, bounds, remove_first, R, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cpu_ops
        from concepthash_amd.distributed import ShardedRetrieval
        from oracle import hamming_oracle as ho
        q, ql = ho.synthetic_codes(61, 64, seed=5, nclass=6)
        g, gl = ho.synthetic_codes(bounds[-1], 64, seed=6, nclass=6)
        lo, hi = bounds[rank], bounds[rank + 1]
        tt = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64) if a.dtype == np.uint64 else a)
        sr = ShardedRetrieval(tt(g[lo:hi]), tt(gl[lo:hi]), ops=cpu_ops)
        assert sr.base == lo and sr.total == bounds[-1]
        # queries "encoded" on different ranks, unequal counts
        qb = [0, 20, 61] if world == 2 else [0, 61]
        q_all = sr.gather_queries(tt(q[qb[rank]:qb[rank + 1]]))
        assert torch.equal(q_all, tt(q))
        idx, dst = sr.topk(q_all, 12)
        ev = sr.evaluate(q_all, tt(ql), R=R, ks=(1, 5, 10), remove_first=remove_first, seg_rows=97)
        if isinstance(R, list):          # a list of R: one AP pass, per-R results -- keep the layout [nR, Qn]
            ev["S"], ev["nrel"] = torch.stack(ev["S"]), torch.stack(ev["nrel"])
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), idx=idx.numpy(), dst=dst.numpy(), S=ev["S"].numpy(),
                 nrel=ev["nrel"].numpy(), hits=ev["hits"].numpy(), total=ev["total"].numpy(), mAP=np.array(ev["mAP"]),
                 P=np.array(ev["precisions"]), Rc=np.array(ev["recalls"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bounds,remove_first,R", [([0, 300, 700], False, -1), ([0, 1, 450], True, 40), ([0, 0, 333], False, -1),
                                                   ([0, 250, 600], True, [5, 100, -1])])
def test_two_rank_sharded_retrieval_matches_oracle(tmp_path, bounds, remove_first, R):
    from oracle import hamming_oracle as ho
    port = _free_port()
    mp.spawn(_worker, args=(2, port, bounds, remove_first, R, str(tmp_path)), nprocs=2, join=True)
    q, ql = ho.synthetic_codes(61, 64, seed=5, nclass=6)
    g, gl = ho.synthetic_codes(bounds[-1], 64, seed=6, nclass=6)
    ridx, rdst = ho.topk(q, g, 12)
    Rs = R if isinstance(R, list) else [R]
    refs = [ho.mean_ap(q, g, ql, gl, R=r_, ks=(1, 5, 10), remove_first=remove_first) for r_ in Rs]
    ref = refs[-1]
    for r in range(2):
        z = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(z["idx"], ridx.astype(np.int64)) and np.array_equal(z["dst"], rdst)
        if isinstance(R, list):
            for i, rf in enumerate(refs):
                assert np.array_equal(z["S"][i].view(np.uint64), rf["S"]) and np.array_equal(z["nrel"][i].astype(np.uint32), rf["nrel"])
                assert abs(float(z["mAP"][i]) - rf["mAP"]) < 1e-12
            z = dict(z, S=z["S"][-1], nrel=z["nrel"][-1], mAP=z["mAP"][-1])
        assert np.array_equal(z["S"].view(np.uint64), ref["S"])
        assert np.array_equal(z["nrel"].astype(np.uint32), ref["nrel"])
        assert np.array_equal(z["hits"].astype(np.uint32), ref["hits"])
        assert np.array_equal(z["total"].astype(np.uint32), ref["total"])
        assert abs(float(z["mAP"]) - ref["mAP"]) < 1e-12
        assert np.allclose(z["P"], ref["precisions"]) and np.allclose(z["Rc"], ref["recalls"])


def test_shard_bounds():
    from concepthash_amd.distributed import shard_bounds
    assert shard_bounds(10, 3) == [0, 4, 7, 10]
    assert shard_bounds(2, 4) == [0, 1, 2, 2, 2]
    assert shard_bounds(0, 2) == [0, 0, 0]


# ---- the trainer's rank-sharded encode loop + gather (host logic, CPU stand-in model) --------------------------------------
def _trainer_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from concepthash_amd.config import DictConfig
        from trainers.coop import COOPTrainer
        from utils.datasets import SyntheticHashingDataset

        class FakeModel(torch.nn.Module):          # codes = a deterministic function of the image, so order is checkable
            def forward(self, x):
                codes = x.flatten(1)[:, :8].float()
                z = torch.zeros(x.shape[0], 5)
                return None, {"codes": codes, "logits_cont": z, "logits_bin": z}

        class FakeCriterion(torch.nn.Module):
            losses = {}

            def forward(self, out, y):
                return out["codes"].sum() * 0

        cfg = DictConfig(device="cpu", batch_size=7, model=DictConfig(), dataset=DictConfig(multiclass=False))
        tr = COOPTrainer(cfg)
        tr.dataset = {"train": [], "test": SyntheticHashingDataset(5, size=23, image_size=4, seed=1),
                      "db": SyntheticHashingDataset(5, size=40, image_size=4, seed=2)}
        tr.load_dataloader()
        tr.model, tr.criterion = FakeModel(), FakeCriterion()
        meters, out = tr.inference_one_epoch("db", True)
        # outputs come back as RowShards: this rank's rows only (nothing gathered, nothing on another rank's host)
        from concepthash_amd.distributed import RowShard, shard_bounds
        b = shard_bounds(40, world)
        assert isinstance(out["codes"], RowShard) and out["codes"].local.shape[0] == b[rank + 1] - b[rank]
        assert out["codes"].shape == (40, 8) and out["codes"].offset == b[rank] and out["codes"].counts == [20, 20]
        sub = out["codes"][:, 2:5]                                           # what sub_code_eval does
        assert sub.shape == (40, 3) and torch.equal(sub.local, out["codes"].local[:, 2:5])
        mean = out["codes"].mean(dim=0, keepdim=True)                        # what zero_mean_eval does
        full = out["codes"].gather(dst=None)
        assert torch.equal(mean, full.mean(dim=0, keepdim=True))
        assert torch.equal((out["codes"] - mean).gather(dst=None), full - mean)
        only0 = out["labels"].gather(dst=0)
        assert (only0 is not None) == (rank == 0)
        torch.save({"codes": full, "labels": out["labels"].gather(dst=None), "n": meters["loss"].count}, os.path.join(out_dir, f"t{rank}.pt"))
        tr.config["gather_outputs"] = True                                   # the round-2 behaviour on request: gathered CPU tensors
        _, outg = tr.inference_one_epoch("db", True)
        assert torch.is_tensor(outg["codes"]) and torch.equal(outg["codes"], full)
        tr.config["gather_outputs"] = False
        # a split with fewer samples than ranks: rank 1's shard is EMPTY -- it must still enter every collective (no hang) and
        # get the gathered outputs
        tr.dataset["test"] = SyntheticHashingDataset(5, size=1, image_size=4, seed=3)
        tr.load_dataloader()
        m1, o1 = tr.inference_one_epoch("test", True)
        assert o1["codes"].counts == [1, 0]
        torch.save({"codes": o1["codes"].gather(dst=None), "labels": o1["labels"].gather(dst=None), "n": m1["loss"].count},
                   os.path.join(out_dir, f"e{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_trainer_shards_encode_and_gathers_in_dataset_order(tmp_path):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from utils.datasets import SyntheticHashingDataset
    mp.spawn(_trainer_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    ds = SyntheticHashingDataset(5, size=40, image_size=4, seed=2)
    want = torch.stack([ds[i][0].flatten()[:8] for i in range(40)])
    labels = torch.stack([ds[i][1] for i in range(40)])
    for r in range(2):
        got = torch.load(tmp_path / f"t{r}.pt")
        assert torch.equal(got["codes"], want) and torch.equal(got["labels"], labels) and got["n"] == 40
    one = SyntheticHashingDataset(5, size=1, image_size=4, seed=3)
    for r in range(2):          # the one-sample split: rank 1 had no batch, both ranks hold the same gathered result
        got = torch.load(tmp_path / f"e{r}.pt")
        assert got["codes"].shape == (1, 8) and torch.equal(got["codes"][0], one[0][0].flatten()[:8]) and got["n"] == 1
        assert torch.equal(got["labels"][0], one[0][1])
