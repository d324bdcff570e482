"""`experiments.test_hashing.RetrievalEvaluation` -- the evaluation entry point named by the north star
(reference experiments/test_hashing.py:18-181): build the trainer from the run's config, load the checkpoint, encode
query and database sets, post-process codes, score retrieval, write `history.json` (and `outputs.pth`).

Flow and option names follow the reference (`R`, `PRs`, `use_last`, `compute_mAP`, `ternary_threshold`, `dist_metric`,
`sub_code_eval`, `sub_code_eval_setting.{start_bit,end_bit,rand_bits}`, `zero_mean_eval`, `test_as_database`,
`save_code`, `eval_logdir`); encode runs in the HIP encoder, retrieval in the HIP Hamming kernels (`utils.hashing`).
Quirk kept as is: the reference takes the *slice* [start_bit, end_bit) when `rand_bits != 0` and draws `rand_bits`
random bit positions when `rand_bits == 0` (:88-98; the branches are swapped w.r.t. the comment in configs/val.yaml:29).
The second branch therefore selects zero bits; the reference would then score empty codes -- here it raises instead.
"""
from __future__ import annotations

import json
import logging
import os
import time
from datetime import datetime

import torch
import torch.nn.functional as F
import yaml

import engine
from concepthash_amd.config import DictConfig, instantiate, to_container
from utils import io
from utils.hashing import calculate_mAP, calculate_pr_curve


def _rows(x, fn):
    """row-wise function of an output: a tensor (single process) or this rank's block of one (`RowShard`, multi-rank)"""
    return x.map(fn) if hasattr(x, "map") and not torch.is_tensor(x) else fn(x)


class RetrievalEvaluation:
    def __init__(self, config: DictConfig):
        io.init_save_queue()
        self.start_time = time.time()
        engine.seeding(config["seed"])
        logdir = config.logdir
        modelfn = "last" if config.get("use_last") else "best"
        self.timing = {}                      # seconds per phase of the command (printed at the end, kept in history.json)

        def phase(name, fn, *a, **k):
            t0 = time.perf_counter()
            out = fn(*a, **k)
            self.timing[name] = round(self.timing.get(name, 0.0) + time.perf_counter() - t0, 3)
            return out
        self._phase = phase
        trainer = instantiate(config.trainer, config)
        phase("datasets", trainer.load_dataset, load_db=True)
        phase("loaders", trainer.load_dataloader)
        if config.exp not in ("descriptor", "extract"):
            trainer.load_for_inference(logdir)
        phase("model_build", trainer.load_model)
        trainer.load_criterion()
        if config.exp not in ("descriptor", "extract"):
            phase("checkpoint", trainer.load_model_state, f"{logdir}/models/{modelfn}.pth")
        phase("to_device", trainer.to_device)
        # one process per GPU: `eval_logdir` defaults to a time-stamped directory (configs/val.yaml), which every rank would
        # compose differently -> rank 0 decides and broadcasts, and only rank 0 writes files
        import torch.distributed as dist
        self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.rank = dist.get_rank() if self.distributed else 0
        box = [str(config.eval_logdir)]
        if self.distributed:
            dist.broadcast_object_list(box, src=0)
        self.eval_logdir = box[0]
        if self.rank == 0:
            os.makedirs(self.eval_logdir, exist_ok=True)
            with open(os.path.join(self.eval_logdir, "eval_config.yaml"), "w") as f:
                yaml.safe_dump(to_container(config), f)
        self.config, self.trainer = config, trainer

    # ---- code post-processing (reference :83-103) ----------------------------------------------------------------
    def _sub_codes(self, db, test):
        st = self.config.sub_code_eval_setting
        if int(st.rand_bits) != 0:              # reference :88-92
            return db[:, st.start_bit:st.end_bit], test[:, st.start_bit:st.end_bit]
        idx = torch.randperm(db.size(1))[: int(st.rand_bits)]   # reference :93-98 -> rand_bits == 0 -> no bits
        if idx.numel() == 0:
            raise ValueError("sub_code_eval with rand_bits == 0 selects zero bits (reference quirk, see module "
                             "docstring); set sub_code_eval_setting.rand_bits != 0 to evaluate [start_bit, end_bit)")
        return db[:, idx], test[:, idx]

    def main(self):
        cfg = self.config
        print("Testing Start")
        res = {}
        test_meters, test_out = self._phase("encode_test", self.trainer.inference_one_epoch, "test", True)
        db_meters, db_out = self._phase("encode_db", self.trainer.inference_one_epoch, "db", True)
        for k, m in test_meters.items():
            res["test_" + k] = m.avg
        for k, m in db_meters.items():
            res["db_" + k] = m.avg
        names = [k for k in test_out if "codes" in k]            # every "*codes*" output is evaluated (:68-73)
        if cfg.exp != "extract":
            for name in names:
                postfix = "_".join(name.split("_")[1:])
                print(f'Evaluating for "{name}"')
                db_labels, test_labels = db_out["labels"].clone(), test_out["labels"].clone()
                if db_labels.dim() == 1:
                    db_labels = _rows(db_labels, lambda t: F.one_hot(t, cfg.dataset.nclass))
                    test_labels = _rows(test_labels, lambda t: F.one_hot(t, cfg.dataset.nclass))
                db_codes, test_codes = db_out[name], test_out[name]
                if cfg.get("compute_mAP") and cfg.get("sub_code_eval"):      # reference: only inside `if compute_mAP` (:76-98)
                    db_codes, test_codes = self._sub_codes(db_codes, test_codes)
                if cfg.get("compute_mAP") and cfg.get("zero_mean_eval"):
                    mean = db_codes.mean(dim=0, keepdim=True)        # database mean, applied to both sets (:100-103)
                    db_codes, test_codes = db_codes - mean, test_codes - mean
                as_db = bool(cfg.get("test_as_database"))
                g_codes, g_labels = (test_codes, test_labels) if as_db else (db_codes, db_labels)
                if cfg.get("compute_mAP"):
                    mAPs, recalls, precisions = self._phase("retrieval", calculate_mAP, g_codes, g_labels, test_codes, test_labels, cfg.R,
                                                            threshold=cfg.ternary_threshold, dist_metric=cfg.dist_metric,
                                                            PRs=cfg.PRs, remove_first_retrieved=as_db)
                    res["mAP" + postfix], res["recalls" + postfix], res["precisions" + postfix] = mAPs, recalls, precisions
                    if isinstance(mAPs, list):
                        for R, m in zip(cfg.R, mAPs):
                            print(f"mAP@{R}: {m:.4f}")
                    else:
                        print(f"mAP@{cfg.R}: {mAPs:.4f}")
                    for k, r, p in zip(cfg.PRs, recalls, precisions):
                        print(f"P@{k}: {p:.4f}; R@{k}: {r:.4f}")
                else:
                    recalls, precisions, Rs = calculate_pr_curve(g_codes, g_labels, test_codes, test_labels,
                                                                 threshold=cfg.ternary_threshold,
                                                                 dist_metric=cfg.dist_metric, remove_first_retrieved=as_db)
                    res["recalls" + postfix], res["precisions" + postfix], res["Rs" + postfix] = recalls, precisions, Rs
                    for R, r, p in zip(Rs, recalls, precisions):
                        print(f"P@{R}: {p:.4f}; R@{R}: {r:.4f}")
                print()
            res["timing_s"] = dict(self.timing, since_start=round(time.time() - self.start_time, 3), written_unix=round(time.time(), 3))
            if self.rank == 0:
                with open(os.path.join(self.eval_logdir, "history.json"), "w") as f:
                    json.dump(res, f)
        if cfg.get("save_code") or cfg.exp == "extract":
            if self.distributed:       # the one place the fp32 codes move: gathered to rank 0's host, which writes the file
                from concepthash_amd.distributed import gather_outputs
                test_out, db_out = gather_outputs(test_out, 0), gather_outputs(db_out, 0)
            if self.rank == 0:
                print("Saving code")
                io.fast_save({"test": test_out, "db": db_out}, os.path.join(self.eval_logdir, "outputs.pth"))
        total = time.time() - self.start_time
        print(f'Testing End at {datetime.today().strftime("%Y-%m-%d %H:%M:%S")}')
        print(f"Total time used: {total / 3600:.2f} hours")
        print("Phases (s): " + ", ".join(f"{k} {v:.2f}" for k, v in self.timing.items()))
        io.join_save_queue()
        print(f"Done: {self.eval_logdir}")
        self.results = res
        return res
