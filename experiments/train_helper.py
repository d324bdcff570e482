"""`experiments.train_helper.RetrievalExperiment` -- the `exp=hashing` training loop (reference experiments/train_helper.py:47-304):
build the trainer, datasets, model, optimizer + scheduler and criterion from the composed config; per epoch
`trainer.train_one_epoch`, every `eval_interval` epochs encode query + database and score retrieval, keep `models/best.pth` /
`models/last.pth`, `train_history.json` / `test_history.json`.

What trains: the adapters + the hashing head + the concept-token generator with the CLIP backbone frozen -- the shipped
ConceptHash setting.  The encoder's forward and backward of every step run in the HIP library (ch_train_forward /
ch_train_backward); evaluation runs the HIP encode + Hamming kernels as `experiments.test_hashing` does.
Not carried over: wandb logging, resume (`resume_logdir`), fine-tuning (`finetune_path`) and the gldv2 landmark protocol.
"""
from __future__ import annotations

import json
import logging
import os
import time
from datetime import datetime

import yaml

import engine
from concepthash_amd.config import DictConfig, instantiate, to_container
from utils import io
from utils.hashing import calculate_mAP


class RetrievalExperiment:
    def __init__(self, config: DictConfig):
        io.init_save_queue()
        self.start_time = time.time()
        engine.seeding(config["seed"])
        for key in ("resume_logdir", "finetune_path"):
            if config.get(key) is not None:
                raise NotImplementedError(f"{key} is not built on the MI355X path")
        self.config, self.logdir = config, str(config.logdir)
        for sub in ("models", "optims", "outputs"):
            os.makedirs(os.path.join(self.logdir, sub), exist_ok=True)
        trainer = instantiate(config.trainer, config)
        trainer.save_config(self.logdir)
        trainer.load_dataset()
        trainer.load_dataloader()
        trainer.load_model()
        trainer.load_optimizer_and_scheduler()
        trainer.load_criterion()
        trainer.to_device()
        self.trainer = trainer
        self.train_history, self.test_history = [], []
        self.best, self.best_ep, self.curr_metric = 0.0, 0, 0.0
        self.nepochs, self.neval, self.nsave = int(config.epochs), int(config.get("eval_interval", 10)), int(config.get("save_interval", 0))
        logging.info("Training Start")

    def record_history(self, stage, stats):
        hist = self.train_history if stage == "train" else self.test_history
        hist.append(stats)
        with open(os.path.join(self.logdir, f"{stage}_history.json"), "w") as f:
            json.dump(hist, f, indent=True)

    def evaluation(self, ep):
        """reference :187-251: encode both splits, score every `codes*` output."""
        tr, config = self.trainer, self.config
        res = {"ep": ep + 1}
        test_meters, test_out = tr.inference_one_epoch("test", True, ep=ep)
        db_meters, db_out = tr.inference_one_epoch("db", True, ep=ep)
        for k, v in test_meters.items():
            res["test_" + k] = v.avg
        for k, v in db_meters.items():
            res["db_" + k] = v.avg
        names = [k for k in test_out if "codes" in k]
        assert names
        for name in names:
            postfix = "_".join(name.split("_")[1:])
            q, g = test_out[name], db_out[name]
            if config.get("zero_mean_eval"):
                mean = g.mean(dim=0, keepdim=True)
                q, g = q - mean, g - mean
            mAP, recalls, precisions = calculate_mAP(g, db_out["labels"], q, test_out["labels"], config.dataset.R,
                                                     dist_metric=config.dist_metric, PRs=[1, 5, 10],
                                                     multiclass=config.dataset.get("multiclass", False))
            res["mAP" + postfix], res["recalls" + postfix], res["precisions" + postfix] = mAP, recalls, precisions
            logging.info("mAP%s: %.6f  R@10 %.6f  P@10 %.6f", postfix, mAP, recalls[-1], precisions[-1])
        return res, test_out, db_out

    def main(self):
        tr = self.trainer
        for ep in range(self.nepochs):
            res = {"ep": ep + 1}
            lrs = tr.get_learning_rate()
            for i, lr in enumerate(lrs):
                res[f"lr/{i}"] = lr
            logging.info("Epoch [%d/%d]; LR: %s", ep + 1, self.nepochs, "; ".join(f"{lr:.6f}" for lr in lrs))
            tr.current_epoch = ep
            t0 = time.time()
            meters = tr.train_one_epoch(ep=ep)
            for k, v in meters.items():
                res["train_" + k] = v.avg
            res["train_seconds"] = time.time() - t0
            self.record_history("train", res)
            if (ep + 1) == self.nepochs or (self.neval != 0 and (ep + 1) % self.neval == 0):
                res, test_out, db_out = self.evaluation(ep)
                self.curr_metric = res["mAP"]
                self.record_history("test", res)
                if self.best < self.curr_metric:
                    self.best, self.best_ep = self.curr_metric, ep + 1
                    tr.save_model_state(f"{self.logdir}/models/best.pth")
            if self.nsave != 0 and (ep + 1) % self.nsave == 0:
                tr.save_model_state(f"{self.logdir}/models/ep{ep + 1}.pth")
            tr.save_model_state(f"{self.logdir}/models/last.pth")
            if self.config.get("save_training_state"):
                tr.save_training_state(f"{self.logdir}/optims/last.pth")
        io.join_save_queue()
        logging.info("Training End at %s; total %.2f hours; best mAP %.6f at %d; Done: %s", datetime.today().strftime("%Y-%m-%d %H:%M:%S"),
                     (time.time() - self.start_time) / 3600, self.best, self.best_ep, self.logdir)
        return {"best": self.best, "best_ep": self.best_ep, "train_history": self.train_history, "test_history": self.test_history}
