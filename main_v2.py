#!/usr/bin/env python3
"""CLI entry point with the reference's surface (reference main_v2.py:14-64):

    python main_v2.py --config-name val.yaml logdir=<run dir> dataset=cub200 [R=-1] [PRs=[1,5,10]] [batch_size=64] ...
    python main_v2.py exp=extract model=concept_hash_final_v1_nosa_apt dataset=synthetic_cub200 ...

`exp` dispatch: hashing -> RetrievalExperiment (train the adapters + hashing head, frozen backbone); validation -> load
<logdir>/config.yaml, overlay the evaluation knobs, RetrievalEvaluation; descriptor / extract -> RetrievalEvaluation on the
composed config; general (the reference's train-without-eval variant) is not built.  Uses Hydra when it is installed; otherwise `concepthash_amd.config` composes the same YAML
tree (hydra-core / omegaconf are absent from the target image).
"""
from __future__ import annotations

import argparse
import logging
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch

from concepthash_amd import config as cfglib

EVAL_KEYS = ("dataset", "data_dir", "work_dir", "eval_logdir", "R", "PRs", "use_last", "compute_mAP", "ternary_threshold",
             "dist_metric", "batch_size", "save_code", "sub_code_eval", "sub_code_eval_setting", "zero_mean_eval",
             "test_as_database")
# knobs of this implementation's evaluation loop (configs/val.yaml), overlaid the same way
LOOP_KEYS = ("eval_batch_min", "meter_stream")


def run(config):
    from experiments.test_hashing import RetrievalEvaluation
    if config.exp == "general":
        raise NotImplementedError("exp='general' (train without evaluation, reference experiments/train_no_eval.py) is not built; "
                                  "use exp=hashing with eval_interval=0")
    if config.exp == "hashing":                  # reference main_v2.py:17-19
        from experiments.train_helper import RetrievalExperiment
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise NotImplementedError("exp=hashing is single-process (the reference trains on one GPU); launch one process")
        return RetrievalExperiment(config).main()
    if config.exp == "validation":
        load_config = cfglib.load(os.path.join(config.logdir, "config.yaml"))
        for k in EVAL_KEYS:                      # reference main_v2.py:23-40
            load_config[k] = config[k]
        for k in LOOP_KEYS:
            if k in config:
                load_config[k] = config[k]
        load_config["logdir"] = config.logdir if os.path.isabs(str(config.logdir)) else f"{config.work_dir}/{config.logdir}"
        load_config["wandb"] = False
        load_config["exp"] = "validation"
        load_config["seed"] = config.get("seed", load_config.get("seed", 42))
        load_config["device"] = config.get("device", "cuda")
        experiment = RetrievalEvaluation(load_config)
    elif config.exp in ("descriptor", "extract"):
        experiment = RetrievalEvaluation(config)
    else:
        raise ValueError(f'Unknown exp value: "{config.exp}"')
    return experiment.main()


def main(argv=None):
    ap = argparse.ArgumentParser(add_help=True)
    ap.add_argument("--config-name", "-cn", default="train.yaml")
    ap.add_argument("--config-path", "-cp", default=os.path.join(ROOT, "configs"))
    ap.add_argument("overrides", nargs="*")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    torch.multiprocessing.set_sharing_strategy("file_system")
    config = cfglib.compose(args.config_path, args.config_name, args.overrides, cwd=os.getcwd())
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:                                 # one process per GPU: gallery-sharded retrieval
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        # nccl = RCCL over xGMI; CH_DIST_BACKEND=gloo only for rehearsing the multi-rank path with several ranks on ONE GPU
        backend = os.environ.get("CH_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    try:
        return run(config)
    finally:
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
