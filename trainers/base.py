"""`trainers.base.BaseTrainer` -- the trainer protocol the evaluator and the training loop drive (reference
trainers/base.py:17-71,128-175,195-210,275-307,340-358).

Kept: method names, argument meaning, return shapes -- `inference_one_epoch(datakey, return_codes=True) ->
(meters, {'codes': FloatTensor[N, nbit] on CPU, 'labels': Tensor[N, C]})`.
Changed on purpose: per-batch outputs stay on the GPU and are copied to the host once per epoch (the reference does a
synchronising `.cpu()` per batch, trainers/base.py:291-296), and so do the loss / accuracy meters (utils.misc.DeviceMeters: the
reference reads every term back with `.item()` per batch, trainers/coop.py:80-101).  Training: the adapters + get_training_modules() with the backbone frozen
(`backbone_lr_scale: 0`, the shipped ConceptHash config); a trainable backbone is not built.
"""
from __future__ import annotations

import logging
import os
from collections import defaultdict

import numpy as np
import torch
import yaml

from concepthash_amd.config import DictConfig, instantiate, to_container
from utils.misc import AverageMeter, DeviceMeters


class BaseTrainer:
    def __init__(self, config: DictConfig):
        self.config = config
        self.dataset = None
        self.dataloader = None
        self.model = None
        self.optimizer = None
        self.scheduler = None
        self.criterion = None
        self.current_epoch = 0
        self.inference_datakey = ""
        self.device = torch.device(config["device"])
        import torch.distributed as dist
        self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.rank = dist.get_rank() if self.distributed else 0
        self.world_size = dist.get_world_size() if self.distributed else 1
        if self.distributed and self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if self.device.type == "cuda" and torch.cuda.is_available():
            # Create the GPU context NOW, before the model is built.  Host memory that was allocated before the HIP runtime came up is
            # copied to the device on a slow path (measured on MI355X / ROCm 7: the 572 tensors of a ViT-B/16 model built first,
            # `model.to("cuda")` 2.05 s + 0.22 s of context creation; context first, 0.16 s -- tools/e2e_validation_demo.py phases), and
            # the reference's order -- build, load the checkpoint, then `.to(device)` (experiments/test_hashing.py:34-43) -- is that case.
            torch.empty(1, device=self.device)

    # ---- loading -------------------------------------------------------------------------------------------------
    def load_criterion(self):
        self.criterion = instantiate(self.config.criterion)

    def load_model(self):
        self.model = instantiate(self.config.model)

    def load_for_inference(self, logdir):
        pass

    def load_model_state(self, fn):
        sd = torch.load(fn, map_location="cpu")
        self.model.load_state_dict(sd)

    def save_model_state(self, fn):
        os.makedirs(os.path.dirname(fn) or ".", exist_ok=True)
        torch.save(self.model.state_dict(), fn)

    def save_config(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        with open(os.path.join(logdir, "config.yaml"), "w") as f:
            yaml.safe_dump(to_container(self.config), f)

    def to_device(self, device=None):
        device = self.device if device is None else device
        from utils.misc import module_to_device      # one host -> device copy per dtype, not one per tensor (0.78 s -> 0.1 s for ViT-B/16)
        if self.model is not None:
            self.model = module_to_device(self.model, device)
        if self.criterion is not None:
            self.criterion = module_to_device(self.criterion, device)

    def is_ready_for_inference(self):
        return all(x is not None for x in (self.dataset, self.dataloader, self.model, self.criterion))

    # ---- inference -----------------------------------------------------------------------------------------------
    def inference_one_batch(self, *args, **kwargs):
        raise NotImplementedError

    def iterate_loader(self, loader):
        """Hook: how a split's loader is iterated (COOPTrainer wraps `gpu_decode` loaders with a host-stage prefetch thread)."""
        return loader

    def inference_one_epoch(self, datakey="test", return_codes=False, **kwargs):
        assert self.is_ready_for_inference()
        self.model.eval()
        self.criterion.eval()
        dmeters = DeviceMeters(self.device)       # sums stay on the GPU: no `.item()` inside the batch loop
        if self.device.type == "cuda" and self.config.get("meter_stream", True):
            # loss terms, accuracies and meter sums of batch i on a side stream, beside the encode of batch i + 1: ~40 tiny launches that
            # would otherwise stand between two batches on the main stream (0.3 ms of a 4.1 ms batch at the reference's batch 64)
            if getattr(self, "_meter_stream", None) is None:
                self._meter_stream = torch.cuda.Stream(self.device)
            dmeters.stream = self._meter_stream
        ret = defaultdict(list)
        self.inference_datakey = datakey
        loader = self.dataloader[datakey]
        n = len(loader) if hasattr(loader, "__len__") else 0
        for i, data in enumerate(self.iterate_loader(loader)):
            output = self.inference_one_batch(data, dmeters, bidx=i, **kwargs)
            if return_codes:
                for key, val in output.items():
                    ret[key].append(val)          # GPU tensors stay on the GPU until the epoch ends
            if n and (i + 1) % max(1, n // 10) == 0:
                # progress line without a synchronisation: the newest COMPLETED asynchronous snapshot of the sums (one interval old)
                logging.info("%s: batch %d/%d %s", datakey, i + 1, n, dmeters.latest() or "")
                dmeters.snapshot()
        meters = defaultdict(AverageMeter, dmeters.finalize())     # the one device -> host read of the epoch's meters
        if not return_codes:
            return meters
        res = {}
        if self.distributed:
            # every rank enters the SAME sequence of collectives, also one whose shard produced no batch: the ranks first agree
            # on the output keys (name, trailing shape, dtype) -- taken from any rank that has data -- and an empty rank
            # contributes zero-row tensors of that description.
            # Tensor outputs are NOT gathered: each comes back as a `RowShard` -- this rank's rows, on its GPU, plus the per-rank row
            # counts -- so that retrieval shards the gallery where it was encoded (`utils.hashing` all-gathers only the packed QUERY
            # codes); `config.gather_outputs: true` restores the gathered CPU tensors on every rank.
            import torch.distributed as dist
            from concepthash_amd.distributed import RowShard
            mine = [(k, tuple(v[0].shape[1:]), str(v[0].dtype).replace("torch.", "")) for k, v in sorted(ret.items())
                    if isinstance(v[0], torch.Tensor)]
            other = sorted(k for k, v in ret.items() if not isinstance(v[0], torch.Tensor))
            allk = [None] * self.world_size
            dist.all_gather_object(allk, (mine, other))
            spec = next((x[0] for x in allk if x[0]), [])
            for key, tail, dt in spec:
                t = torch.cat(ret[key]) if key in ret else torch.zeros((0,) + tuple(tail), dtype=getattr(torch, dt), device=self.device)
                shard = RowShard(t.contiguous())                # ranks hold contiguous blocks -> rank-major == dataset order
                res[key] = shard.gather(dst=None) if self.config.get("gather_outputs") else shard
            for key in sorted(set().union(*[x[1] for x in allk])):      # non-tensor (numpy / list) outputs: small, replicated
                parts = [None] * self.world_size
                dist.all_gather_object(parts, np.concatenate(ret[key]) if key in ret else None)
                res[key] = np.concatenate([p for p in parts if p is not None])
        else:
            for key, vals in ret.items():
                if isinstance(vals[0], torch.Tensor):
                    res[key] = torch.cat(vals).cpu()  # one device->host copy per output per epoch
                else:
                    res[key] = np.concatenate(vals)
        if self.distributed:                      # meters: sample-weighted average over ranks
            import torch.distributed as dist
            names = [None] * self.world_size
            dist.all_gather_object(names, sorted(meters))          # a rank without batches has no meters of its own
            for k in sorted(set().union(*names)):
                v = torch.tensor([meters[k].sum, float(meters[k].count)], dtype=torch.float64, device=self.device)
                dist.all_reduce(v)
                meters[k].sum, meters[k].count = float(v[0]), int(v[1])
                meters[k].avg = meters[k].sum / max(meters[k].count, 1)
        return meters, res

    # ---- training (reference :133-175, :340-358) --------------------------------------------------------------------------
    def load_optimizer_and_scheduler(self):
        assert self.model is not None
        if self.config.get("backbone_lr_scale", 0) != 0:
            raise NotImplementedError("backbone_lr_scale != 0 (a trainable backbone) is not built on the MI355X path: the shipped "
                                      "ConceptHash config freezes it (configs/model/concept_hash_final_v1_nosa_apt.yaml)")
        groups = []
        if self.config.model.get("has_adapter", False):
            groups.append({"params": list(self.model.get_adapter().parameters())})
        groups.append({"params": [p for p in self.model.get_training_modules().parameters() if p is not None]})
        self.model.requires_grad_(False)          # only what the optimizer holds is trainable (reference :147-152)
        count = 0
        for g in groups:
            for p in g["params"]:
                p.requires_grad_(True)
                count += p.numel()
        logging.info("Number of trainable params: %.3f%s", count / (1e6 if count >= 1e6 else 1e3), "M" if count >= 1e6 else "K")
        self.optimizer = instantiate(self.config.optim, groups)
        if self.config.model.get("has_adapter", False):
            # the adapters are 168 views of one arena: update them with one launch instead of torch's per-tensor kernels
            from concepthash_amd.training import fuse_adapter_sgd
            self.optimizer = fuse_adapter_sgd(self.optimizer, self.model)
        self.scheduler = instantiate(self.config.scheduler, self.optimizer)

    def get_learning_rate(self):
        return [0.0] if self.scheduler is None else self.scheduler.get_last_lr()

    def is_ready_for_training(self):
        return all(x is not None for x in (self.dataset, self.dataloader, self.model, self.optimizer, self.scheduler, self.criterion))

    def save_training_state(self, fn):
        os.makedirs(os.path.dirname(fn) or ".", exist_ok=True)
        state = {"optim": self.optimizer.state_dict(), "scheduler": self.scheduler.state_dict()}
        eng = getattr(self.model, "_train_engine", None)
        if eng is not None and eng.momentum_buf is not None:      # the adapters' momentum lives in the fused arena step, not in torch's state
            state["adapter_momentum"] = eng.momentum_buf.cpu()
        torch.save(state, fn)

    def load_training_state(self, fn):
        sd = torch.load(fn, map_location="cpu")
        self.optimizer.load_state_dict(sd["optim"])
        self.scheduler.load_state_dict(sd["scheduler"])
        self.optimizer.restored_adapter_momentum = sd.get("adapter_momentum")   # picked up by the fused arena step on its first call

    def train_one_batch(self, *args, **kwargs):
        raise NotImplementedError

    def train_one_epoch(self, **kwargs):
        assert self.is_ready_for_training()
        self.model.train()
        self.criterion.train()
        dmeters = DeviceMeters(self.device)       # as in inference_one_epoch: nothing is read back inside the batch loop
        loader = self.dataloader["train"]
        n = len(loader) if hasattr(loader, "__len__") else 0
        for i, data in enumerate(loader):
            self.train_one_batch(data, dmeters, bidx=i, **kwargs)
            if n and (i + 1) % max(1, n // 5) == 0:
                logging.info("train: batch %d/%d %s", i + 1, n, dmeters.latest() or "")
                dmeters.snapshot()
        self.scheduler.step()
        return defaultdict(AverageMeter, dmeters.finalize())
