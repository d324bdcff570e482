"""`trainers.coop.COOPTrainer` -- the ConceptHash trainer's inference side (reference trainers/coop.py:13-106)."""
from __future__ import annotations

import logging
import os

import torch

import engine
from concepthash_amd.config import DictConfig, instantiate
from trainers.base import BaseTrainer


class COOPTrainer(BaseTrainer):
    def __init__(self, config: DictConfig):
        super().__init__(config)
        os.environ["TOKENIZERS_PARALLELISM"] = "false"

    def load_criterion(self):
        super().load_criterion()
        self._sync_attention_tap()

    def load_model(self):
        super().load_model()
        self._sync_attention_tap()

    def _sync_attention_tap(self):
        """The attention-diversity term of the loss reads the concept tokens' last-layer attention rows: ask the model for them
        exactly when that term is on (reference models/loss/coop.py:161-187 reads outputs['attn_cache'])."""
        if self.model is not None and self.criterion is not None and hasattr(self.model, "return_concept_attention"):
            if getattr(self.criterion, "loss_scales", {}).get("attn_div_loss", 0):
                self.model.return_concept_attention = "all" if getattr(self.criterion, "avg_attn", False) else True

    def load_dataset(self, load_db=True):
        ds = self.config.dataset
        self.dataset = {"train": instantiate(ds.train_dataset) if ds.get("train_dataset") is not None else [],
                        "test": instantiate(ds.test_dataset),
                        "db": instantiate(ds.db_dataset) if load_db else []}
        logging.info("Number of Query data: %d; Database data: %d", len(self.dataset["test"]), len(self.dataset["db"]))

    def load_dataloader(self):
        """One process per GPU: every rank encodes a contiguous block of each split (SURVEY.md section 8e -- encode shards by
        image, no collective in the loop); `inference_one_epoch` all-gathers the per-rank outputs afterwards."""
        assert self.dataset is not None
        bs = self.config.batch_size
        self.dataloader = {}
        for k in ("test", "db"):
            ds = self.dataset[k]
            sampler = None
            if self.world_size > 1 and len(ds) > 0:
                from concepthash_amd.distributed import shard_bounds
                b = shard_bounds(len(ds), self.world_size)
                sampler = engine.get_sequential_sampler(list(range(b[self.rank], b[self.rank + 1])))
            # Evaluation splits on the GPU decode path are read `eval_batch_min` files at a time when the configured batch is smaller
            # (configs/val.yaml:10 says 64): unshuffled, nothing dropped, every image encoded on its own -- the same codes in the same order,
            # so the same mAP bit for bit -- but the encoder runs at 20k instead of 14k images/s.  Loss / accuracy
            # meters are sample-weighted means either way (equal up to fp32 rounding).  0 / 1 = the configured batch size as it is.
            ebs = bs
            if getattr(ds, "gpu_decode", False) and not getattr(ds, "file_workers", 0):
                ebs = max(bs, int(self.config.get("eval_batch_min", 256) or 0))
            self.dataloader[k] = engine.dataloader(ds, ebs, shuffle=False, drop_last=False, sampler=sampler)
        train = self.dataset.get("train") or []
        self.dataloader["train"] = engine.dataloader(train, bs, shuffle=True, drop_last=True) if len(train) else []

    def parse_model_output(self, output):
        codes, logits = output
        return logits if isinstance(logits, dict) else {"codes": codes, "logits": logits}

    def _gpu_preprocess(self, raw):
        """`dataset.gpu_preprocess: true`: decoded uint8 images of a batch -> the encoder's bf16 NCHW input, on the GPU, with the
        geometry and normalisation constants of the dataset config (`resize`, `crop`, `norm`; reference
        configs/dataset/cub200.yaml:31-47)."""
        if getattr(self, "_gpu_pre", None) is None:
            from concepthash_amd.preprocess import GpuPreprocess
            from utils.transforms import _NORMS
            ds = self.config.dataset
            mean, std = _NORMS[int(ds.get("norm", 3))]
            self._gpu_pre = GpuPreprocess(int(ds.get("resize", 256)), int(ds.get("crop", 224)), mean, std, device=self.device)
        # a training dataset's batches carry the crop boxes / flips its workers drew (RandomResizedCrop -> RandomHorizontalFlip,
        # configs/dataset/cub200.yaml:13-23): the same kernels resize the box instead of Resize -> CenterCrop
        return self._gpu_pre(raw.pixels, raw.sizes, boxes=getattr(raw, "boxes", None), flips=getattr(raw, "flips", None))

    def _jpeg_decoder(self):
        if getattr(self, "_gpu_jpeg", None) is None:
            from concepthash_amd.jpeg import GpuJpegDecoder
            self._gpu_jpeg = GpuJpegDecoder(device=self.device, threads=self.config.dataset.get("decode_threads", None) or None)
        return self._gpu_jpeg

    def _gpu_decode(self, raw):
        """`dataset.gpu_decode: true`: the batch's JPEG FILES -> decoded RGB bytes on the GPU (host threads entropy-decode, the GPU does
        inverse DCT / upsampling / colour conversion; bit-equal to the PIL decode of the reference's workers, engine.py:41-54).  `raw` is
        a RawJpegBatch (undecoded files) or, from `iterate_loader`, a StagedJpegBatch whose host half already ran on the prefetch thread."""
        from utils.datasets import RawImageBatch
        pixels, sizes = raw.finish() if hasattr(raw, "staged") else self._jpeg_decoder().decode(raw)
        return RawImageBatch(pixels, sizes, getattr(raw, "boxes", None), getattr(raw, "flips", None))

    def iterate_loader(self, loader):
        """A `gpu_decode` loader is iterated with the host half of the JPEG decode one or two batches ahead, on a background thread."""
        if getattr(getattr(loader, "dataset", None), "gpu_decode", False):
            from concepthash_amd.jpeg import prefetch_decoded
            return prefetch_decoded(loader, self._jpeg_decoder())
        return loader

    def compute_features_one_batch(self, data):
        image, labels, index = data
        image = image.to(self.device, non_blocking=True)
        if hasattr(image, "files") or hasattr(image, "staged"):   # gpu_decode dataset: undecoded files / host-staged coefficients
            image = self._gpu_decode(image)
        if hasattr(image, "pixels"):                      # RawImageBatch from a gpu_preprocess dataset
            image = self._gpu_preprocess(image)
        labels = labels.to(self.device, non_blocking=True)
        output = self.model(image, labels) if self.config.model.get("pass_labels") else self.model(image)
        return (image, labels, index), self.parse_model_output(output)

    @staticmethod
    def _accuracies(output, labels):
        """accuracy per logits tensor, named as the reference names them (:90-101, :140-150) -- 0-dim DEVICE tensors"""
        accs = {}
        for key, val in output.items():
            if "logits" in key and torch.is_tensor(val):
                val = val.detach()
                pred = val.mean(dim=0).argmax(1) if val.dim() == 3 else val.argmax(1)
                parts = key.split("_")
                accs["acc" if len(parts) == 1 else f"acc_{parts[1]}"] = (pred == labels.argmax(1)).float().mean()
        return accs

    @staticmethod
    def _record(meters, values, n):
        """One update for all meters of the batch.  `meters` from BaseTrainer's loops is a utils.misc.DeviceMeters: the values stay
        on the GPU (two tiny launches, no `.item()`); a plain dict of AverageMeter (a caller driving the batch methods itself, as
        the reference's loops do) gets the reference's per-value host update."""
        if hasattr(meters, "update_many"):
            meters.update_many(values, n)
        else:
            for key, val in values.items():
                meters[key].update(val.item() if torch.is_tensor(val) else val, n)

    def train_one_batch(self, *args, **kwargs):
        """reference trainers/coop.py:107-154: zero_grad -> forward -> criterion -> backward -> step -> meters.  The encoder's forward
        and backward (adapter gradients included) run in the HIP library; the head and the loss on torch autograd."""
        data, meters = args
        self.optimizer.zero_grad()
        (image, labels, index), output = self.compute_features_one_batch(data)
        target = labels if self.config.dataset.get("multiclass") else labels.argmax(1)
        loss = self.criterion(output, target)
        loss.backward()
        self.optimizer.step()
        n = image.size(0)
        vals = {"loss": loss.detach()}
        vals.update({k: v.detach() for k, v in self.criterion.losses.items()})
        vals.update(self._accuracies(output, labels))
        self._record(meters, vals, n)

    def inference_one_batch(self, *args, **kwargs):
        data, meters = args
        with torch.no_grad():
            (image, labels, index), output = self.compute_features_one_batch(data)
            n = image.size(0)
            side = getattr(meters, "stream", None)
            if side is not None:
                # the bookkeeping of this batch goes to the meters' stream: it waits for the forward, reads the outputs there (recorded on
                # it for the allocator) and leaves the main stream free for the next batch's pre-processing and encode
                main = torch.cuda.current_stream(self.device)
                side.wait_stream(main)
                for t in list(output.values()) + [labels]:
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(side)
            import contextlib
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                target = labels if self.config.dataset.get("multiclass") else labels.argmax(1)
                loss = self.criterion(output, target)
                vals = {"loss": loss}
                vals.update(self.criterion.losses)
                vals.update(self._accuracies(output, labels))
                self._record(meters, vals, n)
        return {"codes": output["codes"], "labels": labels}
