"""`engine` helpers the trainers use (reference engine.py:41-61): DataLoader factory and seeding."""
from __future__ import annotations

import os
import random

import numpy as np
import torch
from torch.utils.data import DataLoader

from concepthash_amd.hostcpu import cpu_budget, limit_torch_threads

default_workers = min(16, cpu_budget())     # the container's CPU quota, not the cores it can see (hostcpu.py)


_forkserver_ready = False


def _worker_context(workers):
    """How loader workers are started.  On MI355X / ROCm a process that has initialised the GPU must not FORK workers: while forked
    children are alive the parent's GPU work runs ~100x slower (measured with tools/loader_probe.py: the same batch of copies + kernels
    takes 2.0 s instead of 8 ms; every copy-on-write fault of the parent is an MMU-notifier round trip through the GPU driver) -- the
    reference's default `DataLoader(num_workers=...)` (engine.py:48-53) forks.  So whenever a GPU is present the workers come from a
    FORKSERVER (a clean helper process that has imported torch and the dataset module once, so a worker starts in milliseconds, not
    the seconds a `spawn` re-import takes).  CPU-only runs keep the platform default."""
    global _forkserver_ready
    if workers <= 0 or not torch.cuda.is_available():
        return None
    import multiprocessing as mp
    if not _forkserver_ready:
        try:
            mp.set_forkserver_preload(["torch", "numpy", "utils.datasets", "utils.transforms"])
        except Exception:
            pass
        _forkserver_ready = True
        try:
            # start the server NOW (its `import torch` takes a second or more): the first loader is created well before it is iterated
            # -- the model build and the checkpoint load sit in between -- so the start-up runs beside them instead of in front of the
            # first batch
            from multiprocessing import forkserver
            forkserver.ensure_running()
        except Exception:
            pass
    return mp.get_context("forkserver")


class FileBatchLoader:
    """The loader of a `gpu_decode` dataset: nothing is decoded on the CPU, so what the reference's worker processes would do
    (engine.py:48-53) is reading files -- done here by `HashingDataset._read_batch` through the library's host helpers (a few threads
    of plain reads, no Python in the loop).  No worker processes: nothing to start per loader (0.35-1.1 s with six forkserver workers,
    a third of a CUB-sized epoch at this path's rate), no shared-memory hop per batch, no forkserver.  Iterating is synchronous -- the
    asynchrony is `concepthash_amd.jpeg.prefetch_decoded`'s fetch thread, which the trainers put around a `gpu_decode` loader
    (`COOPTrainer.iterate_loader`).  Same batches as `DataLoader(d, bs, shuffle, drop_last=..., sampler=...)`: the torch samplers draw
    from the same generators."""
    num_workers = 0

    def __init__(self, dataset, batch_sampler):
        self.dataset, self.batch_sampler = dataset, batch_sampler
        self.batch_size = getattr(batch_sampler, "batch_size", None)

    def __len__(self):
        return len(self.batch_sampler)

    def __iter__(self):
        # a DataLoader draws one int64 from the global generator per epoch (the base seed of its workers) before its sampler draws:
        # drawn and dropped here, so that one `torch.manual_seed` gives this loader the batches -- and, loading in-process, the crops and
        # flips -- of `DataLoader(d, bs, shuffle, num_workers=0)` (tests/test_preprocess.py)
        import torch
        torch.empty((), dtype=torch.int64).random_()
        for indices in self.batch_sampler:
            yield self.dataset[list(indices)]


def dataloader(d, bs=256, shuffle=False, workers=-1, drop_last=False, sampler=None):
    if len(d) == 0:
        return []
    limit_torch_threads()                   # torch's CPU pool inside the container's quota before any loader thread starts (hostcpu.py)
    if workers < 0:
        workers = default_workers
    if getattr(d, "in_memory", False):      # tensor-backed datasets need no worker processes
        workers = 0
    if getattr(d, "gpu_decode", False):
        # `gpu_decode` datasets fetch a whole BATCH per call (one buffer, no per-item tensors, no collate) and need no worker processes:
        # FileBatchLoader.  (`workers` > 0 with `file_workers=True` on the dataset keeps the earlier arrangement -- DataLoader workers
        # that only read files, persistent across epochs -- for storage where a read blocks for long.)
        from torch.utils.data import BatchSampler, RandomSampler, SequentialSampler
        base = sampler if sampler is not None else (RandomSampler(d) if shuffle else SequentialSampler(d))
        if not getattr(d, "file_workers", False):
            return FileBatchLoader(d, BatchSampler(base, bs, drop_last))
        workers = min(workers, 6)
        return DataLoader(d, batch_size=None, sampler=BatchSampler(base, bs, drop_last), num_workers=workers, pin_memory=False,
                          multiprocessing_context=_worker_context(workers), persistent_workers=workers > 0)
    return DataLoader(d, bs, shuffle, drop_last=drop_last, num_workers=workers, sampler=sampler,
                      pin_memory=workers > 0, collate_fn=getattr(d, "collate_fn", None), multiprocessing_context=_worker_context(workers))


class _SequentialSubset(torch.utils.data.Sampler):
    def __init__(self, indices):
        self.indices = list(indices)

    def __iter__(self):
        return iter(self.indices)

    def __len__(self):
        return len(self.indices)


def get_sequential_sampler(idxs):
    """Fixed-order subset sampler (reference engine.py:36-38)."""
    return _SequentialSubset(idxs)


def seeding(seed):
    if seed != -1:
        torch.manual_seed(seed)
        np.random.seed(seed)
        random.seed(seed)
