"""`engine` helpers the trainers use (reference engine.py:41-61): DataLoader factory and seeding."""
from __future__ import annotations

import os
import random

import numpy as np
import torch
from torch.utils.data import DataLoader

default_workers = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))


def dataloader(d, bs=256, shuffle=False, workers=-1, drop_last=False, sampler=None):
    if len(d) == 0:
        return []
    if workers < 0:
        workers = default_workers
    if getattr(d, "in_memory", False):      # tensor-backed datasets need no worker processes
        workers = 0
    return DataLoader(d, bs, shuffle, drop_last=drop_last, num_workers=workers, sampler=sampler,
                      pin_memory=workers > 0, collate_fn=getattr(d, "collate_fn", None))


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
