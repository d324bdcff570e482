"""Class centres for `LGHWithFixedPrompt(fixed_center=...)`.

The reference builds the (C, 512) centre buffer from CLIP TEXT features of the class names
(trainers/orthohash.py:94-260 `get_codebook`, codebook_method "L"), which needs the CLIP text tower -- outside this path.  Here the
buffer is data: a tensor file produced elsewhere (e.g. by the reference) when `path` exists, otherwise seeded random +-1 rows
(enough to train and evaluate end to end on synthetic data; a checkpoint's own `center` always overrides it on load)."""
from __future__ import annotations

import logging
import os

import torch


def class_centers(nclass: int, dim: int = 512, path: str = None, seed: int = 0) -> torch.Tensor:
    if path and os.path.exists(str(path)):
        c = torch.load(str(path), map_location="cpu")
        c = c["center"] if isinstance(c, dict) else c
        c = torch.as_tensor(c, dtype=torch.float32)
        if tuple(c.shape) != (int(nclass), int(dim)):
            raise ValueError(f"{path}: centre tensor has shape {tuple(c.shape)}, expected {(int(nclass), int(dim))}")
        return c
    logging.info("class centres: %s not found -> seeded random +-1 rows (%d x %d)", path, nclass, dim)
    g = torch.Generator().manual_seed(int(seed) + 101)
    return torch.randn(int(nclass), int(dim), generator=g).sign()
