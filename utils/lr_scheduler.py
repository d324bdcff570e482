"""`utils.lr_scheduler` -- the two schedulers the reference configs name (configs/scheduler/csw.yaml, no_decay.yaml).

The reference imports them from an un-vendored `utils` package that is absent from its snapshot, so their exact
form is UNPINNED; these are the standard definitions the names say, stepped once per epoch (trainers/base.py:356):
  cosine_decay_linear_warmup: lr factor (e + 1) / warmup_epochs for e < warmup_epochs, then
                              0.5 (1 + cos(pi (e - warmup) / (epochs - warmup)));
  no_decay:                   factor 1.
"""
from __future__ import annotations

import math

from torch.optim.lr_scheduler import LambdaLR


def cosine_decay_linear_warmup(optimizer, epochs: int, warmup_epochs: int = 10):
    epochs, warmup_epochs = int(epochs), int(warmup_epochs)

    def factor(e):
        if e < warmup_epochs:
            return (e + 1) / max(1, warmup_epochs)
        return 0.5 * (1.0 + math.cos(math.pi * (e - warmup_epochs) / max(1, epochs - warmup_epochs)))

    return LambdaLR(optimizer, factor)


def no_decay(optimizer):
    return LambdaLR(optimizer, lambda e: 1.0)
