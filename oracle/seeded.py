"""Seeded full-size inputs shared by oracle/gen_seeded_golden.py (which runs the REFERENCE on them) and the tests (which rebuild them
from the seeds): TEST INFRASTRUCTURE, not product code.  No reference import here."""
from __future__ import annotations

import torch

from concepthash_amd import synthetic

SD_SEED, IMG_SEED, COT_SEED, DIR_SEED = 77, 5, 6, 7
# config -> (nbit, nclass, batch): the BASELINE.json model sizes
SETUPS = {"vit_b16": (64, 200, 2), "vit_l14": (128, 555, 2), "vit_s16": (16, 200, 2)}


def seeded_inputs(config="vit_b16"):
    """(cfg, state_dict, images, cotangent): bf16-representable weights and images of a synthetic.CONFIGS entry, rebuilt from the seeds"""
    nbit, nclass, batch = SETUPS[config]
    cfg = synthetic.CONFIGS[config]
    sd = synthetic.synthetic_state_dict(cfg, nbit=nbit, nclass=nclass, seed=SD_SEED)
    sd = {k: (v.to(torch.bfloat16).float() if v.is_floating_point() else v) for k, v in sd.items()}
    x = synthetic.synthetic_images(batch, cfg["image"], seed=IMG_SEED).to(torch.bfloat16).float()
    cot = torch.randn(batch, 4, cfg["D"], generator=torch.Generator().manual_seed(COT_SEED))
    return cfg, sd, x, cot


def direction(name, shape):
    """seeded unit-variance direction for the gradient signature of tensor `name`"""
    g = torch.Generator().manual_seed(DIR_SEED + sum(name.encode()) * 7919 % 1000003)
    return torch.randn(*shape, generator=g)


def signature(name, grad):
    """(L2 norm, projection on direction(name)) in fp64"""
    g = grad.detach().double().cpu()
    return float(g.norm()), float((g * direction(name, g.shape).double()).sum())
