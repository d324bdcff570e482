"""Seeded synthetic workloads for benchmarks and tests (no reference data exists offline: SURVEY.md F7 / section 8d).

Model weights in the reference's state_dict layout, post-normalisation image batches, and packed hash codes with class
structure.  Pure generators -- no arithmetic of the encode/retrieve path lives here.  Used by bench.py, tools/ and the
tests; the oracle re-exports them.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

VM = "backbone.vision_model."

CONFIGS = {
    # name: D, L, heads, M(ffn), patch, image, P(projection), b(adapter)
    "tiny": dict(D=64, L=2, heads=4, M=128, patch=16, image=64, P=32, b=384),
    "vit_s16": dict(D=384, L=12, heads=6, M=1536, patch=16, image=224, P=512, b=384),
    "vit_b32": dict(D=768, L=12, heads=12, M=3072, patch=32, image=224, P=512, b=384),
    "vit_b16": dict(D=768, L=12, heads=12, M=3072, patch=16, image=224, P=512, b=384),
    "vit_l14": dict(D=1024, L=24, heads=16, M=4096, patch=14, image=224, P=768, b=384),
}


def synthetic_state_dict(cfg: dict, nbit: int, nclass: int, Q: int = 4, seed: int = 42,
                         center_dim: int = 512) -> Dict[str, torch.Tensor]:
    """Seeded random weights: N(0,0.02) linears, LN gamma=1+N(0,.02) beta=N(0,.02), non-zero adapter up-proj
    (the reference zero-inits up_proj, models/layers/adapter.py:42, which would make adapters a no-op),
    BN running mean N(0,0.1) / var U(0.5,1.5)."""
    g = torch.Generator().manual_seed(seed)
    D, L, M, p, P, b = cfg["D"], cfg["L"], cfg["M"], cfg["patch"], cfg["P"], cfg["b"]
    npos = (cfg["image"] // p) ** 2 + 1

    def n(*shape, std=0.02):
        return torch.randn(*shape, generator=g) * std

    sd = {}

    def ln(prefix, dim):
        sd[prefix + ".weight"] = 1.0 + n(dim)
        sd[prefix + ".bias"] = n(dim)

    def lin(prefix, out_f, in_f, bias=True, std=0.02):
        sd[prefix + ".weight"] = n(out_f, in_f, std=std)
        if bias:
            sd[prefix + ".bias"] = n(out_f)

    sd[VM + "embeddings.class_embedding"] = n(D)
    sd[VM + "embeddings.patch_embedding.weight"] = n(D, 3, p, p)
    sd[VM + "embeddings.position_embedding.weight"] = n(npos, D)
    ln(VM + "pre_layrnorm", D)
    for i in range(L):
        pre = VM + f"encoder.layers.{i}."
        for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
            lin(pre + f"self_attn.{nm}", D, D)
        ln(pre + "layer_norm1", D)
        lin(pre + "mlp.fc1", M, D)
        lin(pre + "mlp.fc2", D, M)
        ln(pre + "layer_norm2", D)
        for a in ("adapt_mlp_1.", "adapt_mlp_2."):
            sd[pre + a + "scale"] = torch.ones(1) + n(1)
            ln(pre + a + "adapter_layer_norm", D)
            lin(pre + a + "down_proj", b, D)
            lin(pre + a + "up_proj", D, b)
    ln(VM + "post_layernorm", D)
    sd["backbone.visual_projection.weight"] = n(P, D)
    sd["hash_pe"] = n(1, Q, D, std=1.0)
    sd["hash_queries"] = n(1, Q, P, std=1.0)
    sd["concept_pe"] = n(1, Q, D)
    sd["center"] = torch.randn(nclass, center_dim, generator=g).sign()
    sd["hash_fc.weight"] = n(nbit // Q, D, std=0.05)
    sd["hash_bn.weight"] = 1.0 + n(nbit)
    sd["hash_bn.bias"] = n(nbit)
    sd["hash_bn.running_mean"] = n(nbit, std=0.1)
    sd["hash_bn.running_var"] = 0.5 + torch.rand(nbit, generator=g)
    sd["hash_attention.sa.in_proj_weight"] = n(3 * P, P)
    sd["hash_attention.sa.in_proj_bias"] = n(3 * P)
    lin("hash_attention.sa.out_proj", P, P)
    lin("hash_attention.ffn.0", P, P)
    lin("hash_attention.ffn.3", P, P)
    ln("hash_attention.norm1", P)
    ln("hash_attention.norm2", P)
    lin("hash_attention.ffn2", D, P)
    sd["concept_ce.centroids"] = n(nclass, D, std=1.0)
    lin("text_projection.0", center_dim, center_dim)
    lin("text_projection.2", nbit, center_dim)
    return sd


def synthetic_images(batch: int, image: int, seed: int = 42) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randn(batch, 3, image, image, generator=g)


def synthetic_codes(rows: int, nbit: int, seed: int = 1234, nclass: int = 0, flip: float = 0.1):
    """i.i.d. Bernoulli(0.5) bits, or (nclass > 0) clustered: class centre + `flip` bit noise.  Returns
    (packed uint64 [rows, W] -- bit i of a code in word i // 64 at position i % 64 --, labels int32 [rows])."""
    rng = np.random.default_rng(seed)
    W = (nbit + 63) // 64
    if nclass > 0:
        # class centres depend on (nclass, nbit) only, so query and gallery sets drawn with different seeds share them
        centres = np.random.default_rng(1000003 * nclass + nbit).integers(0, 2, size=(nclass, nbit), dtype=np.uint8)
        labels = rng.integers(0, nclass, size=rows, dtype=np.int32)
        bits = centres[labels] ^ (rng.random((rows, nbit)) < flip).astype(np.uint8)
    else:
        labels = rng.integers(0, 1 << 30, size=rows, dtype=np.int32)
        bits = rng.integers(0, 2, size=(rows, nbit), dtype=np.uint8)
    pad = W * 64 - nbit
    if pad:
        bits = np.concatenate([bits, np.zeros((rows, pad), np.uint8)], axis=1)
    out = np.ascontiguousarray(np.packbits(bits, axis=1, bitorder="little")).view("<u8").reshape(rows, W)
    return out, labels
