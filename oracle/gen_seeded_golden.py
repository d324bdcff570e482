#!/usr/bin/env python3
"""Generate tests/golden/seeded_{vit_b16,vit_l14,vit_s16}.npz: the REFERENCE's own model (unmodified source from /root/reference, through
oracle/_ref_shim.py) at the headline size -- ViT-B/16, 12 layers, D = 768, 201 tokens, adapters b = 384 -- on SEEDED weights and images.

The weights (86 M parameters) are not stored: `concepthash_amd.synthetic.synthetic_state_dict(CONFIGS['vit_b16'], seed=...)` rebuilds them
bit for bit from the seed (CPU torch.Generator), rounded to bf16-representable values; the fixture holds a per-tensor checksum of
them, the reference's eval outputs for 2 images, and -- for the training step -- the reference's gradients for a seeded cotangent on
hash_features as SIGNATURES: per adapter tensor its L2 norm and its dot product with a seeded random direction (the full gradients
would be 57 MB).

Run in the build container only:   python -B oracle/gen_seeded_golden.py [config ...]     (a few minutes of CPU for all three)
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np
import torch

import _ref_shim as shim

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
from oracle.seeded import COT_SEED, DIR_SEED, IMG_SEED, SD_SEED, SETUPS, direction, seeded_inputs, signature


def main():
    for config in (sys.argv[1:] or list(SETUPS)):
        generate(config)


def generate(CONFIG):
    NBIT, NCLASS, BATCH = SETUPS[CONFIG]
    cfg, sd, x, cot = seeded_inputs(CONFIG)
    vd = dict(hidden_size=cfg["D"], intermediate_size=cfg["M"], num_hidden_layers=cfg["L"], num_attention_heads=cfg["heads"],
              image_size=cfg["image"], patch_size=cfg["patch"], projection_dim=cfg["P"])
    model = shim.build_reference_model(vd, nbit=NBIT, nclass=NCLASS, adapter_bottleneck_dim=cfg["b"], seed=1, center_dim=sd["center"].shape[1],
                                       hidden_act="quick_gelu", upt_dropout=0.0)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    # every key the synthetic state dict does not carry must be one the path never reads: the ParameterDict ALIASES of tensors that
    # were loaded under their module names (same Parameter objects), and the text-side leftovers of the CLIP wrapper.  Anything else
    # would silently keep the reference's seed-1 initial value while the HIP engine uses the seeded one (checked here in round 3:
    # exactly these for all three configs).
    not_on_the_path = [k for k in missing if not (k.startswith(("adapter_params.", "trainable_params."))
                                                  or k in ("backbone.logit_scale", "backbone.text_projection.weight"))]
    assert not not_on_the_path, not_on_the_path
    named = dict(model.named_parameters(remove_duplicate=False))
    loaded_ptrs = {named[k].data_ptr() for k in sd if k in named}
    assert all(named[k].data_ptr() in loaded_ptrs for k in missing if k.startswith(("adapter_params.", "trainable_params."))), \
        "an alias entry does not share storage with a loaded tensor"
    payload = {"meta/config": np.array(CONFIG), "meta/seeds": np.array([SD_SEED, IMG_SEED, COT_SEED, DIR_SEED]),
               "meta/nbit_nclass_batch": np.array([NBIT, NCLASS, BATCH])}
    for k, v in sd.items():          # checksum of the regenerated weights: sum and sum of squares in fp64
        if v.is_floating_point():
            payload["chk/" + k] = np.array([float(v.double().sum()), float(v.double().pow(2).sum())])
    model.eval()
    with torch.no_grad():
        feats, out = model(x)
    for key in ("codes", "hash_features", "logits_cont", "logits_bin", "logits_concept"):
        payload["out/" + key] = out[key].numpy()
    payload["out/image_features"] = feats.numpy()
    # training-mode forward + VJP for the cotangent on hash_features (dropout 0; BatchNorm is downstream of hash_features)
    model.train()
    model.requires_grad_(False)
    params = dict(model.get_adapter().named_parameters())
    for p in params.values():
        p.requires_grad_(True)
    model.hash_queries.requires_grad_(True)
    _, out = model(x)
    out["hash_features"].backward(cot)
    named = dict(model.named_parameters(remove_duplicate=False))
    n = 0
    for k, p in named.items():
        if ".adapt_mlp_" in k and k.startswith("backbone.") and p.grad is not None:
            payload["sig/" + k] = np.array(signature(k, p.grad))
            n += 1
    payload["sig/hash_queries"] = np.array(signature("hash_queries", model.hash_queries.grad))
    path = os.path.join(GOLDEN, f"seeded_{CONFIG}.npz")
    np.savez_compressed(path, **payload)
    print(f"seeded_{CONFIG} ->", path, f"{os.path.getsize(path) / 1e3:.1f} KB, {n} adapter gradient signatures")


if __name__ == "__main__":
    main()
