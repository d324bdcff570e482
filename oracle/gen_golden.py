#!/usr/bin/env python3
"""Generate tests/golden/encode_*.npz by running the REFERENCE's own model classes (unmodified source,
imported from /root/reference through oracle/_ref_shim.py) on seeded inputs.

Run in the build container only:   python -B oracle/gen_golden.py
Outputs are data (inputs, weights, expected outputs) -- no reference source or bytecode is written.
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
from oracle import encoder_oracle as eo

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")

# name -> input resolution when it differs from the model's pretrain resolution (reference interpolate_pos_encoding,
# models/arch/coop.py:429-450: bicubic resize of the position-embedding grid)
INPUT_SIZE = {"encode_interp": 96}

FIXTURES = {
    # name: (vision dims, nbit, nclass, adapter b, center_dim, batch, hidden_act)
    "encode_tiny": (dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                         image_size=64, patch_size=16, projection_dim=32), 16, 10, 384, 32, 3, "quick_gelu"),
    "encode_hd64": (dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                         image_size=64, patch_size=16, projection_dim=64), 64, 20, 64, 48, 4, "quick_gelu"),
    # pretrain grid 4 x 4 (64 px), evaluated at 96 px (6 x 6): 1 + 36 + 4 = 41 tokens
    "encode_interp": (dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                           image_size=64, patch_size=16, projection_dim=64), 64, 20, 64, 48, 2, "quick_gelu"),
}
# Fixtures whose weights and images are rounded to bf16-representable values BEFORE the reference runs (stored as the
# uint16 bit patterns, "sdbf/..." / "inbf/...": half the bytes, and the HIP path's weight conversion is exact).
# encode_n201: image 224 / patch 16 -> 196 patches + CLS + 4 concept tokens = 201 tokens, D = 256 (4 heads x 64), so that
# the dispatched kernels of the real configs -- the 256x256 ping-pong GEMM (with CH_GEMM_PP_MIN_K=256), the KB = 7
# attention with its masked 201 -> 224 tail, the LN-fold chain and final-layer pruning -- meet REFERENCE output.
FIXTURES_BF16 = {
    "encode_n201": (dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                         image_size=224, patch_size=16, projection_dim=128), 64, 20, 256, 64, 2, "quick_gelu"),
}


def _to_bf16_bits(t: torch.Tensor) -> np.ndarray:
    return t.detach().to(torch.bfloat16).view(torch.int16).cpu().numpy().view(np.uint16)


def gen_bf16_fixture(name):
    vd, nbit, nclass, b, cdim, batch, act = FIXTURES_BF16[name]
    model = shim.build_reference_model(vd, nbit=nbit, nclass=nclass, adapter_bottleneck_dim=b, seed=7,
                                       center_dim=cdim, hidden_act=act)
    cfg = dict(D=vd["hidden_size"], L=vd["num_hidden_layers"], heads=vd["num_attention_heads"],
               M=vd["intermediate_size"], patch=vd["patch_size"], image=vd["image_size"],
               P=vd["projection_dim"], b=b)
    syn = eo.synthetic_state_dict(cfg, nbit=nbit, nclass=nclass, seed=13, center_dim=cdim)
    missing, unexpected = model.load_state_dict(syn, strict=False)
    assert not unexpected, unexpected
    model.eval()
    with torch.no_grad():
        for v in model.state_dict().values():          # shares storage with the parameters / buffers (aliases included)
            if v.is_floating_point():
                v.copy_(v.to(torch.bfloat16).float())
    x = eo.synthetic_images(batch, vd["image_size"], seed=5).to(torch.bfloat16).float()
    with torch.no_grad():
        feats, out = model(x)
    sd = model.state_dict()
    payload = {}
    for k, v in sd.items():
        if (k.startswith("adapter_params.") or k.startswith("trainable_params.") or k.startswith("backbone.text_projection")
                or k.startswith("backbone.text_model.") or k in ("backbone.logit_scale", "hash_bn.num_batches_tracked")):
            continue
        if v.is_floating_point():
            assert torch.equal(v.to(torch.bfloat16).float(), v.float()), k
            payload["sdbf/" + k] = _to_bf16_bits(v)
        else:
            payload["sd/" + k] = v.cpu().numpy()
    payload["meta/heads"] = np.int64(vd["num_attention_heads"])
    payload["meta/upt_heads"] = np.int64(8)
    payload["meta/act"] = np.array(act)
    payload["inbf/images"] = _to_bf16_bits(x)
    for key in ("codes", "hash_features", "logits_cont", "logits_bin", "logits_concept"):
        payload["out/" + key] = out[key].numpy()
    payload["out/image_features"] = feats.numpy()
    hs = out["image_hidden_states"]
    payload["out/h0"] = hs[0].numpy()
    payload["out/h1"] = hs[1].numpy()
    payload["out/h_last"] = hs[-1].numpy()
    Q = 4
    payload["out/concept_attn_last"] = out["attn_cache"][-1][:, :, -Q:, 1:-Q].numpy()
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **payload)
    print(name, "->", path, f"{os.path.getsize(path) / 1e6:.2f} MB", "codes", out["codes"].shape,
          "|codes| mean", float(out["codes"].abs().mean()))


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    only = set(sys.argv[1:])          # optional: fixture names to (re)generate; default = all
    for name in FIXTURES_BF16:
        if not only or name in only:
            gen_bf16_fixture(name)
    for name, (vd, nbit, nclass, b, cdim, batch, act) in FIXTURES.items():
        if only and name not in only:
            continue
        model = shim.build_reference_model(vd, nbit=nbit, nclass=nclass, adapter_bottleneck_dim=b, seed=7,
                                           center_dim=cdim, hidden_act=act)
        cfg = dict(D=vd["hidden_size"], L=vd["num_hidden_layers"], heads=vd["num_attention_heads"],
                   M=vd["intermediate_size"], patch=vd["patch_size"], image=vd["image_size"],
                   P=vd["projection_dim"], b=b)
        # randomise everything the reference zero/one-initialises (adapters' up_proj, LN, BN stats) so the
        # whole path is exercised; keys follow the reference state_dict layout.
        syn = eo.synthetic_state_dict(cfg, nbit=nbit, nclass=nclass, seed=11, center_dim=cdim)
        missing, unexpected = model.load_state_dict(syn, strict=False)
        assert not unexpected, unexpected
        model.eval()
        x = eo.synthetic_images(batch, INPUT_SIZE.get(name, vd["image_size"]), seed=5)
        with torch.no_grad():
            feats, out = model(x)
        sd = model.state_dict()
        all_keys = sorted(sd.keys())
        keep = {k: v.detach().cpu().numpy() for k, v in sd.items()
                if not k.startswith("adapter_params.") and not k.startswith("trainable_params.")
                and not k.startswith("backbone.text_projection") and k != "backbone.logit_scale"
                and k != "hash_bn.num_batches_tracked"}
        payload = {"sd/" + k: v for k, v in keep.items()}
        payload["meta/all_state_dict_keys"] = np.array(all_keys)
        payload["meta/heads"] = np.int64(vd["num_attention_heads"])
        payload["meta/upt_heads"] = np.int64(8)
        payload["meta/act"] = np.array(act)
        payload["in/images"] = x.numpy()
        payload["out/codes"] = out["codes"].numpy()
        payload["out/hash_features"] = out["hash_features"].numpy()
        payload["out/logits_cont"] = out["logits_cont"].numpy()
        payload["out/logits_bin"] = out["logits_bin"].numpy()
        payload["out/logits_concept"] = out["logits_concept"].numpy()
        payload["out/image_features"] = feats.numpy()
        hs = out["image_hidden_states"]
        payload["out/h0"] = hs[0].numpy()
        payload["out/h1"] = hs[1].numpy()
        payload["out/h_last"] = hs[-1].numpy()
        payload["out/attn0"] = out["attn_cache"][0].numpy()
        path = os.path.join(GOLDEN, name + ".npz")
        np.savez_compressed(path, **payload)
        print(name, "->", path, f"{os.path.getsize(path) / 1e6:.2f} MB",
              "codes", out["codes"].shape, "|codes| mean", float(out["codes"].abs().mean()))

    if only and "cossim" not in only:
        return
    # directly importable reference module (no shim needed): CosSim
    shim.install()
    from models.layers.cossim import CosSim  # unmodified reference source
    torch.manual_seed(3)
    cs = CosSim(24, 7)
    xin = torch.randn(5, 24)
    with torch.no_grad():
        y = cs(xin)
    np.savez_compressed(os.path.join(GOLDEN, "cossim.npz"), centroids=cs.centroids.detach().numpy(),
                        x=xin.numpy(), logits=y.numpy())
    print("cossim ->", y.shape)


if __name__ == "__main__":
    main()
