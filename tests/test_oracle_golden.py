"""CPU: the oracle (oracle/encoder_oracle.py) against golden vectors produced by the reference's own model code
(tests/golden/encode_*.npz, generator oracle/gen_golden.py).  fp32, atol 2e-5 on every tapped stage."""
import os

import numpy as np
import pytest
import torch

from conftest import fixture_images, load_fixture
from oracle import encoder_oracle as eo

ATOL = 2e-5


@pytest.mark.parametrize("name", ["encode_tiny", "encode_hd64"])
def test_encode_matches_reference_outputs(name):
    sd, z = load_fixture(name)
    stages = {}
    out = eo.encode(sd, torch.from_numpy(z["in/images"]), heads=int(z["meta/heads"]), upt_heads=int(z["meta/upt_heads"]),
                    act=str(z["meta/act"]), stages=stages)
    for key in ("codes", "hash_features", "logits_cont", "logits_bin", "logits_concept", "image_features"):
        ref = torch.from_numpy(z["out/" + key])
        assert out[key].shape == ref.shape, key
        assert torch.allclose(out[key], ref, atol=ATOL, rtol=0), (key, float((out[key] - ref).abs().max()))
    L = eo.infer_dims(sd)["L"]
    for key, st in (("h0", "h0"), ("h1", "h1"), ("h_last", f"h{L}"), ("attn0", "attn0")):
        ref = torch.from_numpy(z["out/" + key])
        assert torch.allclose(stages[st], ref, atol=ATOL, rtol=0), (key, float((stages[st] - ref).abs().max()))
    # the bits the retrieval path consumes are identical
    assert bool(((out["codes"] > 0) == (torch.from_numpy(z["out/codes"]) > 0)).all())


def test_encode_matches_reference_outputs_at_201_tokens():
    """encode_n201: the reference model at image 224 / patch 16 (201 tokens), D = 256, bf16-representable weights and images
    (models/arch/coop.py:524-598 run by oracle/gen_golden.py).  Same atol as the small fixtures."""
    sd, z = load_fixture("encode_n201")
    stages = {}
    x = fixture_images(z)
    assert x.shape == (2, 3, 224, 224)
    out = eo.encode(sd, x, heads=int(z["meta/heads"]), upt_heads=int(z["meta/upt_heads"]), act=str(z["meta/act"]), stages=stages)
    for key in ("codes", "hash_features", "logits_cont", "logits_bin", "logits_concept", "image_features"):
        ref = torch.from_numpy(z["out/" + key])
        assert out[key].shape == ref.shape, key
        assert torch.allclose(out[key], ref, atol=ATOL, rtol=0), (key, float((out[key] - ref).abs().max()))
    for key, st in (("h0", "h0"), ("h1", "h1"), ("h_last", "h2")):
        ref = torch.from_numpy(z["out/" + key])
        assert ref.shape == (2, 201, 256)
        assert torch.allclose(stages[st], ref, atol=ATOL, rtol=0), (key, float((stages[st] - ref).abs().max()))
    ca = stages["attn1"][:, :, -4:, 1:-4]
    assert torch.allclose(ca, torch.from_numpy(z["out/concept_attn_last"]), atol=ATOL, rtol=0)
    assert bool(((out["codes"] > 0) == (torch.from_numpy(z["out/codes"]) > 0)).all())


def test_non_pretrain_resolution_matches_reference_outputs():
    """encode_interp: the reference model pretrained at 64 px (4 x 4 grid) evaluated on 96 px inputs -- its
    interpolate_pos_encoding (models/arch/coop.py:429-450) resizes the position table to 6 x 6.  Oracle vs reference outputs, and
    the product's load-time fold of that table (concepthash_amd.encoder.interpolate_pos_embedding) vs the torch call."""
    from concepthash_amd.encoder import interpolate_pos_embedding
    sd, z = load_fixture("encode_interp")
    x = torch.from_numpy(z["in/images"])
    assert x.shape[-1] == 96 and sd[eo.VM + "embeddings.position_embedding.weight"].shape[0] == 17
    stages = {}
    out = eo.encode(sd, x, heads=int(z["meta/heads"]), upt_heads=int(z["meta/upt_heads"]), act=str(z["meta/act"]), stages=stages)
    for key in ("codes", "hash_features", "logits_cont", "logits_bin", "logits_concept", "image_features"):
        ref = torch.from_numpy(z["out/" + key])
        assert torch.allclose(out[key], ref, atol=ATOL, rtol=0), (key, float((out[key] - ref).abs().max()))
    assert stages["h0"].shape == (2, 1 + 36 + 4, 128)
    assert torch.allclose(stages["h0"], torch.from_numpy(z["out/h0"]), atol=ATOL, rtol=0)
    pos = sd[eo.VM + "embeddings.position_embedding.weight"]
    for new_grid in (6, 3, 9):
        want = eo.interpolate_pos_encoding(pos, new_grid * 16, new_grid * 16, 16)
        got = interpolate_pos_embedding(pos, new_grid)
        assert got.shape == want.shape and torch.allclose(got, want, atol=1e-5, rtol=0), float((got - want).abs().max())
    assert torch.equal(interpolate_pos_embedding(pos, 4), pos)


def test_state_dict_key_layout_matches_reference():
    """Every reference state_dict key is either consumed by the oracle/HIP loader or a documented alias."""
    sd, z = load_fixture("encode_tiny")
    all_keys = [str(k) for k in z["meta/all_state_dict_keys"]]
    from concepthash_amd.encoder import _SKIP_KEYS, _SKIP_PREFIXES
    used = set(sd.keys())
    for k in all_keys:
        assert k in used or k.startswith(_SKIP_PREFIXES) or k in _SKIP_KEYS, k
    # aliases really are aliases of tensors we keep (adapter_params.adapter_N_x_y <-> ...layers.N.x.y)
    assert any(k.startswith("adapter_params.adapter_0_adapt_mlp_1_") for k in all_keys)
    assert "trainable_params.hash_pe" in all_keys and "hash_pe" in used


def test_bf16_emulation_stays_close_and_keeps_bits():
    sd, z = load_fixture("encode_hd64")
    out = eo.encode(sd, torch.from_numpy(z["in/images"]), heads=int(z["meta/heads"]), emulate_bf16=True)
    ref = torch.from_numpy(z["out/codes"])
    assert float((out["codes"] - ref).abs().max()) < 5e-3


def test_cossim_fixture():
    """CosSim.forward (models/layers/cossim.py:37-82, group=1) -- fixture from the directly importable module."""
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "cossim.npz"))
    x, cen = torch.from_numpy(z["x"]), torch.from_numpy(z["centroids"])
    got = torch.nn.functional.normalize(x, dim=-1) @ torch.nn.functional.normalize(cen, dim=-1).t()
    assert torch.allclose(got, torch.from_numpy(z["logits"]), atol=1e-6)


def test_synthetic_state_dict_roundtrip_through_oracle():
    cfg = eo.CONFIGS["tiny"]
    sd = eo.synthetic_state_dict(cfg, nbit=16, nclass=10, center_dim=32)
    out = eo.encode(sd, eo.synthetic_images(2, cfg["image"]), heads=cfg["heads"])
    assert out["codes"].shape == (2, 16) and torch.isfinite(out["codes"]).all()
    d = eo.infer_dims(sd)
    assert (d["D"], d["L"], d["Q"], d["nbit"], d["C"]) == (64, 2, 4, 16, 10)


def test_training_step_matches_reference_losses_and_gradients():
    """train_tiny: one training step of the reference's own model + its own LGHLoss (trainers/coop.py:107-131,
    models/loss/coop.py:120-189; generated by oracle/gen_train_golden.py with dropout 0).  The restatement in
    oracle/train_oracle.py must reproduce the loss terms, the train-mode outputs (BatchNorm on batch statistics) and the
    gradient of EVERY trainable tensor (adapters, concept-token generator, hashing head, text projection)."""
    from oracle import train_oracle as to
    sd, z = load_fixture("train_tiny")
    x = fixture_images(z)
    labels = torch.from_numpy(z["in/labels"])
    res = to.train_step_grads(sd, x, labels, heads=int(z["meta/heads"]), upt_heads=int(z["meta/upt_heads"]), act=str(z["meta/act"]))
    assert abs(float(res["loss"]) - float(z["out/loss"])) < 2e-5
    for k in ("concept", "cont", "bin"):
        assert abs(float(res["losses"][k]) - float(z["out/loss_" + k])) < 2e-5, k
    for k in ("codes", "hash_features", "logits_cont", "logits_bin", "logits_concept"):
        ref = torch.from_numpy(z["out/" + k])
        # codes: BatchNorm over a batch of 6 divides by small batch deviations and amplifies fp32 summation-order noise
        atol = 2e-4 if k == "codes" else ATOL
        assert torch.allclose(res["out"][k], ref, atol=atol, rtol=0), (k, float((res["out"][k] - ref).abs().max()))
    gkeys = [k[5:] for k in z.files if k.startswith("grad/")]
    assert len(gkeys) == 53 and set(gkeys) == set(to.trainable_keys(sd)), set(gkeys) ^ set(to.trainable_keys(sd))
    for k in gkeys:
        ref = torch.from_numpy(z["grad/" + k])
        got = res["grads"][k]
        assert got.shape == ref.shape, k
        scale = float(ref.abs().max())
        # hash_pe: a per-(concept, bit) constant over the batch, removed again by the train-mode BatchNorm -> its true gradient is 0
        # and both sides hold fp32 noise of ~2e-5
        # 2e-3 of the tensor's largest gradient: the train-mode BatchNorm over 6 samples divides by small batch deviations; an fp64
        # run of this oracle sits 3e-4..9e-4 (relative) from BOTH the reference's fp32 gradients and this oracle's fp32 gradients
        assert float((got - ref).abs().max()) <= 5e-5 + 2e-3 * scale, (k, float((got - ref).abs().max()), scale)
    # running statistics after the step (BatchNorm1d momentum 0.1, unbiased variance)
    rm = 0.9 * sd["hash_bn.running_mean"] + 0.1 * res["out"]["bn_batch_mean"]
    rv = 0.9 * sd["hash_bn.running_var"] + 0.1 * res["out"]["bn_batch_var_unbiased"]
    assert torch.allclose(rm, torch.from_numpy(z["out/bn_running_mean"]), atol=1e-5)
    assert torch.allclose(rv, torch.from_numpy(z["out/bn_running_var"]), atol=1e-5)


def test_attention_diversity_term_matches_reference():
    """train_tiny `attn/*`: the same step with loss_scales.attn_div_loss = 25 -- the reference's loss reads
    attn_cache[-1][:, :, -Q:, 1:-Q] (models/loss/coop.py:161-187) and its gradient enters the last layer's attention."""
    from oracle import train_oracle as to
    sd, z = load_fixture("train_tiny")
    x = fixture_images(z)
    labels = torch.from_numpy(z["in/labels"])
    res = to.train_step_grads(sd, x, labels, heads=int(z["meta/heads"]), upt_heads=8, act=str(z["meta/act"]), attn_div_scale=25.0)
    ca = torch.from_numpy(z["attn/concept_attention"])
    assert torch.allclose(res["out"]["concept_attention"], ca, atol=2e-6)
    assert abs(float(res["losses"]["attn_div"]) - float(z["attn/loss_attn_div"])) < 2e-6
    assert abs(float(res["loss"]) - float(z["attn/loss"])) < 3e-5
    keys = [k[9:] for k in z.files if k.startswith("attngrad/")]
    assert len(keys) == 28 + 14 + 1          # 4 adapters x 7 tensors, hash_attention x 14, hash_queries
    changed = 0
    for k in keys:
        ref = torch.from_numpy(z["attngrad/" + k])
        got = res["grads"][k]
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= 5e-5 + 2e-3 * scale, (k, float((got - ref).abs().max()), scale)
        if ".layers.1.adapt_mlp_" not in k:       # the term acts through the last layer's attention: everything upstream of it moves
            changed += float((ref - torch.from_numpy(z["grad/" + k])).abs().max()) > 1e-3 * scale
    assert changed >= 10


def test_attention_diversity_term_with_avg_attn_matches_reference():
    """train_tiny `attnavg/*`: the reference's loss with `avg_attn: True` (models/loss/coop.py:164-167) averages EVERY layer's
    attention map; `attnavg/concept_attention_layers` is torch.stack(attn_cache)[:, :, :, -Q:, 1:-Q] of the reference's own forward."""
    from oracle import train_oracle as to
    sd, z = load_fixture("train_tiny")
    x = fixture_images(z)
    labels = torch.from_numpy(z["in/labels"])
    res = to.train_step_grads(sd, x, labels, heads=int(z["meta/heads"]), upt_heads=8, act=str(z["meta/act"]), attn_div_scale=25.0,
                              avg_attn=True)
    rows = torch.from_numpy(z["attnavg/concept_attention_layers"])
    assert rows.shape[0] == 2 and torch.allclose(res["out"]["concept_attention_layers"], rows, atol=2e-6)
    assert torch.equal(rows[-1], torch.from_numpy(z["attn/concept_attention"]))
    assert abs(float(res["losses"]["attn_div"]) - float(z["attnavg/loss_attn_div"])) < 2e-6
    assert abs(float(res["loss"]) - float(z["attnavg/loss"])) < 3e-5
    assert abs(float(z["attnavg/loss_attn_div"]) - float(z["attn/loss_attn_div"])) > 1e-4      # a different term from the last-layer one
    keys = [k[12:] for k in z.files if k.startswith("attnavggrad/")]
    assert len(keys) == 28 + 14 + 1
    for k in keys:
        ref = torch.from_numpy(z["attnavggrad/" + k])
        got = res["grads"][k]
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= 5e-5 + 2e-3 * scale, (k, float((got - ref).abs().max()), scale)


@pytest.mark.parametrize("config", ["vit_b16", "vit_s16", "vit_l14"])
def test_full_size_seeded_fixtures_pin_the_oracle(config):
    """tests/golden/seeded_<config>.npz: the reference's own model at the BASELINE.json model sizes (ViT-B/16 x 12 layers, 201 tokens;
    ViT-S/16 x 12; ViT-L/14 x 24, 261 tokens) on weights rebuilt from a seed (checksummed), 2 images: eval outputs, and -- training
    mode -- gradient signatures (norm, projection on a seeded direction) of every adapter tensor for a seeded cotangent on
    hash_features."""
    from conftest import GOLDEN
    from oracle import seeded as gen
    from oracle import train_oracle as to
    z = np.load(os.path.join(GOLDEN, f"seeded_{config}.npz"))
    cfg, sd, x, cot = gen.seeded_inputs(config)
    for k, v in sd.items():
        if v.is_floating_point():
            chk = z["chk/" + k]
            assert abs(float(v.double().sum()) - chk[0]) <= 1e-9 * max(1.0, abs(chk[0])) and \
                abs(float(v.double().pow(2).sum()) - chk[1]) <= 1e-9 * max(1.0, chk[1]), f"seeded weights drifted: {k}"
    out = eo.encode(sd, x, heads=cfg["heads"])
    for key in ("codes", "hash_features", "logits_cont", "logits_bin", "logits_concept", "image_features"):
        ref = torch.from_numpy(z["out/" + key])
        assert torch.allclose(out[key], ref, atol=2e-4, rtol=0), (key, float((out[key] - ref).abs().max()))
    sdg = {k: v.clone() for k, v in sd.items()}
    keys = [k for k in to.trainable_keys(sdg) if ".adapt_mlp_" in k] + ["hash_queries"]
    for k in keys:
        sdg[k] = sdg[k].float().requires_grad_(True)
    to.forward_train(sdg, x, heads=cfg["heads"])["hash_features"].backward(cot)
    assert len([k for k in z.files if k.startswith("sig/")]) == 14 * cfg["L"] + 1
    for k in keys:
        norm, proj = gen.signature(k, sdg[k].grad)
        rn, rp = z["sig/" + k]
        assert abs(norm - rn) <= 2e-3 * rn and abs(proj - rp) <= 2e-3 * rn, (k, norm, rn, proj, rp)
