#!/usr/bin/env python3
"""Generate tests/golden/train_tiny.npz: one training step of the REFERENCE's own model (unmodified source, imported from
/root/reference through oracle/_ref_shim.py) with the reference's own LGHLoss -- loss terms and the gradient of every
trainable parameter -- on seeded, bf16-representable weights and inputs.

Run in the build container only:   python -B oracle/gen_train_golden.py
Outputs are data (inputs, weights, expected losses / gradients); no reference source or bytecode is written.
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
VD = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, image_size=64, patch_size=16,
          projection_dim=64)
NBIT, NCLASS, B_ADAPTER, CDIM, BATCH, ACT = 64, 20, 64, 48, 6, "quick_gelu"


def bits(t):
    return t.detach().to(torch.bfloat16).view(torch.int16).cpu().numpy().view(np.uint16)


def main():
    model = shim.build_reference_model(VD, nbit=NBIT, nclass=NCLASS, adapter_bottleneck_dim=B_ADAPTER, seed=7, center_dim=CDIM,
                                       hidden_act=ACT, upt_dropout=0.0)
    cfg = dict(D=VD["hidden_size"], L=VD["num_hidden_layers"], heads=VD["num_attention_heads"], M=VD["intermediate_size"],
               patch=VD["patch_size"], image=VD["image_size"], P=VD["projection_dim"], b=B_ADAPTER)
    syn = eo.synthetic_state_dict(cfg, nbit=NBIT, nclass=NCLASS, seed=23, center_dim=CDIM)
    missing, unexpected = model.load_state_dict(syn, strict=False)
    assert not unexpected, unexpected
    with torch.no_grad():
        for v in model.state_dict().values():
            if v.is_floating_point():
                v.copy_(v.to(torch.bfloat16).float())
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = eo.synthetic_images(BATCH, VD["image_size"], seed=9).to(torch.bfloat16).float()
    labels = torch.tensor([3, 7, 7, 0, 19, 12])

    # what the reference trains (trainers/base.py:133-152, backbone_lr_scale 0, has_adapter True)
    params = list(model.get_adapter().parameters()) + [p for p in model.get_training_modules().parameters() if p is not None]
    model.requires_grad_(False)
    for p in params:
        p.requires_grad_(True)
    model.train()
    from models.loss.coop import LGHLoss  # unmodified reference source (importable as is)
    crit = LGHLoss(margin=0.2, scale=8, loss_scales=dict(logits=0, hash_logits=0, bin_logits=1, cont_logits=1, l2=0, attn_div_loss=0,
                                                         concept_logits=1), avg_before_softmax=False, lmbd=0.5, div_method=1, ncontext=4)
    feats, out = model(x)
    loss = crit(out, labels)
    loss.backward()

    payload = {}
    skip = ("adapter_params.", "trainable_params.", "backbone.text_projection", "backbone.text_model.")
    for k, v in sd0.items():
        if k.startswith(skip) or k in ("backbone.logit_scale", "hash_bn.num_batches_tracked"):
            continue
        if v.is_floating_point():
            assert torch.equal(v.to(torch.bfloat16).float(), v.float()), k
            payload["sdbf/" + k] = bits(v)
        else:
            payload["sd/" + k] = v.cpu().numpy()
    payload["meta/heads"] = np.int64(VD["num_attention_heads"])
    payload["meta/upt_heads"] = np.int64(8)
    payload["meta/act"] = np.array(ACT)
    payload["inbf/images"] = bits(x)
    payload["in/labels"] = labels.numpy()
    payload["out/loss"] = loss.detach().numpy()
    for k, v in crit.losses.items():
        payload["out/loss_" + k] = v.detach().numpy()
    for key in ("codes", "hash_features", "logits_cont", "logits_bin", "logits_concept"):
        payload["out/" + key] = out[key].detach().numpy()
    named = dict(model.named_parameters())
    ngrad = 0
    for k, p in named.items():
        if k.startswith(("adapter_params.", "trainable_params.")):
            continue                      # aliases of parameters that also appear under their module path
        if p.grad is not None:
            payload["grad/" + k] = p.grad.detach().numpy()
            ngrad += 1
    # parameters that named_parameters() reports only under their ParameterDict alias keep their state_dict names
    for k in ("hash_queries", "hash_pe", "concept_pe", "concept_ce.centroids"):
        if "grad/" + k not in payload:
            obj = model
            for part in k.split("."):
                obj = getattr(obj, part)
            payload["grad/" + k] = obj.grad.detach().numpy()
            ngrad += 1
    bn_after = (model.hash_bn.running_mean.detach().numpy().copy(), model.hash_bn.running_var.detach().numpy().copy())
    # ---- the same step with the attention-diversity term on (loss_scales.attn_div_loss = 1, div_method 1 as shipped): the loss reads
    # attn_cache[-1][:, :, -Q:, 1:-Q] and its gradient enters the last layer's attention.  Weight 25: the term itself is ~0.9 and its
    # gradient small next to the three cross-entropies.
    model.zero_grad()
    crit2 = LGHLoss(margin=0.2, scale=8, loss_scales=dict(logits=0, hash_logits=0, bin_logits=1, cont_logits=1, l2=0, attn_div_loss=25,
                                                          concept_logits=1), avg_before_softmax=False, lmbd=0.5, div_method=1, ncontext=4)
    rm, rv = model.hash_bn.running_mean.clone(), model.hash_bn.running_var.clone()
    feats2, out2 = model(x)
    loss2 = crit2(out2, labels)
    loss2.backward()
    payload["attn/loss"] = loss2.detach().numpy()
    payload["attn/loss_attn_div"] = crit2.losses["attn_div"].detach().numpy()
    payload["attn/concept_attention"] = out2["attn_cache"][-1][:, :, -4:, 1:-4].detach().numpy()
    for k, p in named.items():
        if k.startswith(("adapter_params.", "trainable_params.")) or p.grad is None:
            continue
        if ".adapt_mlp_" in k or k.startswith("hash_attention.") :
            payload["attngrad/" + k] = p.grad.detach().numpy()
    payload["attngrad/hash_queries"] = model.hash_queries.grad.detach().numpy()
    with torch.no_grad():
        model.hash_bn.running_mean.copy_(rm)
        model.hash_bn.running_var.copy_(rv)
    # ---- and with `avg_attn: True` (models/loss/coop.py:164-167): the loss averages EVERY layer's attention map before slicing out the
    # concept tokens' rows, so its gradient enters the attention of every layer.  `attnavg/concept_attention_layers` =
    # torch.stack(attn_cache)[:, :, :, -Q:, 1:-Q]  (L, B, heads, Q, Np)
    model.zero_grad()
    crit3 = LGHLoss(margin=0.2, scale=8, loss_scales=dict(logits=0, hash_logits=0, bin_logits=1, cont_logits=1, l2=0, attn_div_loss=25,
                                                          concept_logits=1), avg_before_softmax=False, lmbd=0.5, div_method=1, ncontext=4,
                    avg_attn=True)
    feats3, out3 = model(x)
    loss3 = crit3(out3, labels)
    loss3.backward()
    payload["attnavg/loss"] = loss3.detach().numpy()
    payload["attnavg/loss_attn_div"] = crit3.losses["attn_div"].detach().numpy()
    payload["attnavg/concept_attention_layers"] = torch.stack(out3["attn_cache"], dim=0)[:, :, :, -4:, 1:-4].detach().numpy()
    for k, p in named.items():
        if k.startswith(("adapter_params.", "trainable_params.")) or p.grad is None:
            continue
        if ".adapt_mlp_" in k or k.startswith("hash_attention."):
            payload["attnavggrad/" + k] = p.grad.detach().numpy()
    payload["attnavggrad/hash_queries"] = model.hash_queries.grad.detach().numpy()
    with torch.no_grad():
        model.hash_bn.running_mean.copy_(rm)
        model.hash_bn.running_var.copy_(rv)
    # BatchNorm running statistics after the step (momentum 0.1)
    payload["out/bn_running_mean"] = bn_after[0]
    payload["out/bn_running_var"] = bn_after[1]
    path = os.path.join(GOLDEN, "train_tiny.npz")
    np.savez_compressed(path, **payload)
    print("train_tiny ->", path, f"{os.path.getsize(path) / 1e6:.2f} MB, loss {float(loss):.6f}, {ngrad} gradient tensors")
    print({k: float(v) for k, v in crit.losses.items()})


if __name__ == "__main__":
    main()
