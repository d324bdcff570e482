"""CPU oracle for the ConceptHash TRAINING step -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/`` and ``__graft_entry__.smoke()`` may import this module.  fp32 torch restatement (autograd on a flat
reference-layout ``state_dict``) of one optimisation step's forward + loss + backward as the reference runs it
(paths relative to /root/reference):

  train_one_batch ....... trainers/coop.py:107-131 (zero_grad, forward, criterion, loss.backward(), optimizer.step())
  what is trainable ..... trainers/base.py:133-152 with `backbone_lr_scale: 0`, `has_adapter: True`
                          (configs/model/concept_hash_final_v1_nosa_apt.yaml): get_adapter() + get_training_modules()
                          (models/arch/coop.py:613-622) = adapters, hash_queries / hash_pe / concept_pe / concept_ce.centroids,
                          hash_fc, hash_bn, hash_attention, text_projection; everything else frozen
  forward, train mode ... models/arch/coop.py:524-598; BatchNorm1d uses the batch statistics (biased variance), dropout
                          (upt_config.dropout, adapter_dropout) is taken as 0 here: the fixtures are generated with it off
  loss .................. models/loss/coop.py:120-189 with the shipped settings (scale 8, margin 0.2; terms concept_logits,
                          cont_logits, bin_logits at weight 1): cosine-margin cross-entropy, :46-66 + :68-90
  SGD ................... configs/optim/sgd.yaml (momentum 0.9, weight_decay 5e-4, nesterov False), torch.optim.SGD semantics

PARITY PINNING: ``tests/golden/train_tiny.npz`` holds the loss terms and every trainable parameter's gradient produced by the
reference's own (unmodified) model class and its own LGHLoss, run by ``oracle/gen_train_golden.py`` in the build container;
``tests/test_oracle_golden.py`` checks this restatement against it.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from oracle import encoder_oracle as eo

VM = eo.VM


def trainable_keys(sd) -> list:
    keys = []
    for k in sd:
        if ".adapt_mlp_" in k and k.startswith(VM):
            keys.append(k)
        elif k in ("hash_queries", "hash_pe", "concept_pe", "concept_ce.centroids", "hash_fc.weight", "hash_bn.weight", "hash_bn.bias"):
            keys.append(k)
        elif k.startswith("hash_attention.") or k.startswith("text_projection."):
            keys.append(k)
    return sorted(keys)


def margin_ce(logits, labels, scale, margin):
    """compute_ce_loss with cossim=True (models/loss/coop.py:46-90): scale * (logits - margin * onehot), cross-entropy; a (Q,B,C)
    input is flattened to (Q*B, C) with the labels repeated and averaged over all Q*B rows (exponential_scale = 0).
    For (Q,B,C) logits with index labels the reference builds the margin with `y_onehot.scatter_(-1, labels[None, :, None],
    margin)` (:55-57): the index has size 1 along the concept axis, so ONLY concept 0 gets the margin.  Restated as is."""
    onehot = F.one_hot(labels, logits.shape[-1]).to(logits.dtype)
    if logits.dim() == 3:
        m = torch.zeros_like(logits)
        m[0] = margin * onehot
        z = scale * (logits - m)
        return F.cross_entropy(z.reshape(-1, z.shape[-1]), labels.repeat(logits.shape[0]))
    return F.cross_entropy(scale * (logits - margin * onehot), labels)


def forward_train(sd: Dict[str, torch.Tensor], images, heads, upt_heads=8, act="quick_gelu", bn_eps=1e-5, ctx=None) -> dict:
    """ctx: optional (1, Q, D) concept tokens to use instead of forward_hash_query() -- a leaf for tests of d(concept tokens)"""
    dims = eo.infer_dims(sd)
    Q = dims["Q"]
    R = eo._R(False)
    if ctx is None:
        ctx = eo.concept_tokens(sd, upt_heads)
    x = eo.embeddings(sd, images, R)
    x = torch.cat([x, ctx.expand(x.shape[0], -1, -1)], dim=1)
    h = eo.layer_norm(x, sd[VM + "pre_layrnorm.weight"].float(), sd[VM + "pre_layrnorm.bias"].float())
    probs, rows = None, []
    for i in range(dims["L"]):
        h, probs = eo.encoder_layer(sd, i, h, heads, R, act, want_probs=True)
        rows.append(probs[:, :, -Q:, 1:-Q])
    hf = h[:, -Q:, :]
    B = hf.shape[0]
    v = ((hf + sd["hash_pe"].float()) @ sd["hash_fc.weight"].float().t()).reshape(B, -1)
    mean, var = v.mean(0), v.var(0, unbiased=False)                       # BatchNorm1d, training mode
    codes = (v - mean) / torch.sqrt(var + bn_eps) * sd["hash_bn.weight"].float() + sd["hash_bn.bias"].float()
    lc, lb = eo.center_logits(sd, codes)
    return dict(codes=codes, hash_features=hf, logits_cont=lc, logits_bin=lb, logits_concept=eo.concept_logits(sd, hf),
                concept_attention=probs[:, :, -Q:, 1:-Q],       # attn_cache[-1][:, :, -Q:, 1:-Q] (coop.py:481-482)
                concept_attention_layers=torch.stack(rows, dim=0),   # torch.stack(attn_cache)[:, :, :, -Q:, 1:-Q]  (L, B, heads, Q, Np)
                bn_batch_mean=mean.detach(), bn_batch_var_unbiased=v.var(0, unbiased=True).detach(), concept_tokens=ctx)


def attn_div(concept_attention, div_method=1, div_min=0.0):
    """attention-diversity term (models/loss/coop.py:161-187, avg_attn False, nregs 0): head mean of the concept tokens' last-layer
    attention rows over the patches, l2 over patches, pairwise cosine (Q, Q) per image, (div_method 0: relu(cos - div_min)), batch
    mean, mean of the strict upper triangle."""
    a = F.normalize(concept_attention.mean(dim=1), dim=-1, p=2)
    cos = a @ a.transpose(1, 2)
    if div_method == 0:
        cos = (cos - div_min).relu()
    cos = cos.mean(dim=0)
    return cos[torch.triu(torch.ones_like(cos, dtype=torch.bool), 1)].mean()


def train_step_grads(sd, images, labels, heads, upt_heads=8, act="quick_gelu", scale=8.0, margin=0.2, attn_div_scale=0.0,
                     div_method=1, avg_attn=False) -> dict:
    """loss terms + gradient of every trainable tensor for one batch (labels: int64 class indices)."""
    sd = {k: v.clone() for k, v in sd.items()}
    keys = trainable_keys(sd)
    for k in keys:
        sd[k] = sd[k].float().requires_grad_(True)
    out = forward_train(sd, images, heads, upt_heads, act)
    losses = dict(concept=margin_ce(out["logits_concept"], labels, scale, margin),
                  cont=margin_ce(out["logits_cont"], labels, scale, margin),
                  bin=margin_ce(out["logits_bin"], labels, scale, margin))
    total = losses["concept"] + losses["cont"] + losses["bin"]
    if attn_div_scale:
        # avg_attn (models/loss/coop.py:164-167): the layer mean of the maps, then the concept rows == the layer mean of the rows
        losses["attn_div"] = attn_div(out["concept_attention_layers"].mean(dim=0) if avg_attn else out["concept_attention"], div_method)
        total = total + attn_div_scale * losses["attn_div"]
    total.backward()
    return dict(loss=total.detach(), losses={k: v.detach() for k, v in losses.items()},
                grads={k: sd[k].grad.detach() for k in keys}, out={k: v.detach() for k, v in out.items()})


def sgd_step(param, grad, buf, lr, momentum=0.9, weight_decay=5e-4):
    """torch.optim.SGD (nesterov False, dampening 0): d = g + wd p; buf = d (first step) or momentum buf + d; p -= lr buf."""
    d = grad + weight_decay * param
    buf = d.clone() if buf is None else momentum * buf + d
    return param - lr * buf, buf
