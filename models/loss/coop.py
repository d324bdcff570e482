"""`models.loss.coop.LGHLoss`: the training objective (under autograd) and the evaluation meters (callers wrap in no_grad).

The reference's trainer calls the criterion during inference to fill the loss/accuracy meters
(trainers/coop.py:80-88 -> models/loss/coop.py:120-189).  This restates the terms the shipped config enables
(`bin_logits`, `cont_logits`, `concept_logits`: margin-cosine cross-entropy with `scale`/`margin`; plus the `quan`
diagnostic) on the small (B, C) logits the HIP head produces.  It is a few KB per batch; plain torch ops, differentiable (the training step, trainers/coop.py train_one_batch, backpropagates
through it into the HIP encoder's backward).  `hash_logits` (mixture of
softmaxes) is included for completeness; `attn_div_loss` reads the concept tokens' attention rows -- the last layer's
(`outputs['concept_attention']`) or, with `avg_attn`, every layer's (`outputs['concept_attention_layers']`) -- tapped from the fused
attention kernel and differentiable through `ch_train_backward`.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class LGHLoss(nn.Module):
    def __init__(self, scale=1, margin=0, loss_scales=None, avg_before_softmax=False, lmbd=0.5, ncontext=8,
                 exponential_scale=0, div_method=0, concept_cossim=True, **kwargs):
        super().__init__()
        self.scale, self.margin, self.lmbd = scale, margin, lmbd
        self.loss_scales = dict(loss_scales) if loss_scales is not None else {
            "logits": 1, "hash_logits": 1, "bin_logits": 1, "cont_logits": 1, "concept_logits": 0, "attn_div_loss": 0}
        if kwargs.get("nregs"):
            raise NotImplementedError("nregs != 0 (register tokens) is not built: the model has none")
        self.avg_attn = bool(kwargs.get("avg_attn", False))   # reference :164-167: the layer mean of the attention maps instead of the last
        self.div_method, self.div_min = div_method, float(kwargs.get("div_min", 0))
        if exponential_scale:
            raise NotImplementedError("exponential_scale != 0 is not built")
        self.avg_before_softmax, self.ncontext, self.concept_cossim = avg_before_softmax, ncontext, concept_cossim
        self.losses = {}

    def _margin_logits(self, logits, labels):
        """cosine-margin logits: scale * (logits - margin * onehot)  (reference :46-66)"""
        if labels.dim() == 2:       # soft / multi-hot labels
            onehot = labels.to(logits.dtype)
            target = onehot / onehot.sum(-1, keepdim=True)
        else:
            onehot = F.one_hot(labels, logits.shape[-1]).to(logits.dtype)
            target = labels
        if logits.dim() == 3:
            if labels.dim() == 2:
                onehot = onehot.unsqueeze(0)
            else:
                # reference :55-57: `y_onehot.scatter_(-1, labels[None, :, None], margin)` -- the index has size 1 along the concept
                # axis, so only concept 0 receives the margin.  Kept as the reference computes it (pinned by tests/golden/train_tiny.npz).
                first = torch.zeros(logits.shape[0], 1, 1, dtype=logits.dtype, device=logits.device)
                first[0] = 1
                onehot = onehot.unsqueeze(0) * first
        return self.scale * (logits - self.margin * onehot), target

    def _ce(self, logits, labels, cossim=True):
        if cossim:
            logits, labels = self._margin_logits(logits, labels)
        elif labels.dim() == 2:
            labels = labels / labels.sum(-1, keepdim=True)
        if logits.dim() == 3:       # (Q, B, C): every concept classifies the same labels (reference :74-85)
            Q = logits.shape[0]
            flat = logits.reshape(Q * logits.shape[1], -1)
            rep = labels.unsqueeze(0).expand(Q, *labels.shape).reshape(-1, *labels.shape[1:])
            return F.cross_entropy(flat, rep)
        return F.cross_entropy(logits, labels)

    def _hash_loss(self, l1, l2, labels):
        if self.avg_before_softmax:
            return self._ce(self.lmbd * l1 + (1 - self.lmbd) * l2, labels)
        l1, _ = self._margin_logits(l1, labels)
        l2, target = self._margin_logits(l2, labels)
        if target.dim() == 1:
            target = F.one_hot(target, l1.shape[-1]).to(l1.dtype)
        prob = self.lmbd * torch.softmax(l1, -1) + (1 - self.lmbd) * torch.softmax(l2, -1)
        return -(target * torch.log(prob.clamp(min=1e-7))).sum(-1).mean()

    def forward(self, outputs, labels):
        codes = outputs["codes"]
        with torch.no_grad():
            self.losses["quan"] = 1 - F.cosine_similarity(codes, codes.sign(), dim=-1).mean()
        total = torch.zeros((), device=codes.device)
        ls = self.loss_scales
        if ls.get("logits", 0) and "logits" in outputs:
            self.losses["aux"] = self._ce(outputs["logits"], labels)
            total = total + ls["logits"] * self.losses["aux"]
        if ls.get("concept_logits", 0):
            self.losses["concept"] = self._ce(outputs["logits_concept"], labels, cossim=self.concept_cossim)
            total = total + ls["concept_logits"] * self.losses["concept"]
        if ls.get("hash_logits", 0):
            self.losses["hash"] = self._hash_loss(outputs["logits_cont"], outputs["logits_bin"], labels)
            total = total + ls["hash_logits"] * self.losses["hash"]
        if ls.get("cont_logits", 0):
            self.losses["cont"] = self._ce(outputs["logits_cont"], labels)
            total = total + ls["cont_logits"] * self.losses["cont"]
        if ls.get("bin_logits", 0):
            self.losses["bin"] = self._ce(outputs["logits_bin"], labels)
            total = total + ls["bin_logits"] * self.losses["bin"]
        if ls.get("attn_div_loss", 0):
            # reference :161-187 on attn_cache[-1][:, :, -Q:, 1:-Q], which this path hands over directly as
            # outputs["concept_attention"] (B, heads, Q, Np): head mean, l2 over the patches, pairwise cosine between the Q concept
            # tokens, (div_method 0: relu(cos - div_min)), batch mean, mean of the strict upper triangle
            key = "concept_attention_layers" if self.avg_attn else "concept_attention"
            if outputs.get(key) is None:
                want = '"all"' if self.avg_attn else "True"
                raise RuntimeError(f"attn_div_loss needs outputs['{key}']: set model.return_concept_attention = {want} "
                                   f"(COOPTrainer does it when the term is enabled)")
            # avg_attn: torch.stack(attn_cache).mean(0) sliced to the concept rows == the layer mean of every layer's concept rows
            att = outputs[key].mean(dim=0) if self.avg_attn else outputs[key]
            a = F.normalize(att.mean(dim=1), dim=-1, p=2)
            # the Q x Q Gram matrix of 4 concept tokens, element-wise: a batched matmul would initialise the BLAS library for it (0.18 s in an
            # evaluation command, where nothing else on the torch side multiplies matrices)
            cos = (a.unsqueeze(2) * a.unsqueeze(1)).sum(-1)
            if self.div_method == 0:
                cos = (cos - self.div_min).relu()
            cos = cos.mean(dim=0)
            tri = torch.triu(torch.ones_like(cos, dtype=torch.bool), 1)
            self.losses["attn_div"] = cos[tri].mean()
            total = total + ls["attn_div_loss"] * self.losses["attn_div"]
        return total
