"""`models.arch.coop.LGHWithFixedPrompt` -- the ConceptHash model under the reference's dotted name
(reference: models/arch/coop.py:180-625), evaluated by the MI355X HIP library.

What is kept: constructor arguments as the Hydra config passes them
(configs/model/concept_hash_final_v1_nosa_apt.yaml), `state_dict()` / `load_state_dict()` in the reference key layout
(including the `adapter_params.*` / `trainable_params.*` aliases), `forward(x) ->
(image_features, dict)` with the reference's output keys, `get_backbone / get_training_modules / get_adapter /
get_center`.  What is different: in eval mode the torch modules are parameter holders only -- `forward` hands the parameters to
`ch_model_create` once (re-done when they change) and every batch to `ch_encode`.  In training mode (reference
trainers/coop.py:107-131) the encoder forward + backward run in the HIP library (`ch_train_forward` / `ch_train_backward`, the
adapters' gradients included) and the three tiny pieces around it -- the 4-token concept generator, the hashing head on
(B, Q, D) and the loss -- run on torch autograd, joined by `concepthash_amd.training.EncoderFunction`.
"""
from __future__ import annotations

import logging
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from concepthash_amd.encoder import ConceptHashEncoder
from models.layers.adapter import clip_add_adapter_


class CosSim(nn.Module):
    """Parameter holder for the concept classifier (reference models/layers/cossim.py; `centroids` (C, D))."""

    def __init__(self, nfeat, nclass):
        super().__init__()
        self.centroids = nn.Parameter(torch.randn(nclass, nfeat))


def _cfg_get(cfg, key, default=None):
    if cfg is None:
        return default
    return cfg.get(key, default) if hasattr(cfg, "get") else getattr(cfg, key, default)


class LGHWithoutText(nn.Module):
    def __init__(self, backbone, nbit: int, nclass: int, ncontext: int, add_bn=False, use_before_projection: bool = True,
                 upt_config=None, fixed_center: Optional[torch.Tensor] = None, additional_blocks: int = 0,
                 concept_reg: bool = False, concept_cossim: bool = True, has_adapter: bool = False,
                 adapter_bottleneck_dim: int = 384, adapter_mlp_1: bool = True, adapter_mlp_2: bool = True,
                 max_batch: int = 256, **kwargs):
        super().__init__()
        unsupported = {k: v for k, v in dict(additional_blocks=additional_blocks, vpt_pe=kwargs.get("vpt_pe", False),
                                             fixed_pe=kwargs.get("fixed_pe", False), nregs=kwargs.get("nregs", 0),
                                             attention_adapter=kwargs.get("attention_adapter", False),
                                             concept_share_pe=kwargs.get("concept_share_pe", False)).items() if v}
        if kwargs.get("hash_fc_nlayers", 1) != 1:
            unsupported["hash_fc_nlayers"] = kwargs["hash_fc_nlayers"]
        if unsupported:
            raise NotImplementedError(f"options outside the shipped ConceptHash config are not built: {unsupported}")
        if not _cfg_get(upt_config, "multi", False) or not _cfg_get(upt_config, "single_hash_fc", False) \
                or _cfg_get(upt_config, "ensemble_method", "concat") != "concat" or not _cfg_get(upt_config, "upt_context", True) \
                or _cfg_get(upt_config, "v2") or _cfg_get(upt_config, "exclude_cls"):
            raise NotImplementedError("only upt_config {multi: True, single_hash_fc: True, ensemble_method: concat} "
                                      "(the shipped concept_hash_final_v1_nosa_apt config) is built")
        if not use_before_projection or add_bn == "dbn" or not concept_cossim:
            raise NotImplementedError("use_before_projection=False / add_bn='dbn' / concept_cossim=False are not built")
        if nbit % ncontext:
            raise ValueError("nbit must be divisible by ncontext")

        self.nbit, self.nclass, self.ncontext = nbit, nclass, ncontext
        self.add_bn, self.use_before_projection, self.upt_config = add_bn, use_before_projection, upt_config
        self.concept_reg, self.has_adapter, self.multi = concept_reg, has_adapter, True
        self.max_batch = max_batch
        self.trainable_params = nn.ParameterDict()
        self.adapter_params = nn.ParameterDict()
        if has_adapter:   # the reference installs adapters on backbone.model before rebinding self.backbone (arch/base.py:29-44)
            clip_add_adapter_(backbone.model.vision_model, adapter_bottleneck_dim, self.adapter_params,
                              adapt_mlp_1=adapter_mlp_1, adapt_mlp_2=adapter_mlp_2)
        self.features_size = backbone.features_size
        self.backbone = backbone.model
        vcfg = self.backbone.vision_model.config
        self.vision_dim, self.embed_dim = vcfg.hidden_size, vcfg.projection_dim
        self._heads, self._act = vcfg.num_attention_heads, vcfg.hidden_act
        D, P, Q = self.vision_dim, self.embed_dim, ncontext

        # ---- hash_initialization (reference :278-395), the single_hash_fc / concat branch
        if _cfg_get(upt_config, "hash_pe", False):
            self.hash_pe = nn.Parameter(torch.randn(1, Q, D))
            self.trainable_params["hash_pe"] = self.hash_pe
        else:
            self.register_buffer("hash_pe", torch.zeros(1, Q, D))
        self.hash_fc = nn.Linear(D, nbit // Q, bias=False)
        self.hash_bn = nn.BatchNorm1d(nbit) if add_bn else nn.Identity()
        self.hash_queries = nn.Parameter(torch.randn(1, Q, P))
        self.trainable_params["hash_queries"] = self.hash_queries
        ha = nn.Module()
        self._upt_heads = int(_cfg_get(upt_config, "num_heads", 8))
        drop = float(_cfg_get(upt_config, "dropout", 0.0))
        ha.sa = nn.MultiheadAttention(P, self._upt_heads, batch_first=True, dropout=drop)
        ha.ffn = nn.Sequential(nn.Linear(P, P), nn.ReLU(), nn.Dropout(drop), nn.Linear(P, P))
        ha.norm1, ha.norm2 = nn.LayerNorm(P), nn.LayerNorm(P)
        ha.ffn2 = nn.Linear(P, D)
        self.hash_attention = ha
        if fixed_center is not None:
            self.register_buffer("center", torch.as_tensor(fixed_center, dtype=torch.float32))
        else:
            self.center = nn.Parameter(torch.randn(nclass, nbit) * 0.02)
            self.trainable_params["center"] = self.center
        if concept_reg:   # concept_initialization (reference :251-267)
            self.concept_pe = nn.Parameter(torch.randn(1, Q, D) * 0.02)
            self.trainable_params["concept_pe"] = self.concept_pe
            self.concept_ce = CosSim(D, nclass)
            self.trainable_params["concept_ce_centroids"] = self.concept_ce.centroids
        # False | True (the last layer's concept-token attention rows) | "all" (every layer's)
        rca = kwargs.get("return_concept_attention", False)
        self.return_concept_attention = "all" if rca == "all" else bool(rca)
        # reference forward always returns every layer's hidden state (`image_hidden_states`, coop.py:474-486, 582-598); nothing on
        # the encode-and-retrieve path reads them, so they are produced on request only -- by the per-layer parity tap
        # ch_encode_hidden, one partial run of the encoder per layer (L + 1 runs: for inspection, not for throughput)
        self.return_hidden_states = bool(kwargs.get("return_hidden_states", False))
        self._engine: Optional[ConceptHashEncoder] = None
        self._engine_key = None
        self.eval()

    # ---- reference accessors -------------------------------------------------------------------------------------
    def get_center(self):
        return self.center

    def get_backbone(self):
        return self.backbone.vision_model

    def get_adapter(self):
        return self.adapter_params

    def get_training_modules(self):
        return nn.ModuleDict({"trainable_params": self.trainable_params, "hash_fc": self.hash_fc, "hash_bn": self.hash_bn,
                              "hash_attention": self.hash_attention})

    def count_parameters(self, mode="trainable"):
        ps = list(self.parameters())
        if mode == "trainable":
            return sum(p.numel() for p in ps if p.requires_grad)
        if mode == "non-trainable":
            return sum(p.numel() for p in ps if not p.requires_grad)
        return sum(p.numel() for p in ps)

    # ---- checkpoint layout ---------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """Reference checkpoints load strictly; a checkpoint that lacks only the alias entries or the text-side leftovers
        (`backbone.text_projection.weight`, `backbone.logit_scale`) is accepted too."""
        own = super().state_dict()
        tolerated = ("adapter_params.", "trainable_params.", "backbone.text_projection", "backbone.logit_scale",
                     "backbone.vision_model.embeddings.position_ids", "hash_bn.num_batches_tracked")
        missing = [k for k in own if k not in state_dict and not k.startswith(tolerated)]
        unexpected = [k for k in state_dict if k not in own and not k.startswith(tolerated)]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]}{'...' if len(missing) > 5 else ''}, "
                               f"unexpected {unexpected[:5]}{'...' if len(unexpected) > 5 else ''}")
        res = super().load_state_dict({k: v for k, v in state_dict.items() if k in own}, strict=False, **kw)
        self._engine_key = None
        self._drop_train_engine()
        return res

    def _drop_train_engine(self):
        eng = getattr(self, "_train_engine", None)
        if eng is not None:
            if eng.momentum_buf is not None:      # the fused arena step's momentum outlives the engine (rebuilds: a larger batch,
                self._carried_momentum = eng.momentum_buf   # .to(), load_state_dict), as torch's optimizer state does
            eng.close()
        self._train_engine = None

    def _apply(self, fn, *a, **k):
        self._drop_train_engine()          # the adapters' parameters are views of its arena: release before they are replaced
        out = super()._apply(fn, *a, **k)
        self._engine_key = None
        return out

    # ---- training mode (reference forward, models/arch/coop.py:524-598, with self.training = True) -----------------------
    def forward_hash_query(self):
        """The concept-token generator (reference :413-427, the non-"v2" residual form norm(x) + f(x)) on torch autograd."""
        ha = self.hash_attention
        x = self.hash_queries
        x = ha.norm1(x) + ha.sa(x, x, x)[0]
        x = ha.norm2(x) + ha.ffn(x)
        return ha.ffn2(x)                                       # (1, Q, D)

    def forward_concept(self, concept_features):
        """reference :269-276 with CosSim (models/layers/cossim.py:37-82, group = 1): (B, Q, D) -> (Q, B, C)"""
        B, Q, D = concept_features.shape
        x = (concept_features + self.concept_pe).reshape(B * Q, D)
        logits = F.normalize(x, p=2, dim=-1) @ F.normalize(self.concept_ce.centroids, p=2, dim=-1).t()
        return logits.reshape(B, Q, -1).transpose(0, 1)

    def _ensure_train_engine(self, device, image_size, batch):
        from concepthash_amd.training import TrainEngine, adapter_modules
        if not self.has_adapter:
            raise NotImplementedError("training on the MI355X path trains the adapters + head (has_adapter=True, the shipped config)")
        key = (str(device), image_size)
        eng = getattr(self, "_train_engine", None)
        if eng is not None and (self._train_engine_key != key or eng.max_batch < batch):
            self._drop_train_engine()
            eng = None
        if eng is None:
            eng = TrainEngine(self._projected_state_dict(), adapter_modules(self.backbone.vision_model), heads=self._heads,
                              upt_heads=self._upt_heads, act=self._act, max_batch=max(batch, getattr(self, "train_max_batch", 0)),
                              device=device, image_size=image_size)
            carried = getattr(self, "_carried_momentum", None)
            if carried is not None and carried.numel() == eng.params.numel():
                eng.momentum_buf = carried.to(eng.device)
            self._carried_momentum = None
            self._train_engine, self._train_engine_key = eng, key
        return eng

    def _forward_train(self, x):
        from concepthash_amd.training import EncoderFunction
        eng = self._ensure_train_engine(x.device, int(x.shape[-1]), int(x.shape[0]))
        eng.sync_versions()                                      # an optimizer step since the last forward -> re-derive working copies
        ctx = self.forward_hash_query()
        concept_attention = None
        concept_attention_layers = None
        if self.return_concept_attention == "all":   # every layer's rows (the `avg_attn` form of the attention-diversity term)
            hash_features, concept_attention_layers = EncoderFunction.apply(ctx, x, eng, eng.anchor, "all")
            concept_attention = concept_attention_layers[-1]
        elif self.return_concept_attention:      # the loss's attention-diversity term reads it (set by the trainer when that term is on)
            hash_features, concept_attention = EncoderFunction.apply(ctx, x, eng, eng.anchor, True)
        else:
            hash_features = EncoderFunction.apply(ctx, x, eng, eng.anchor)   # (B, Q, D): HIP forward, HIP backward
        B = x.shape[0]
        vision_hash = self.hash_fc(hash_features + self.hash_pe).reshape(B, -1)      # reference :544-553
        vision_hash = self.hash_bn(vision_hash)                                      # BatchNorm1d on the batch statistics
        center = self.get_center()
        v_l2 = F.normalize(vision_hash, dim=-1, p=2)
        c_l2 = F.normalize(center, dim=-1, p=2)
        outputs = {"logits_cont": v_l2 @ c_l2.t(), "logits_bin": v_l2 @ (c_l2.sign() / (self.nbit ** 0.5)).t(),
                   "codes": vision_hash, "image_hidden_states": (), "hash_features": hash_features, "attn_cache": None}
        if self.concept_reg:
            outputs["logits_concept"] = self.forward_concept(hash_features)
        if concept_attention is not None:
            # = attn_cache[-1][:, :, -Q:, 1:-Q] of the reference (coop.py:481-482), (B, heads, Q, Np), differentiable; the full
            # per-layer (B, heads, N, N) maps are never materialised
            outputs["concept_attention"] = concept_attention
        if concept_attention_layers is not None:     # (L, B, heads, Q, Np) = torch.stack(attn_cache)[:, :, :, -Q:, 1:-Q]
            outputs["concept_attention_layers"] = concept_attention_layers
        # image_features (pooled CLS -> post-LN -> projection, reference :498-501) feeds no term of the shipped loss
        # (loss_scales.logits = 0): not computed in training mode
        return None, outputs

    # ---- engine --------------------------------------------------------------------------------------------------
    def _projected_state_dict(self):
        return super().state_dict()

    def _ensure_engine(self, device, image_size=None) -> ConceptHashEncoder:
        """One engine per (device, weights version, input resolution): a resolution other than the pretrain one gets the
        interpolated position table (reference interpolate_pos_encoding, coop.py:429-450), folded when the engine is built."""
        key = (str(device), image_size, tuple(p._version for p in self.parameters()), tuple(b._version for b in self.buffers()))
        if self._engine is None or self._engine_key != key:
            if self._engine is not None:
                self._engine.close()
            self._engine = ConceptHashEncoder(self._projected_state_dict(), heads=self._heads, upt_heads=self._upt_heads,
                                              act=self._act, max_batch=self.max_batch, device=device, image_size=image_size)
            self._engine_key = key
        return self._engine

    def forward(self, x, y=None, cache=False, update_cache=False):
        if not x.is_cuda:
            raise RuntimeError("LGHWithFixedPrompt.forward needs a GPU tensor; there is no CPU fallback")
        if x.dim() != 4 or x.shape[-1] != x.shape[-2]:
            raise ValueError("LGHWithFixedPrompt.forward expects square [B, 3, S, S] inputs")
        if self.training and torch.is_grad_enabled():
            return self._forward_train(x)
        eng = self._ensure_engine(x.device, int(x.shape[-1]))
        want = ["codes", "logits_cont", "logits_bin", "hash_features"]
        if self.return_concept_attention:
            want.append("concept_attn")
        if self.return_concept_attention == "all":
            want.append("concept_attn_layers")
        if self.concept_reg:
            want.append("logits_concept")
        if eng.has_pooled:
            want.append("image_features")
        out = eng.encode(x, want=want)
        outputs = {"logits_cont": out["logits_cont"], "logits_bin": out["logits_bin"], "codes": out["codes"],
                   # the reference returns every layer's hidden state / attention map; the fused path does not
                   # materialise them (retrieval never reads them)
                   "image_hidden_states": (), "hash_features": out["hash_features"], "attn_cache": None}
        if self.return_hidden_states:     # (L + 1) x (B, N, D) fp32: after pre-LN, then after every encoder layer
            outputs["image_hidden_states"] = tuple(eng.hidden_states(x, layer) for layer in range(eng.cfg["layers"] + 1))
        if self.return_concept_attention:
            # what the reference's consumers slice out of attn_cache[-1]: [:, :, -Q:, 1:-Q]  (B, heads, Q, Np)
            outputs["concept_attention"] = out["concept_attn"]
        if self.return_concept_attention == "all":   # every layer's: torch.stack(attn_cache)[:, :, :, -Q:, 1:-Q]  (L, B, heads, Q, Np)
            outputs["concept_attention_layers"] = out["concept_attn_layers"]
        if self.concept_reg:
            outputs["logits_concept"] = out["logits_concept"]
        return out.get("image_features"), outputs


class LGHWithFixedPrompt(LGHWithoutText):
    def __init__(self, backbone, nbit: int, nclass: int, ncontext: int, add_bn=False, use_before_projection: bool = True,
                 upt_config=None, fixed_center: Optional[torch.Tensor] = None, additional_blocks: int = 0,
                 text_projection: Optional[nn.Module] = None, **kwargs):
        if fixed_center is None:
            raise ValueError("LGHWithFixedPrompt needs `fixed_center` (C, 512); it is checkpoint data on this path "
                             "(building it needs the CLIP text tower, trainers/orthohash.py:94-260)")
        super().__init__(backbone, nbit, nclass, ncontext, add_bn, use_before_projection, upt_config, fixed_center,
                         additional_blocks, **kwargs)
        cd = int(self.center.shape[1])
        self.text_projection = text_projection if text_projection is not None else nn.Linear(cd, nbit)
        keys = set(self.text_projection.state_dict().keys())
        if keys not in ({"weight", "bias"}, {"0.weight", "0.bias", "2.weight", "2.bias"}):
            raise NotImplementedError("text_projection must be Linear or Sequential(Linear, ReLU, Linear)")

    def get_training_modules(self):
        m = super().get_training_modules()
        m["text_projection"] = self.text_projection
        return m

    def get_center(self):
        """text_projection(center) (reference :624-625) -- small, input independent; evaluated with torch on the
        parameters' device for callers that ask for it (the encode path folds it inside ch_model_create; the training forward
        differentiates through it)."""
        return self.text_projection(self.center)
