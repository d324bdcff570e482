"""`models.backbone.clip.CLIP` -- same dotted name and duck type as the reference wrapper
(models/backbone/clip.py:107-134): an object with `.model` (a CLIP-structured parameter tree:
`.vision_model.{embeddings, pre_layrnorm, encoder.layers, post_layernorm, config}` + `.visual_projection`) and
`.features_size`.

Differences, by design: the reference loads `CLIPModel.from_pretrained(<hub name>)`, which needs the network.  Here
`name` is (a) a local directory in HF layout (`config.json` + `model.safetensors` / `pytorch_model.bin`), (b) a known
CLIP id whose dimensions are built in, or (c) a dict of dimensions.  For (b)/(c) without local weights the tree is
seeded-random and `allow_random_init=True` must be passed (a trained ConceptHash checkpoint loaded afterwards
overwrites every tensor anyway).  Only parameters live here; all arithmetic runs in the HIP library.
"""
from __future__ import annotations

import json
import logging
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

PRESETS = {
    "openai/clip-vit-base-patch32": dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                                         patch_size=32, image_size=224, projection_dim=512, hidden_act="quick_gelu"),
    "openai/clip-vit-base-patch16": dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                                         patch_size=16, image_size=224, projection_dim=512, hidden_act="quick_gelu"),
    "openai/clip-vit-large-patch14": dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                                          intermediate_size=4096, patch_size=14, image_size=224, projection_dim=768,
                                          hidden_act="quick_gelu"),
    "laion/CLIP-ViT-B-32-laion2B-s34B-b79K": dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                                                  intermediate_size=3072, patch_size=32, image_size=224, projection_dim=512,
                                                  hidden_act="gelu"),
    # build-defined: there is no CLIP ViT-S/16; BASELINE.json config 1 uses ViT-S dimensions in CLIP structure
    "synthetic/clip-vit-small-patch16": dict(hidden_size=384, num_hidden_layers=12, num_attention_heads=6,
                                             intermediate_size=1536, patch_size=16, image_size=224, projection_dim=512,
                                             hidden_act="quick_gelu"),
}


class _Attn(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = (nn.Linear(d, d) for _ in range(4))


class _MLP(nn.Module):
    def __init__(self, d, m):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(d, m), nn.Linear(m, d)


class EncoderLayer(nn.Module):
    """Parameter holder with the key layout of HF `CLIPEncoderLayer` (+ adapters added by `add_adapters`)."""

    def __init__(self, d, m, eps):
        super().__init__()
        self.self_attn = _Attn(d)
        self.layer_norm1 = nn.LayerNorm(d, eps=eps)
        self.mlp = _MLP(d, m)
        self.layer_norm2 = nn.LayerNorm(d, eps=eps)


class _Embeddings(nn.Module):
    def __init__(self, d, patch, image):
        super().__init__()
        self.patch_size = patch
        self.class_embedding = nn.Parameter(torch.randn(d) * 0.02)
        self.patch_embedding = nn.Conv2d(3, d, kernel_size=patch, stride=patch, bias=False)
        n = (image // patch) ** 2 + 1
        self.position_embedding = nn.Embedding(n, d)
        self.register_buffer("position_ids", torch.arange(n).unsqueeze(0), persistent=False)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.layers = nn.ModuleList([EncoderLayer(cfg.hidden_size, cfg.intermediate_size, cfg.layer_norm_eps)
                                     for _ in range(cfg.num_hidden_layers)])


class _VisionModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.embeddings = _Embeddings(cfg.hidden_size, cfg.patch_size, cfg.image_size)
        self.pre_layrnorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)   # (sic) HF spelling
        self.encoder = _Encoder(cfg)
        self.post_layernorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class CLIPModelShell(nn.Module):
    """The slice of `transformers.CLIPModel` that LGHWithoutText keeps after `del self.backbone.text_model`
    (models/arch/coop.py:245-246): vision tower, visual_projection, text_projection, logit_scale."""

    def __init__(self, dims: dict, text_dim: int = 512):
        super().__init__()
        cfg = SimpleNamespace(layer_norm_eps=1e-5, **dims)
        self.vision_config = cfg
        self.vision_model = _VisionModel(cfg)
        self.visual_projection = nn.Linear(cfg.hidden_size, cfg.projection_dim, bias=False)
        self.text_projection = nn.Linear(text_dim, cfg.projection_dim, bias=False)
        self.logit_scale = nn.Parameter(torch.tensor(2.6592))


def _load_local_weights(model: CLIPModelShell, path: str):
    st = os.path.join(path, "model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        sd = load_file(st)
    else:
        sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu")
    own = model.state_dict()
    picked = {k: v for k, v in sd.items() if k in own and tuple(v.shape) == tuple(own[k].shape)}
    missing = [k for k in own if k not in picked and not k.startswith("text_projection") and k != "logit_scale"]
    if missing:
        raise KeyError(f"{path}: {len(missing)} vision tensors missing or mis-shaped, e.g. {missing[:3]}")
    model.load_state_dict(picked, strict=False)


class CLIP(nn.Module):
    def __init__(self, name="openai/clip-vit-base-patch32", allow_random_init: bool = False, seed: int = 0, **kwargs):
        super().__init__()
        local = isinstance(name, str) and os.path.isdir(name)
        if isinstance(name, dict):
            dims = dict(name)
        elif local:
            cfg = json.load(open(os.path.join(name, "config.json")))
            v = cfg.get("vision_config", cfg)
            dims = {k: v[k] for k in ("hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size",
                                      "patch_size", "image_size")}
            dims["projection_dim"] = cfg.get("projection_dim", v.get("projection_dim", 512))
            dims["hidden_act"] = v.get("hidden_act", "quick_gelu")
        elif name in PRESETS:
            dims = dict(PRESETS[name])
        else:
            raise FileNotFoundError(
                f"CLIP backbone '{name}' is neither a local directory nor a known id ({sorted(PRESETS)}); "
                "hub downloads are not available offline")
        dims.setdefault("hidden_act", "quick_gelu")
        if dims["hidden_size"] != 64 * dims["num_attention_heads"]:
            raise ValueError("the MI355X path supports head_dim == 64 only")
        gen_state = torch.random.get_rng_state()
        torch.manual_seed(seed)
        self.model = CLIPModelShell(dims)
        torch.random.set_rng_state(gen_state)
        if local:
            _load_local_weights(self.model, name)
        elif not allow_random_init:
            raise FileNotFoundError(
                f"no local weights for '{name}': pass a local HF directory as `model.backbone.name`, or "
                "`allow_random_init=True` when a trained ConceptHash checkpoint will be loaded over it")
        else:
            logging.warning("CLIP backbone '%s': seeded random initialisation (no local weights)", name)
        self.name = name
        self.downscale = dims["patch_size"]
        self.features_size = dims["hidden_size"]

    def forward(self, image):
        raise NotImplementedError("the stand-alone CLIP pooled forward is outside the ConceptHash path; "
                                  "use models.arch.coop.LGHWithFixedPrompt")
