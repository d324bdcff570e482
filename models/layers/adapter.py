"""Parameter holders for the AdaptFormer-style adapters the reference splices into every CLIP encoder layer
(models/layers/adapter.py:11-60 `Adapter`, :398-416 `clip_add_adapter_`).  Arithmetic is in csrc/ (LayerNorm kernel +
two fused-epilogue GEMMs); these classes only reproduce the state_dict key layout and the reference initialisation."""
from __future__ import annotations

import math

import torch
import torch.nn as nn


class Adapter(nn.Module):
    def __init__(self, in_dim: int, bottleneck_dim: int):
        super().__init__()
        self.scale = nn.Parameter(torch.ones(1))                 # "learnable_scalar"
        self.adapter_layer_norm = nn.LayerNorm(in_dim)           # layernorm option "in"
        self.down_proj = nn.Linear(in_dim, bottleneck_dim)
        self.up_proj = nn.Linear(bottleneck_dim, in_dim)
        with torch.no_grad():                                    # same init family as the reference (:39-44)
            nn.init.kaiming_uniform_(self.down_proj.weight, a=math.sqrt(5))
            nn.init.zeros_(self.up_proj.weight)
            nn.init.zeros_(self.down_proj.bias)
            nn.init.zeros_(self.up_proj.bias)


def clip_add_adapter_(vision_model, bottleneck_dim: int, trainable_params: nn.ParameterDict = None,
                      adapt_mlp_1: bool = True, adapt_mlp_2: bool = True):
    """Adds `adapt_mlp_{1,2}` to every encoder layer and registers the alias entries `adapter_<i>_<name>` exactly as the
    reference does (these aliases are why a reference state_dict lists every adapter tensor twice)."""
    if trainable_params is None:
        trainable_params = nn.ParameterDict()
    d = vision_model.config.hidden_size
    for i, layer in enumerate(vision_model.encoder.layers):
        mods = {}
        if adapt_mlp_1:
            layer.adapt_mlp_1 = Adapter(d, bottleneck_dim)
            mods["adapt_mlp_1"] = layer.adapt_mlp_1
        if adapt_mlp_2:
            layer.adapt_mlp_2 = Adapter(d, bottleneck_dim)
            mods["adapt_mlp_2"] = layer.adapt_mlp_2
        for pname, param in nn.ModuleDict(mods).named_parameters():
            trainable_params[f"adapter_{i}_{pname.replace('.', '_')}"] = param
    return trainable_params
