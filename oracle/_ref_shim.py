"""Test-time tooling (THIS CONTAINER ONLY): import the *unmodified* reference model code from
/root/reference so that golden vectors can be generated from it.

This is oracle/test infrastructure, not product code: nothing under ``concepthash_amd/`` may import it,
and nothing here travels to the GPU box in a usable form (``/root/reference`` does not exist there).

What the shims are, and are not (SURVEY.md §8c recipe):
  * ``omegaconf``, ``torchvision``, ``timm`` are absent from the image.  The reference imports them at
    module scope only for a type annotation (``DictConfig``) and for backbones that ConceptHash never
    instantiates (resnet50, timm ViT).  Empty stand-in modules let those *import statements* succeed;
    no arithmetic of the hot path runs through a stand-in.
  * transformers 5.x dropped a few names that the reference imports but the hot path never calls
    (``ViTOutput``, docstring decorators, 4-D mask helpers).  They are re-added as inert placeholders.
  * transformers 5.x ``CLIPEncoder.forward`` passes kwargs that the reference's 4.x-style
    ``CLIPEncoderLayerWithAdapter.forward(hidden, attention_mask, causal_attention_mask, output_attentions)``
    rejects, so the encoder *loop* (not the layers) is replaced by a 4.x-style loop that calls every
    reference layer with the signature it was written for.
All stock-block arithmetic (attention, MLP, LayerNorm, embeddings) therefore comes from the installed
``transformers`` 5.15.0 eager path + the reference's own layer/arch code.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
os.environ.setdefault("HF_HUB_OFFLINE", "1")
os.environ.setdefault("TRANSFORMERS_OFFLINE", "1")

REFERENCE_ROOT = "/root/reference"


class _AttrDict(dict):
    """Minimal stand-in for omegaconf.DictConfig: attribute access + .get()."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    __setattr__ = dict.__setitem__


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        raise RuntimeError("stand-in object called: this path is not part of the ConceptHash hot path")


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy


def install():
    import torch  # noqa: F401
    import transformers  # noqa: F401
    from transformers.models.vit import modeling_vit
    from transformers.models.clip import modeling_clip
    import typing

    if "omegaconf" not in sys.modules:
        oc = _StubModule("omegaconf")
        oc.DictConfig = _AttrDict
        oc.OmegaConf = _Dummy
        sys.modules["omegaconf"] = oc
    for name in ("torchvision", "torchvision.models", "torchvision.models.resnet", "torchvision.transforms",
                 "timm", "timm.models", "timm.models.vision_transformer", "timm.models.swin_transformer",
                 "timm.models.layers"):
        if name not in sys.modules:
            sys.modules[name] = _StubModule(name)

    if not hasattr(modeling_vit, "ViTOutput"):
        modeling_vit.ViTOutput = _Dummy
    ident_deco = lambda *a, **k: (lambda f: f)
    for name, val in dict(
            CLIPTextTransformer=getattr(modeling_clip, "CLIPTextTransformer", _Dummy),
            add_start_docstrings_to_model_forward=ident_deco,
            replace_return_docstrings=ident_deco,
            CLIP_TEXT_INPUTS_DOCSTRING="",
            _create_4d_causal_attention_mask=_Dummy,
            _prepare_4d_attention_mask=_Dummy,
            Optional=typing.Optional, Tuple=typing.Tuple, Union=typing.Union).items():
        if not hasattr(modeling_clip, name):
            setattr(modeling_clip, name, val)

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def attr_dict(**kw):
    return _AttrDict(**kw)


class FourXEncoderLoop:
    """4.x-style CLIPEncoder loop: calls each (reference) layer as
    layer(hidden, None, None, output_attentions=True) and collects hidden states + attentions."""

    def __init__(self, layers, config):
        self.layers = layers
        self.config = config

    def __call__(self, inputs_embeds, output_attentions=True, output_hidden_states=True, return_dict=True, **kw):
        h = inputs_embeds
        all_h, all_a = (), ()
        for layer in self.layers:
            all_h = all_h + (h,)
            out = layer(h, None, None, output_attentions=True)
            h = out[0]
            all_a = all_a + (out[1],)
        all_h = all_h + (h,)
        return _AttrDict(last_hidden_state=h, hidden_states=all_h, attentions=all_a)


def build_reference_model(vision_dims, nbit, nclass, ncontext=4, adapter_bottleneck_dim=384, seed=0,
                          center_dim=512, hidden_act="quick_gelu", upt_dropout=0.1):
    """Instantiate the reference's LGHWithFixedPrompt around a locally-built (random-init) CLIPModel.

    vision_dims: dict(hidden_size, intermediate_size, num_hidden_layers, num_attention_heads,
                      image_size, patch_size, projection_dim)
    """
    install()
    import torch
    import torch.nn as nn
    from transformers import CLIPConfig, CLIPModel
    import models.arch.coop as ref_coop  # unmodified reference source

    torch.manual_seed(seed)
    proj = vision_dims["projection_dim"]
    vcfg = dict(vision_dims)
    vcfg["hidden_act"] = hidden_act
    cfg = CLIPConfig(vision_config=vcfg,
                     text_config=dict(hidden_size=32, intermediate_size=64, num_hidden_layers=1,
                                      num_attention_heads=2, vocab_size=64, max_position_embeddings=8,
                                      projection_dim=proj),
                     projection_dim=proj)
    cfg.vision_config._attn_implementation = "eager"
    cfg._attn_implementation = "eager"
    clip = CLIPModel(cfg)

    class _Backbone(nn.Module):
        def __init__(self, model):
            super().__init__()
            self.model = model
            self.features_size = model.vision_model.config.hidden_size

    bb = _Backbone(clip)
    center = torch.randn(nclass, center_dim).sign()
    text_projection = nn.Sequential(nn.Linear(center_dim, center_dim), nn.ReLU(), nn.Linear(center_dim, nbit))
    upt = attr_dict(multi=True, num_heads=8, dropout=upt_dropout, ensemble_method="concat", single_hash_fc=True,
                    hash_pe=True)
    model = ref_coop.LGHWithFixedPrompt(bb, nbit, nclass, ncontext, add_bn=True, use_before_projection=True,
                                        upt_config=upt, fixed_center=center, text_projection=text_projection,
                                        has_adapter=True, adapter_bottleneck_dim=adapter_bottleneck_dim,
                                        concept_reg=True)
    vm = model.backbone.vision_model
    enc_cfg = vm.encoder.config
    vm.encoder = _EncoderModule(vm.encoder.layers, enc_cfg)
    return model


def _make_encoder_module():
    import torch.nn as nn

    class _EncoderModuleImpl(nn.Module):
        def __init__(self, layers, config):
            super().__init__()
            self.layers = layers
            self.config = config

        def forward(self, inputs_embeds, **kw):
            return FourXEncoderLoop(self.layers, self.config)(inputs_embeds, **kw)

    return _EncoderModuleImpl


def _EncoderModule(layers, config):
    return _make_encoder_module()(layers, config)
