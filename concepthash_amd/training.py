"""Training step of the ConceptHash adapters on the MI355X HIP library (C-ABI `ch_trainer_*` / `ch_train_*`,
include/concepthash_hip.h).

What runs where (reference: trainers/coop.py:107-131 `train_one_batch`):
  * HIP library: the encoder forward with saved activations and its backward -- everything on the [B*N, *] activations, the
    gradients of the 24 adapters' parameters, and the gradient w.r.t. the concept tokens;
  * this module: the parameter plumbing.  The adapters' parameters live in ONE fp32 device arena (the layout
    `ch_adapter_arena_numel` describes); the `nn.Parameter`s of the adapter modules are re-pointed at views of it, their
    `.grad`s at views of the gradient arena the library overwrites, so an ordinary optimizer updates the arena in place and
    `refresh()` re-derives the library's bf16 / folded / transposed working copies;
  * the caller's autograd: the 4-token concept generator, the hashing head on [B, Q, D] and the loss (models/arch/coop.py,
    models/loss/coop.py) -- a few MFLOP per step -- joined to the library by `EncoderFunction`.

There is no CPU fallback: without the HIP library (or without a GPU) construction raises.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Sequence

import torch

from . import _lib
from .encoder import VM, ConceptHashEncoder

ADAPTER_FIELDS = ("adapter_layer_norm.weight", "adapter_layer_norm.bias", "down_proj.weight", "down_proj.bias", "up_proj.weight",
                  "up_proj.bias", "scale")


class TrainEngine:
    """Owns a frozen-weights `ch_model` + a `ch_trainer` for batches up to `max_batch`, and the adapter arenas.

    adapters: list over layers of (adapt_mlp_1, adapt_mlp_2) modules whose parameters are re-pointed into the arena."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], adapters: Sequence[Sequence[torch.nn.Module]], heads: int,
                 upt_heads: int = 8, act: str = "quick_gelu", max_batch: int = 64, device=None, image_size=None, options=None):
        """options: ch_model_set_option settings of the frozen model the trainer is created on -- "train_chains", "train_chain_min_rows",
        "train_prune_last" are read by ch_trainer_create, "pp_min_k" etc. by every launch (ConceptHashEncoder)."""
        self.lib = _lib.load()
        # the frozen model: its own workspace is never used by training, so it is sized for one image
        self.encoder = ConceptHashEncoder(state_dict, heads=heads, upt_heads=upt_heads, act=act, max_batch=1, device=device,
                                          image_size=image_size, options=options)
        self.device = self.encoder.device
        self.cfg = self.encoder.cfg
        self.max_batch = int(max_batch)
        n = int(self.lib.ch_adapter_arena_numel(self.encoder._h))
        if n <= 0:
            raise RuntimeError("the model has no adapters: nothing to train in the encoder")
        npad = (n + 3) // 4 * 4          # ch_sgd_step works on float4s; the pad elements stay zero
        self.params = torch.zeros(npad, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(npad, dtype=torch.float32, device=self.device)
        self.momentum_buf = None         # created by the fused SGD step (fuse_adapter_sgd)
        self._views: List[tuple] = []   # (parameter, grad view)
        D, b, L = self.cfg["dim"], self.cfg["adapter_dim"], self.cfg["layers"]
        sizes = (D, D, b * D, b, D * b, D, 1)
        if len(adapters) != L or any(len(pair) != 2 for pair in adapters):
            raise ValueError("adapters must be a list over layers of (adapt_mlp_1, adapt_mlp_2)")
        off = 0
        with torch.no_grad():
            for pair in adapters:
                for mod in pair:
                    named = dict(mod.named_parameters())
                    for field, size in zip(ADAPTER_FIELDS, sizes):
                        p = named[field]
                        if p.numel() != size:
                            raise ValueError(f"adapter parameter {field} has {p.numel()} elements, expected {size}")
                        view = self.params[off:off + size].view(p.shape)
                        view.copy_(p.detach().to(self.device, torch.float32))
                        p.data = view                     # the module's parameter IS the arena from here on
                        self._views.append((p, self.grads[off:off + size].view(p.shape)))
                        off += size
        assert off == n
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ch_trainer_create(self.encoder._h, self.max_batch, _lib.ptr(self.params), _lib.ptr(self.grads),
                                                  ctypes.byref(h)), "ch_trainer_create")
        self._t = h
        self._stale = False
        # makes autograd call EncoderFunction.backward (which produces the adapters' gradients) even when nothing upstream of the
        # concept tokens requires a gradient
        self.anchor = torch.zeros((), device=self.device, requires_grad=True)
        self.generation = 0          # forwards so far: the saved activations belong to the LAST one only

    def close(self):
        if getattr(self, "_t", None) is not None and self._t.value:
            self.lib.ch_trainer_destroy(self._t)
            self._t = ctypes.c_void_p()
        if getattr(self, "encoder", None) is not None:
            self.encoder.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_bytes(self) -> int:
        return int(self.lib.ch_trainer_bytes(self._t)) + self.encoder.device_bytes

    def adapter_parameters(self) -> List[torch.nn.Parameter]:
        return [p for p, _ in self._views]

    def mark_stale(self):
        """The arena changed (an optimizer step): the working copies are re-derived before the next forward."""
        self._stale = True

    def sync_versions(self):
        """In-place updates of the adapter parameters (optimizer.step(), load_state_dict) bump their tensor versions: compare
        with the versions seen at the last refresh."""
        ver = sum(p._version for p, _ in self._views)
        if ver != getattr(self, "_ver", None):
            self._ver = ver
            self._stale = True

    def refresh(self, stream=None):
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ch_trainer_refresh(self._t, _lib.stream_ptr(stream)), "ch_trainer_refresh")
        self._stale = False

    def forward(self, images: torch.Tensor, concept_tokens: torch.Tensor, want_cls: bool = False, want_attn=False):
        """want_attn: False | True (the last layer's concept-token attention rows [B, heads, Q, Np]) | "all" (every layer's,
        [L, B, heads, Q, Np]); `backward` then takes the cotangent in the same form."""
        self.encoder._check_images(images)
        B = images.shape[0]
        if B > self.max_batch:
            raise ValueError(f"batch {B} exceeds the trainer's max_batch {self.max_batch}")
        c = self.cfg
        ct = concept_tokens.detach().to(self.device, torch.float32).reshape(c["ncontext"], c["dim"]).contiguous()
        if self._stale:
            self.refresh()
        images = images.contiguous()
        hf = torch.empty(B, c["ncontext"], c["dim"], dtype=torch.float32, device=self.device)
        cls = torch.empty(B, c["dim"], dtype=torch.float32, device=self.device) if want_cls else None
        npatch = (c["image_size"] // c["patch"]) ** 2
        all_layers = want_attn == "all"
        shape = ((c["layers"],) if all_layers else ()) + (B, c["heads"], c["ncontext"], npatch)
        attn = torch.empty(shape, dtype=torch.float32, device=self.device) if want_attn else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ch_train_forward(self._t, _lib.ptr(images), 0 if images.dtype == torch.float32 else 1, B,
                                                 _lib.ptr(ct), _lib.ptr(hf), _lib.ptr(cls), _lib.ptr(attn), 1 if all_layers else 0,
                                                 _lib.stream_ptr()), "ch_train_forward")
        self.generation += 1
        return (hf, cls, attn) if want_attn else (hf, cls)

    def grads_live(self) -> bool:
        """True when the adapters' `.grad`s still are the arena views of an earlier backward, i.e. no `zero_grad()` (set_to_none,
        the default) ran since: the next backward must ADD to them, as autograd does for every other parameter."""
        return any(p.grad is not None and p.grad.data_ptr() == gview.data_ptr() for p, gview in self._views)

    def drop_grads(self) -> None:
        """`optimizer.zero_grad(set_to_none=True)` for the adapters alone: the next backward overwrites the arena instead of accumulating
        (what a timing loop that calls `backward` repeatedly wants -- otherwise every call after the first also pays the accumulation's
        clone + add of the arena and the gradients grow without bound)."""
        for p, _ in self._views:
            p.grad = None

    def backward(self, d_hash_features: torch.Tensor, d_concept_attn: torch.Tensor = None) -> torch.Tensor:
        """`ch_train_backward` OVERWRITES the gradient arena.  Gradient accumulation (two backward calls without a zero_grad in
        between: micro-batches, several losses) therefore keeps the earlier arena aside and adds it back -- the adapters then
        accumulate exactly as the head's parameters do under autograd (one extra copy + add of the arena per accumulated call)."""
        c = self.cfg
        g = d_hash_features.detach().to(self.device, torch.float32).contiguous()
        ga = d_concept_attn.detach().to(self.device, torch.float32).contiguous() if d_concept_attn is not None else None
        dct = torch.empty(c["ncontext"], c["dim"], dtype=torch.float32, device=self.device)
        earlier = self.grads.clone() if self.grads_live() else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ch_train_backward(self._t, _lib.ptr(g), _lib.ptr(ga), _lib.ptr(dct), _lib.stream_ptr()),
                       "ch_train_backward")
        if earlier is not None:
            self.grads.add_(earlier)
        for p, gview in self._views:      # optimizer.zero_grad(set_to_none=True) drops the views: put them back
            p.grad = gview
        return dct


class EncoderFunction(torch.autograd.Function):
    """hash_features = encoder(images; concept_tokens, adapters): forward / backward by the HIP library.  The adapters'
    gradients do not travel through autograd -- backward writes them into the parameters' `.grad` views (TrainEngine)."""

    @staticmethod
    def forward(ctx, concept_tokens, images, engine: TrainEngine, anchor, want_attn=False):
        """-> hash_features [B, Q, D], or (hash_features, concept_attention) with want_attn: the attention rows of the concept tokens
        over the patch tokens -- the last layer's [B, heads, Q, Np] (True) or every layer's [L, B, heads, Q, Np] ("all") --
        differentiable as well."""
        out = engine.forward(images, concept_tokens, want_attn=want_attn)     # want_attn: False | True (last layer) | "all"
        ctx.engine = engine
        ctx.generation = engine.generation
        ctx.ct_shape = concept_tokens.shape
        ctx.want_attn = want_attn
        return (out[0], out[2]) if want_attn else out[0]

    @staticmethod
    def backward(ctx, d_hf, d_attn=None):
        if ctx.generation != ctx.engine.generation:
            raise RuntimeError("EncoderFunction.backward: another training forward ran on this engine since this graph was built; the "
                               "library keeps the saved activations of the last forward only (one forward -> one backward)")
        dct = ctx.engine.backward(d_hf, d_attn if ctx.want_attn else None)
        return dct.view(ctx.ct_shape), None, None, None, None


def fuse_adapter_sgd(optimizer, model):
    """torch.optim.SGD over the adapters = foreach kernels over 168 views of one arena (0.45 ms per step for B/16).  When the first
    param group is exactly the adapters of `model` and its training engine exists, `optimizer.step()` updates that group with ONE
    launch over the arena (`ch_sgd_step`, same arithmetic: weight decay, momentum buffer, dampening, nesterov) and lets torch step
    the remaining groups.  Anything else (another optimizer class, maximize, a closure, the engine not built yet, a missing gradient)
    falls through to the unmodified step.  Call BEFORE the lr scheduler is created (schedulers wrap `optimizer.step`)."""
    if type(optimizer) is not torch.optim.SGD:
        return optimizer
    torch_step = optimizer.step
    fused = {"steps": 0}

    def step(self, closure=None):      # installed as a bound method: lr schedulers wrap `optimizer.step.__func__`
        eng = getattr(model, "_train_engine", None)
        g0 = optimizer.param_groups[0]
        params = g0["params"]
        ok = (closure is None and eng is not None and not g0.get("maximize", False) and len(params) == len(eng._views)
              and {id(p) for p in params} == {id(v) for v, _ in eng._views} and all(p.grad is not None for p in params))
        if not ok:
            return torch_step(closure)
        first = eng.momentum_buf is None
        if first and g0["momentum"] != 0:
            restored = getattr(optimizer, "restored_adapter_momentum", None)
            if restored is not None and restored.numel() == eng.params.numel():
                eng.momentum_buf, first = restored.to(eng.device, torch.float32).clone(), False
            else:
                eng.momentum_buf = torch.zeros_like(eng.params)
        with torch.cuda.device(eng.device):
            _lib.check(eng.lib.ch_sgd_step(_lib.ptr(eng.params), _lib.ptr(eng.grads), _lib.ptr(eng.momentum_buf), eng.params.numel(),
                                           float(g0["lr"]), float(g0["momentum"]), float(g0["weight_decay"]), float(g0["dampening"]),
                                           int(bool(g0["nesterov"])), int(first), _lib.stream_ptr()), "ch_sgd_step")
        eng.mark_stale()                                   # the library's working copies are re-derived before the next forward
        torch.autograd.graph.increment_version(params[0])  # ... and the model's evaluation engine sees changed parameters
        fused["steps"] += 1
        g0["params"] = []
        try:
            return torch_step()
        finally:
            g0["params"] = params

    import types
    optimizer.step = types.MethodType(step, optimizer)
    optimizer.fused_adapter_steps = fused
    return optimizer


def adapters_from_state_dict(state_dict, layers: int, dim: int, bottleneck: int) -> list:
    """Stand-alone adapter modules (parameter holders) initialised from a reference-layout state_dict: what TrainEngine needs when
    there is no model object around it (benchmarks)."""
    from models.layers.adapter import Adapter
    out = []
    for l in range(layers):
        pair = []
        for a in (1, 2):
            prefix = VM + f"encoder.layers.{l}.adapt_mlp_{a}."
            m = Adapter(dim, bottleneck)
            m.load_state_dict({k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)})
            pair.append(m)
        out.append(tuple(pair))
    return out


def encoder_step_flops(cfg: dict) -> tuple:
    """(forward, backward) algorithmic FLOPs per image of the encoder's training step: the backward's input-gradient products
    equal the forward's linears, the adapters add their weight-gradient products, attention backward is 2.5x its forward."""
    D, L, M, b = cfg["dim"], cfg["layers"], cfg["ffn"], cfg["adapter_dim"]
    N = 1 + (cfg["image_size"] // cfg["patch"]) ** 2 + cfg["ncontext"]
    lin = 2.0 * N * (4 * D * D + 2 * D * M + 4 * D * b)
    attn = 4.0 * N * N * D
    return L * (lin + attn), L * (lin + 2.5 * attn + 2.0 * N * 4 * D * b)


def adapter_modules(vision_model) -> list:
    return [(layer.adapt_mlp_1, layer.adapt_mlp_2) for layer in vision_model.encoder.layers]


def benchmark_full_step(cfg: dict, state_dict, batches, steps: int = 10, warmup: int = 3) -> dict:
    """Wall clock of the whole training step through the drop-in surface -- the reference's train_one_batch (trainers/coop.py:107-131):
    `LGHWithFixedPrompt` in train mode, `LGHLoss` (shipped terms), `loss.backward()`, `torch.optim.SGD.step()` with the adapters' group
    fused (fuse_adapter_sgd), synchronised once per measurement.  cfg: a concepthash_amd.synthetic.CONFIGS entry."""
    import time

    from concepthash_amd import config as cfglib
    from concepthash_amd import synthetic
    from models.arch.coop import LGHWithFixedPrompt
    from models.backbone.clip import CLIP
    from models.loss.coop import LGHLoss
    dims = dict(hidden_size=cfg["D"], num_hidden_layers=cfg["L"], num_attention_heads=cfg["heads"], intermediate_size=cfg["M"],
                patch_size=cfg["patch"], image_size=cfg["image"], projection_dim=cfg["P"], hidden_act="quick_gelu")
    upt = cfglib.DictConfig(multi=True, num_heads=8, dropout=0.1, ensemble_method="concat", single_hash_fc=True, hash_pe=True)
    C, cd = state_dict["center"].shape
    nbit = state_dict["hash_fc.weight"].shape[0] * 4
    tp = torch.nn.Sequential(torch.nn.Linear(cd, cd), torch.nn.ReLU(), torch.nn.Linear(cd, nbit))
    model = LGHWithFixedPrompt(CLIP(dims, allow_random_init=True), nbit, C, 4, add_bn=True, upt_config=upt, fixed_center=torch.zeros(C, cd),
                               text_projection=tp, has_adapter=True, adapter_bottleneck_dim=cfg["b"], concept_reg=True)
    model.load_state_dict(state_dict)
    model = model.cuda().train()
    model.train_max_batch = max(batches)
    crit = LGHLoss(margin=0.2, scale=8, loss_scales=dict(bin_logits=1, cont_logits=1, concept_logits=1), ncontext=4)
    groups = [{"params": list(model.get_adapter().parameters())}, {"params": list(model.get_training_modules().parameters())}]
    model.requires_grad_(False)
    for g in groups:
        for p in g["params"]:
            p.requires_grad_(True)
    opt = fuse_adapter_sgd(torch.optim.SGD(groups, lr=1e-3, momentum=0.9, weight_decay=5e-4), model)
    out = {}
    for B in batches:
        x = synthetic.synthetic_images(B, cfg["image"]).to("cuda", torch.bfloat16)
        y = torch.randint(0, C, (B,), device="cuda")
        for it in range(warmup + steps):
            if it == warmup:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            opt.zero_grad()
            crit(model(x)[1], y).backward()
            opt.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        out[B] = {"full_step_ms": round(ms, 3), "images_per_s": round(B / ms * 1e3, 1),
                  "what": "model.train() forward + LGHLoss + backward + SGD step (adapter group fused), wall clock"}
    model._drop_train_engine()
    return out
