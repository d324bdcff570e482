"""Host-side owner of a ``ch_model`` (HIP encoder).  PyTorch is used for device memory and streams only.

Mirrors the reference's model object for the encode path: built from a reference-layout ``state_dict``
(aliases ``adapter_params.*`` / ``trainable_params.*`` are dropped), evaluated with
``encode(images) -> dict`` whose keys follow ``LGHWithoutText.forward`` (models/arch/coop.py:582-598).
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict, Iterable, Optional

import numpy as np
import torch

from . import _lib

VM = "backbone.vision_model."
_SKIP_PREFIXES = ("adapter_params.", "trainable_params.", "backbone.text_model.", "backbone.text_projection",
                  "token_embeds")
_SKIP_KEYS = ("backbone.logit_scale", "hash_bn.num_batches_tracked", VM + "embeddings.position_ids")


def infer_config(sd: Dict[str, torch.Tensor], heads: Optional[int] = None) -> dict:
    """Model dimensions from a reference-layout state_dict."""
    pw = sd[VM + "embeddings.patch_embedding.weight"]
    D, _, p, _ = pw.shape
    npos = sd[VM + "embeddings.position_embedding.weight"].shape[0]
    grid = int(round(math.sqrt(npos - 1)))
    if grid * grid != npos - 1:
        raise ValueError(f"position embedding has {npos} rows; expected 1 + grid^2")
    L = 0
    while (VM + f"encoder.layers.{L}.layer_norm1.weight") in sd:
        L += 1
    if L == 0:
        raise ValueError("state_dict has no encoder layers under " + VM + "encoder.layers.*")
    ad_key = VM + "encoder.layers.0.adapt_mlp_1.down_proj.weight"
    Q = sd["hash_pe"].shape[1] if "hash_pe" in sd else sd["hash_queries"].shape[1]
    cfg = dict(image_size=grid * p, patch=p, dim=D, layers=L, heads=heads if heads else D // 64,
               ffn=sd[VM + "encoder.layers.0.mlp.fc1.weight"].shape[0],
               adapter_dim=sd[ad_key].shape[0] if ad_key in sd else 0, ncontext=Q,
               nbit=sd["hash_fc.weight"].shape[0] * Q, nclass=sd["center"].shape[0],
               proj_dim=sd["hash_queries"].shape[2], center_dim=sd["center"].shape[1])
    return cfg


def interpolate_pos_embedding(pos: torch.Tensor, new_grid: int) -> torch.Tensor:
    """Position-embedding table for an input resolution other than the pretrain one: `interpolate_pos_encoding`
    (models/arch/coop.py:429-450; twin models/backbone/clip.py:69-89), square inputs.

    pos [1 + g*g, D] -> [1 + new_grid^2, D]: the class row is kept, the g x g patch grid is resized with the arithmetic of the
    call the reference makes -- `F.interpolate(scale_factor=(new_grid + 0.1) / g, mode="bicubic")`, align_corners False:
    source coordinate (dst + 0.5) / scale - 0.5, Keys cubic with A = -0.75 on the four neighbours, indices clamped to the grid.
    The table does not depend on the image, so it is folded once when the model is built (host, fp32) instead of per forward."""
    pos = pos.detach().to("cpu", torch.float32)
    n = pos.shape[0] - 1
    g = int(round(math.sqrt(n)))
    if g * g != n:
        raise ValueError(f"position embedding has {n + 1} rows; expected 1 + grid^2")
    if new_grid == g:
        return pos.clone()
    grid = pos[1:].reshape(g, g, -1).numpy().astype(np.float32)
    inv_scale = np.float32(1.0) / np.float32((new_grid + 0.1) / g)      # torch: scale = 1 / scale_factor (fp32 arithmetic)
    A = np.float32(-0.75)

    def taps(o):
        real = inv_scale * np.float32(o + 0.5) - np.float32(0.5)
        i = int(math.floor(real))
        t = np.float32(real - i)

        def c1(x):   # |x| <= 1
            return ((A + 2) * x - (A + 3)) * x * x + 1

        def c2(x):   # 1 < |x| < 2
            return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
        w = np.array([c2(t + 1), c1(t), c1(1 - t), c2(2 - t)], dtype=np.float32)
        idx = np.clip(np.arange(i - 1, i + 3), 0, g - 1)
        return idx, w

    rows = [taps(o) for o in range(new_grid)]
    out = np.empty((new_grid, new_grid, grid.shape[2]), dtype=np.float32)
    for oy, (iy, wy) in enumerate(rows):
        for ox, (ix, wx) in enumerate(rows):
            patch = grid[np.ix_(iy, ix)]                                 # [4, 4, D]
            # torch accumulates the horizontal interpolation of each of the four rows, then the vertical one
            horiz = (patch * wx[None, :, None]).sum(1, dtype=np.float32)
            out[oy, ox] = (horiz * wy[:, None]).sum(0, dtype=np.float32)
    return torch.cat([pos[:1], torch.from_numpy(out.reshape(new_grid * new_grid, -1))], dim=0)


class ConceptHashEncoder:
    """ViT + concept tokens + hashing head on one MI355X, through the C-ABI."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], heads: Optional[int] = None, upt_heads: int = 8,
                 act: str = "quick_gelu", max_batch: int = 256, device: Optional[torch.device] = None,
                 ln_eps: float = 1e-5, bn_eps: float = 1e-5, image_size: Optional[int] = None,
                 options: Optional[Dict[str, int]] = None):
        """options: ch_model_set_option settings of this handle (`_lib.OPTION_KEYS`: "streams", "ln_fold", "prune_last", "pp_min_k",
        "resid_nt", "nt_out", ...).  The CH_* environment variables of earlier rounds are DEBUG overrides read here, in the wrapper
        (`_lib.env_option_overrides`), and lose against an explicit entry of `options`."""
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("ConceptHashEncoder needs a GPU (MI355X); there is no CPU fallback")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        cfg = infer_config(state_dict, heads)
        cfg.update(upt_heads=upt_heads, act={"quick_gelu": 0, "gelu": 1}[act], max_batch=max_batch)
        self.pretrain_image_size = cfg["image_size"]
        if image_size is not None and int(image_size) != cfg["image_size"]:
            # input resolution other than the pretrain one: interpolated position table (reference coop.py:429-450)
            if int(image_size) % cfg["patch"]:
                raise ValueError(f"image_size {image_size} is not a multiple of the patch size {cfg['patch']}")
            state_dict = dict(state_dict)
            key = VM + "embeddings.position_embedding.weight"
            state_dict[key] = interpolate_pos_embedding(state_dict[key], int(image_size) // cfg["patch"])
            cfg["image_size"] = int(image_size)
        self.cfg = cfg
        c = _lib.ModelConfig(ln_eps=ln_eps, bn_eps=bn_eps, **cfg)
        keep = []
        entries = []
        for k, v in state_dict.items():
            if k.startswith(_SKIP_PREFIXES) or k in _SKIP_KEYS or not torch.is_tensor(v):
                continue
            # where it is (a model that was moved to the GPU is ingested device -> device: 572 pageable round trips of a ViT-B/16
            # were 0.7 s of an evaluation command's first batch); tensors on ANOTHER device go through the host
            t = v.detach()
            if t.is_cuda and t.device != self.device:
                t = t.to("cpu")
            t = t.to(torch.float32).contiguous()
            keep.append(t)
            entries.append((k.encode(), t))
        arr = (_lib.Tensor * len(entries))()
        for i, (name, t) in enumerate(entries):
            arr[i].name = name
            arr[i].data = ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_float))
            arr[i].numel = t.numel()
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            torch.cuda.current_stream(self.device).synchronize()     # device-resident tensors: whatever produced them has finished
            _lib.check(self.lib.ch_model_create(ctypes.byref(c), arr, len(entries), ctypes.byref(handle)),
                       "ch_model_create")
        self._h = handle
        # ("graph_max_batch": replay of small batches as one captured hipGraph -- measured neutral, 2.07 vs 2.09 ms at batch 8: the step is
        #  tile-latency bound on the GPU, not launch bound on the host -- stays opt-in)
        for k, v in {**_lib.env_option_overrides(), **(options or {})}.items():
            self.set_option(k, v)
        self.nbit = cfg["nbit"]
        self.words = (self.nbit + 63) // 64
        self.ntok = 1 + (cfg["image_size"] // cfg["patch"]) ** 2 + cfg["ncontext"]
        self.has_concept = "concept_ce.centroids" in state_dict
        self.has_pooled = (VM + "post_layernorm.weight") in state_dict and "backbone.visual_projection.weight" in state_dict

    # -- lifetime ------------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.ch_model_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_bytes(self) -> int:
        return int(self.lib.ch_model_device_bytes(self._h))

    # -- per-handle options (include/concepthash_hip.h: ch_model_set_option) ------------------------------------------
    def set_option(self, key: str, value: int) -> None:
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ch_model_set_option(self._h, key.encode(), int(value)), f"ch_model_set_option({key!r})")

    def get_option(self, key: str) -> int:
        v = ctypes.c_int64()
        _lib.check(self.lib.ch_model_get_option(self._h, key.encode(), ctypes.byref(v)), f"ch_model_get_option({key!r})")
        return int(v.value)

    @property
    def flops_per_image(self) -> float:
        return float(self.lib.ch_model_flops_per_image(self._h))

    # -- launch profiler (bench.py roofline) ------------------------------------------------------------------------
    def profile_begin(self, max_launches: int) -> None:
        _lib.check(self.lib.ch_model_profile_begin(self._h, int(max_launches)), "ch_model_profile_begin")

    def profile_end(self) -> Dict[str, dict]:
        n = len(_lib.CATEGORIES)
        ms = (ctypes.c_double * n)()
        cnt = (ctypes.c_int64 * n)()
        fl = (ctypes.c_double * n)()
        _lib.check(self.lib.ch_model_profile_end(self._h, n, ms, cnt, fl), "ch_model_profile_end")
        return {name: dict(ms=ms[i], launches=int(cnt[i]), flops=fl[i]) for i, name in enumerate(_lib.CATEGORIES)
                if name != "end"}

    @property
    def launches_per_encode(self) -> int:
        """upper bound of the kernel launches of one ch_encode chain (profile-event capacity)"""
        L, ad = self.cfg["layers"], 1 if self.cfg["adapter_dim"] > 0 else 0
        return 3 + L * (7 + 8 * ad) + 4

    # -- encode ---------------------------------------------------------------------------------------------------
    def _check_images(self, images: torch.Tensor):
        s = self.cfg["image_size"]
        if images.dim() != 4 or images.shape[1] != 3 or images.shape[2] != s or images.shape[3] != s:
            raise ValueError(f"images must be [B,3,{s},{s}], got {tuple(images.shape)}")
        if images.dtype not in (torch.float32, torch.bfloat16):
            raise TypeError("images must be float32 or bfloat16")
        if images.device != self.device:
            raise ValueError(f"images are on {images.device}, the model is on {self.device}")

    def encode(self, images: torch.Tensor, want: Iterable[str] = ("codes", "packed"), stream=None) -> Dict[str, torch.Tensor]:
        """want: subset of {codes, packed, logits_cont, logits_bin, logits_concept, hash_features, image_features, concept_attn,
        concept_attn_layers}; 'codes' is always produced.  Batches larger than max_batch are processed in chunks on the same stream.
        concept_attn: the last layer's attention of the concept tokens over the patch tokens [B, heads, Q, Np];
        concept_attn_layers: the same rows of EVERY layer [L, B, heads, Q, Np] (what the reference's `attn_cache` holds for those
        tokens, models/arch/coop.py:481-482)."""
        self._check_images(images)
        images = images.contiguous()
        want = set(want) | {"codes"}
        unknown = want - {"codes", "packed", "logits_cont", "logits_bin", "logits_concept", "hash_features", "image_features",
                          "concept_attn", "concept_attn_layers"}
        if unknown:
            raise KeyError(f"unknown outputs {sorted(unknown)}")
        B = images.shape[0]
        c = self.cfg
        dev = self.device
        out = {"codes": torch.empty(B, c["nbit"], dtype=torch.float32, device=dev)}
        if "packed" in want:
            out["packed"] = torch.empty(B, self.words, dtype=torch.int64, device=dev)
        if "logits_cont" in want:
            out["logits_cont"] = torch.empty(B, c["nclass"], dtype=torch.float32, device=dev)
        if "logits_bin" in want:
            out["logits_bin"] = torch.empty(B, c["nclass"], dtype=torch.float32, device=dev)
        if "hash_features" in want:
            out["hash_features"] = torch.empty(B, c["ncontext"], c["dim"], dtype=torch.float32, device=dev)
        if "image_features" in want:
            out["image_features"] = torch.empty(B, c["proj_dim"], dtype=torch.float32, device=dev)
        all_layers = "concept_attn_layers" in want
        if "concept_attn" in want and not all_layers:   # last-layer attention of the concept tokens over the patch tokens [B, heads, Q, Np]
            out["concept_attn"] = torch.empty(B, c["heads"], c["ncontext"], self.ntok - 1 - c["ncontext"], dtype=torch.float32,
                                              device=dev)
        concept_chunks, attn_chunks = [], []
        dt = 0 if images.dtype == torch.float32 else 1
        sp = _lib.stream_ptr(stream)
        mb = c["max_batch"]
        with torch.cuda.device(dev):
            for b0 in range(0, B, mb):
                b1 = min(B, b0 + mb)
                lc = None
                if "logits_concept" in want:
                    lc = torch.empty(c["ncontext"], b1 - b0, c["nclass"], dtype=torch.float32, device=dev)
                    concept_chunks.append(lc)

                def sl(name):
                    return _lib.ptr(out[name][b0:b1]) if name in out else ctypes.c_void_p(0)

                attn_ptr = sl("concept_attn")
                if all_layers:       # [L, chunk, heads, Q, Np]: the library strides the layers by the chunk's batch
                    al = torch.empty(c["layers"], b1 - b0, c["heads"], c["ncontext"], self.ntok - 1 - c["ncontext"],
                                     dtype=torch.float32, device=dev)
                    attn_chunks.append(al)
                    attn_ptr = _lib.ptr(al)
                _lib.check(self.lib.ch_encode(self._h, _lib.ptr(images[b0:b1]), dt, b1 - b0, sl("codes"), sl("packed"),
                                              sl("logits_cont"), sl("logits_bin"), _lib.ptr(lc), sl("hash_features"),
                                              sl("image_features"), attn_ptr, 1 if all_layers else 0, sp), "ch_encode")
        if concept_chunks:
            out["logits_concept"] = concept_chunks[0] if len(concept_chunks) == 1 else torch.cat(concept_chunks, dim=1)
        if attn_chunks:
            out["concept_attn_layers"] = attn_chunks[0] if len(attn_chunks) == 1 else torch.cat(attn_chunks, dim=1)
            if "concept_attn" in want:
                out["concept_attn"] = out["concept_attn_layers"][-1]
        return out

    def hidden_states(self, images: torch.Tensor, layer: int, stream=None) -> torch.Tensor:
        """Parity tap: fp32 residual stream [B, N, D] after `layer` encoder layers (0 = after pre-LN)."""
        self._check_images(images)
        images = images.contiguous()
        B = images.shape[0]
        out = torch.empty(B, self.ntok, self.cfg["dim"], dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ch_encode_hidden(self._h, _lib.ptr(images), 0 if images.dtype == torch.float32 else 1, B,
                                                 layer, _lib.ptr(out), _lib.stream_ptr(stream)), "ch_encode_hidden")
        return out
