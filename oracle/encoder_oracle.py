"""CPU oracle for the ConceptHash *encode* path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
The product path (``concepthash_amd``) never does; it fails loudly when the HIP library is missing.

This is a from-scratch fp32 restatement (plain torch CPU ops on a flat ``state_dict``) of what the reference
computes in ``LGHWithFixedPrompt.forward``.  Each function cites the reference lines it follows
(paths relative to /root/reference):

  embeddings ........ models/arch/coop.py:452-466  (+ HF CLIPVisionEmbeddings: Conv2d k=s=patch, no bias)
  concept tokens .... models/arch/coop.py:413-427  (non-"v2" residual form: norm(x) + f(x))
  concat + pre-LN ... models/arch/coop.py:468-472
  encoder layer ..... models/layers/adapter.py:127-177 (CLIPEncoderLayerWithAdapter.forward)
  adapter ........... models/layers/adapter.py:46-60   (LN -> down -> exact GELU -> up -> * scale)
  attention / MLP ... third-party transformers.models.clip.modeling_clip (5.15.0: eager_attention_forward
                      :259-277, CLIPAttention :280-335, CLIPMLP :338-350); quick_gelu = x*sigmoid(1.702x)
  token select ...... models/arch/coop.py:503-509 (raw last-layer hidden state of the last Q tokens)
  hash head ......... models/arch/coop.py:544-559 (+hash_pe -> shared Linear(D->nbit/Q) -> concat -> BN eval)
  centre logits ..... models/arch/coop.py:573-580, :624-625
  concept logits .... models/arch/coop.py:269-276 + models/layers/cossim.py:37-82 (group=1 branch)

PARITY PINNING: this restatement is pinned by ``tests/golden/encode_*.npz`` -- outputs of the reference's own
(unmodified) model classes run in the build container via ``oracle/gen_golden.py``.  The stock ViT block
arithmetic is third-party (``transformers``, requirement unpinned in the reference; 5.15.0 installed), so
parity is pinned w.r.t. 5.15.0 eager semantics, not the author's original 4.x (differences ~1e-6 fp32).

``emulate_bf16=True`` rounds activations/weights to bf16 at the same points where the HIP path stores bf16
(GEMM operands), with fp32 accumulation and an fp32 residual stream -- used to separate "kernel bug" from
"bf16 rounding" when comparing against the HIP path.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

VM = "backbone.vision_model."


def _bf16(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


class _R:
    """rounding policy: identity in fp32 mode, bf16 round-trip in emulate mode."""

    def __init__(self, emulate_bf16: bool):
        self.on = emulate_bf16

    def __call__(self, x):
        return _bf16(x) if self.on else x


def infer_dims(sd: Dict[str, torch.Tensor]) -> dict:
    """Model dimensions from a reference-layout state_dict."""
    pw = sd[VM + "embeddings.patch_embedding.weight"]
    D, _, p, _ = pw.shape
    npos = sd[VM + "embeddings.position_embedding.weight"].shape[0]
    L = 0
    while (VM + f"encoder.layers.{L}.layer_norm1.weight") in sd:
        L += 1
    M = sd[VM + "encoder.layers.0.mlp.fc1.weight"].shape[0]
    bdim = sd[VM + "encoder.layers.0.adapt_mlp_1.down_proj.weight"].shape[0] \
        if (VM + "encoder.layers.0.adapt_mlp_1.down_proj.weight") in sd else 0
    Q = sd["hash_pe"].shape[1]
    sub = sd["hash_fc.weight"].shape[0]
    P = sd["hash_queries"].shape[2]
    C = sd["center"].shape[0]
    return dict(D=D, patch=p, npos=npos, grid=int(round(math.sqrt(npos - 1))), L=L, M=M, b=bdim, Q=Q,
                nbit=sub * Q, sub=sub, P=P, C=C)


def layer_norm(x, w, b, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def concept_tokens(sd, num_heads: int) -> torch.Tensor:
    """forward_hash_query (models/arch/coop.py:413-427), eval mode (dropout off).  Input independent.

    x = hash_queries (1,Q,P); x = LN1(x) + MHA(x,x,x); x = LN2(x) + FFN(x); x = Linear(P->D)(x)
    (the non-standard residual -- norm(x) + f(x) with f applied to the *un-normalised* x -- is what the
    reference does when upt_config.v2 is unset.)
    """
    x = sd["hash_queries"].float()  # (1,Q,P)
    _, Q, P = x.shape
    hd = P // num_heads
    W = sd["hash_attention.sa.in_proj_weight"].float()
    bqkv = sd["hash_attention.sa.in_proj_bias"].float()
    qkv = x @ W.t() + bqkv  # (1,Q,3P)
    q, k, v = qkv.split(P, dim=-1)

    def heads(t):
        return t.reshape(1, Q, num_heads, hd).transpose(1, 2)  # (1,h,Q,hd)

    q, k, v = heads(q), heads(k), heads(v)
    att = torch.softmax((q * (hd ** -0.5)) @ k.transpose(-1, -2), dim=-1)
    o = (att @ v).transpose(1, 2).reshape(1, Q, P)
    sa = o @ sd["hash_attention.sa.out_proj.weight"].float().t() + sd["hash_attention.sa.out_proj.bias"].float()
    x = layer_norm(x, sd["hash_attention.norm1.weight"].float(), sd["hash_attention.norm1.bias"].float()) + sa
    ffn = torch.relu(x @ sd["hash_attention.ffn.0.weight"].float().t() + sd["hash_attention.ffn.0.bias"].float())
    ffn = ffn @ sd["hash_attention.ffn.3.weight"].float().t() + sd["hash_attention.ffn.3.bias"].float()
    x = layer_norm(x, sd["hash_attention.norm2.weight"].float(), sd["hash_attention.norm2.bias"].float()) + ffn
    x = x @ sd["hash_attention.ffn2.weight"].float().t() + sd["hash_attention.ffn2.bias"].float()
    return x  # (1,Q,D)


def projected_center(sd) -> torch.Tensor:
    """get_center (models/arch/coop.py:624-625): text_projection(center); Sequential(Linear, ReLU, Linear)
    per configs/model/concept_hash_final_v1_nosa_apt.yaml:39-49, or a single Linear when keys say so."""
    c = sd["center"].float()
    if "text_projection.0.weight" in sd:
        c = torch.relu(c @ sd["text_projection.0.weight"].float().t() + sd["text_projection.0.bias"].float())
        c = c @ sd["text_projection.2.weight"].float().t() + sd["text_projection.2.bias"].float()
    else:
        c = c @ sd["text_projection.weight"].float().t() + sd["text_projection.bias"].float()
    return c  # (C, nbit)


def embeddings(sd, images: torch.Tensor, R: _R) -> torch.Tensor:
    """forward_visual_embeddings (models/arch/coop.py:452-466) at the pretrain resolution (no interpolation)."""
    w = sd[VM + "embeddings.patch_embedding.weight"].float()
    D, _, p, _ = w.shape
    B = images.shape[0]
    # Conv2d(k=s=p, bias=False) == unfold + GEMM (patch row-major, inner order (c, ky, kx))
    cols = F.unfold(R(images.float()), kernel_size=p, stride=p).transpose(1, 2)  # (B, Np, 3pp)
    pe = cols @ R(w.reshape(D, -1)).t()  # (B, Np, D)
    cls = sd[VM + "embeddings.class_embedding"].float().expand(B, 1, D)
    x = torch.cat([cls, pe], dim=1)
    pos = sd[VM + "embeddings.position_embedding.weight"].float()
    if pos.shape[0] != x.shape[1]:
        pos = interpolate_pos_encoding(pos, images.shape[2], images.shape[3], p)
    return x + pos.unsqueeze(0)


def interpolate_pos_encoding(pos: torch.Tensor, w: int, h: int, patch: int) -> torch.Tensor:
    """interpolate_pos_encoding (models/arch/coop.py:429-450): bicubic resize of the position-embedding grid for inputs other
    than the pretrain resolution -- the same torch call with the same arguments ((w0 + 0.1) / sqrt(N) scale factors)."""
    N = pos.shape[0] - 1
    g = int(math.sqrt(N))
    D = pos.shape[1]
    w0, h0 = w // patch + 0.1, h // patch + 0.1
    grid = F.interpolate(pos[1:].reshape(1, g, g, D).permute(0, 3, 1, 2), scale_factor=(w0 / math.sqrt(N), h0 / math.sqrt(N)),
                         mode="bicubic")
    assert int(w0) == grid.shape[-2] and int(h0) == grid.shape[-1]
    return torch.cat([pos[:1], grid.permute(0, 2, 3, 1).reshape(-1, D)], dim=0)


def attention(sd, pre: str, x_ln: torch.Tensor, heads: int, R: _R, want_probs=False):
    """CLIPAttention eager path: q,k,v,out Linear(D,D)+bias; softmax(q k^T * d^-0.5) v; no mask."""
    B, N, D = x_ln.shape
    hd = D // heads
    xin = R(x_ln)

    def lin(name, t):
        return t @ R(sd[pre + f"self_attn.{name}.weight"].float()).t() + sd[pre + f"self_attn.{name}.bias"].float()

    q = R(lin("q_proj", xin)).reshape(B, N, heads, hd).transpose(1, 2)
    k = R(lin("k_proj", xin)).reshape(B, N, heads, hd).transpose(1, 2)
    v = R(lin("v_proj", xin)).reshape(B, N, heads, hd).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
    p = torch.softmax(s, dim=-1)
    if R.on:
        # HIP path: P is rounded to bf16 before P@V, normalisation by the fp32 row-sum of un-rounded exp
        m = s.max(dim=-1, keepdim=True).values
        e = torch.exp(s - m)
        o = (_bf16(e) @ v) / e.sum(dim=-1, keepdim=True)
    else:
        o = p @ v
    o = R(o.transpose(1, 2).reshape(B, N, D))
    out = lin("out_proj", o)
    return (out, p) if want_probs else (out, None)


def adapter(sd, pre: str, x: torch.Tensor, R: _R) -> torch.Tensor:
    """Adapter.forward (models/layers/adapter.py:46-60), layernorm option 'in', learnable scalar."""
    h = layer_norm(R(x) if R.on else x, sd[pre + "adapter_layer_norm.weight"].float(),
                   sd[pre + "adapter_layer_norm.bias"].float())
    down = R(h) @ R(sd[pre + "down_proj.weight"].float()).t() + sd[pre + "down_proj.bias"].float()
    down = F.gelu(down)  # nn.GELU() default = exact erf form (adapter.py:36)
    up = R(down) @ R(sd[pre + "up_proj.weight"].float()).t() + sd[pre + "up_proj.bias"].float()
    return up * sd[pre + "scale"].float()


def encoder_layer(sd, i: int, h: torch.Tensor, heads: int, R: _R, act: str = "quick_gelu", want_probs=False):
    """CLIPEncoderLayerWithAdapter.forward (models/layers/adapter.py:127-177)."""
    pre = VM + f"encoder.layers.{i}."
    r = h
    a, probs = attention(sd, pre, layer_norm(h, sd[pre + "layer_norm1.weight"].float(),
                                             sd[pre + "layer_norm1.bias"].float()), heads, R, want_probs)
    ad = adapter(sd, pre + "adapt_mlp_1.", a, R) if (pre + "adapt_mlp_1.scale") in sd else 0
    h = r + a + ad
    r = h
    x = R(layer_norm(h, sd[pre + "layer_norm2.weight"].float(), sd[pre + "layer_norm2.bias"].float()))
    m = x @ R(sd[pre + "mlp.fc1.weight"].float()).t() + sd[pre + "mlp.fc1.bias"].float()
    m = quick_gelu(m) if act == "quick_gelu" else F.gelu(m)
    m = R(m) @ R(sd[pre + "mlp.fc2.weight"].float()).t() + sd[pre + "mlp.fc2.bias"].float()
    ad = adapter(sd, pre + "adapt_mlp_2.", m, R) if (pre + "adapt_mlp_2.scale") in sd else 0
    h = r + m + ad
    return h, probs


def _fold_linear(xb, W, bias, gamma, beta, eps=1e-5):
    """LayerNorm folded into the consumer Linear, with the HIP path's rounding points (DESIGN.md section 3.6,
    fold_ln_kernel + the EPI_FOLD_* epilogues): xb = bf16-rounded RAW rows; W' = bf16(W * gamma); c = row sums of W';
    d = bias + W beta; single-pass statistics of the rounded rows; y = rstd * (xb W'^T) + (c * (-mean * rstd) + d)."""
    K = xb.shape[-1]
    Wf = _bf16(W * gamma)
    c = Wf.sum(dim=1)
    d = bias + W @ beta
    mean = xb.sum(dim=-1, keepdim=True) / K
    var = (xb.pow(2).sum(dim=-1, keepdim=True) / K - mean * mean).clamp_min(0.0)
    rstd = torch.rsqrt(var + eps)
    return (xb @ Wf.t()) * rstd + (c * (-mean * rstd) + d)


def _adapter_fold(sd, pre: str, a_b: torch.Tensor) -> torch.Tensor:
    """adapter on the bf16 copy of the sub-block output, adapter LayerNorm folded into down_proj; returns scale * up(...)"""
    down = _fold_linear(a_b, sd[pre + "down_proj.weight"].float(), sd[pre + "down_proj.bias"].float(),
                        sd[pre + "adapter_layer_norm.weight"].float(), sd[pre + "adapter_layer_norm.bias"].float())
    ad = _bf16(F.gelu(down))
    up = ad @ _bf16(sd[pre + "up_proj.weight"].float()).t() + sd[pre + "up_proj.bias"].float()
    return up * sd[pre + "scale"].float()


def encoder_layer_fold(sd, i: int, h: torch.Tensor, heads: int, act: str = "quick_gelu"):
    """The same layer with the rounding points of the HIP path's DEFAULT chain (LayerNorm folded into the consumer GEMMs,
    the sub-block output entering the residual as its bf16 copy): used only to separate "kernel bug" from "bf16 rounding"."""
    pre = VM + f"encoder.layers.{i}."
    B, N, D = h.shape
    hd = D // heads
    R = _R(True)
    if i == 0:   # layer 0: LN1 is computed in fp32 by assemble_preln and rounded afterwards
        xin = _bf16(layer_norm(h, sd[pre + "layer_norm1.weight"].float(), sd[pre + "layer_norm1.bias"].float()))
        qkv = [xin @ _bf16(sd[pre + f"self_attn.{n}.weight"].float()).t() + sd[pre + f"self_attn.{n}.bias"].float()
               for n in ("q_proj", "k_proj", "v_proj")]
    else:
        qkv = [_fold_linear(_bf16(h), sd[pre + f"self_attn.{n}.weight"].float(), sd[pre + f"self_attn.{n}.bias"].float(),
                            sd[pre + "layer_norm1.weight"].float(), sd[pre + "layer_norm1.bias"].float())
               for n in ("q_proj", "k_proj", "v_proj")]
    q, k, v = [_bf16(t).reshape(B, N, heads, hd).transpose(1, 2) for t in qkv]
    s = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
    e = torch.exp(s - s.max(dim=-1, keepdim=True).values)
    o = _bf16(((_bf16(e) @ v) / e.sum(dim=-1, keepdim=True)).transpose(1, 2).reshape(B, N, D))
    a_b = _bf16(o @ _bf16(sd[pre + "self_attn.out_proj.weight"].float()).t() + sd[pre + "self_attn.out_proj.bias"].float())
    h = h + a_b + _adapter_fold(sd, pre + "adapt_mlp_1.", a_b)
    m = _fold_linear(_bf16(h), sd[pre + "mlp.fc1.weight"].float(), sd[pre + "mlp.fc1.bias"].float(),
                     sd[pre + "layer_norm2.weight"].float(), sd[pre + "layer_norm2.bias"].float())
    m = _bf16(quick_gelu(m) if act == "quick_gelu" else F.gelu(m))
    m_b = _bf16(m @ _bf16(sd[pre + "mlp.fc2.weight"].float()).t() + sd[pre + "mlp.fc2.bias"].float())
    return h + m_b + _adapter_fold(sd, pre + "adapt_mlp_2.", m_b)


def hash_head(sd, hash_features: torch.Tensor) -> torch.Tensor:
    """models/arch/coop.py:544-559: (hf + hash_pe) @ hash_fc.W^T -> (B, Q*sub) concept-major -> BatchNorm1d eval."""
    B = hash_features.shape[0]
    v = (hash_features + sd["hash_pe"].float()) @ sd["hash_fc.weight"].float().t()  # (B,Q,sub)
    v = v.reshape(B, -1)
    if "hash_bn.running_mean" in sd:
        v = (v - sd["hash_bn.running_mean"].float()) / torch.sqrt(sd["hash_bn.running_var"].float() + 1e-5)
        v = v * sd["hash_bn.weight"].float() + sd["hash_bn.bias"].float()
    return v


def center_logits(sd, codes: torch.Tensor):
    """models/arch/coop.py:573-580."""
    nbit = codes.shape[1]
    c = projected_center(sd)
    cl2 = F.normalize(c, dim=-1, p=2)
    vl2 = F.normalize(codes, dim=-1, p=2)
    return vl2 @ cl2.t(), vl2 @ (cl2.sign() / (nbit ** 0.5)).t()


def concept_logits(sd, hash_features: torch.Tensor) -> torch.Tensor:
    """forward_concept (coop.py:269-276) with CosSim (cossim.py:37-82, group=1, input_group=1) -> (Q,B,C)."""
    B, Q, D = hash_features.shape
    x = (hash_features + sd["concept_pe"].float()).reshape(B * Q, D)
    cen = sd["concept_ce.centroids"].float()
    logits = F.normalize(x, p=2, dim=-1) @ F.normalize(cen, p=2, dim=-1).t()
    return logits.reshape(B, Q, -1).transpose(0, 1)


@torch.no_grad()
def encode(sd: Dict[str, torch.Tensor], images: torch.Tensor, heads: int, upt_heads: int = 8,
           act: str = "quick_gelu", emulate_bf16: bool = False, stages: Optional[dict] = None,
           with_pooled: bool = True, emulate_fold: bool = False) -> dict:
    """LGHWithFixedPrompt.forward (models/arch/coop.py:524-598), eval mode.

    Returns dict(codes (B,nbit) fp32 pre-sign, hash_features (B,Q,D), logits_cont, logits_bin (B,C),
    logits_concept (Q,B,C), image_features (B,P) [pooled branch, coop.py:498-501]).
    """
    R = _R(emulate_bf16)
    dims = infer_dims(sd)
    Q = dims["Q"]
    ctx = concept_tokens(sd, upt_heads)  # (1,Q,D)
    x = embeddings(sd, images, R)  # (B,1+Np,D)
    x = torch.cat([x, ctx.expand(x.shape[0], -1, -1)], dim=1)  # concept tokens appended AFTER pos-embed
    h = layer_norm(x, sd[VM + "pre_layrnorm.weight"].float(), sd[VM + "pre_layrnorm.bias"].float())
    if stages is not None:
        stages["concept_tokens"] = ctx.clone()
        stages["h0"] = h.clone()
    if emulate_fold and not (emulate_bf16 and dims["b"] > 0):
        raise ValueError("emulate_fold restates the default HIP chain: needs emulate_bf16=True and a model with adapters")
    for i in range(dims["L"]):
        if emulate_fold:
            h = encoder_layer_fold(sd, i, h, heads, act)
            if stages is not None:
                stages[f"h{i + 1}"] = h.clone()
            continue
        h, probs = encoder_layer(sd, i, h, heads, R, act, want_probs=stages is not None)
        if stages is not None:
            stages[f"h{i + 1}"] = h.clone()
            stages[f"attn{i}"] = probs.clone()
    hf = h[:, -Q:, :]  # raw hidden state, no post-LN (use_before_projection=True, hash_head=Identity)
    codes = hash_head(sd, hf)
    lc, lb = center_logits(sd, codes)
    out = dict(codes=codes, hash_features=hf, logits_cont=lc, logits_bin=lb)
    if "concept_ce.centroids" in sd:
        out["logits_concept"] = concept_logits(sd, hf)
    if with_pooled and (VM + "post_layernorm.weight") in sd and "backbone.visual_projection.weight" in sd:
        pooled = layer_norm(h[:, 0, :], sd[VM + "post_layernorm.weight"].float(),
                            sd[VM + "post_layernorm.bias"].float())
        out["image_features"] = pooled @ sd["backbone.visual_projection.weight"].float().t()
    return out


# synthetic workloads live in concepthash_amd/synthetic.py (bench.py must not reach into oracle/ for its inputs); re-exported
# here for the tests' convenience
from concepthash_amd.synthetic import CONFIGS, synthetic_images, synthetic_state_dict  # noqa: E402,F401
