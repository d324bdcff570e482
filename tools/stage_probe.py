#!/usr/bin/env python3
"""Stage-by-stage comparison of ONE encoder layer: every activation buffer the HIP chain leaves behind after layer 0
(QKV, AO, Xn = bf16(H) after the first adapter, F1, A = bf16 MLP output, AD = second adapter's bottleneck, H) against the
oracle that rounds at the same points (oracle/encoder_oracle.py, emulate_fold), so that a residual-stream difference can be
traced to the stage where it first appears.      python tools/stage_probe.py [--model vit_b16]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from concepthash_amd import _lib
from concepthash_amd.encoder import ConceptHashEncoder
from oracle import encoder_oracle as eo   # tools/ script run by hand: the oracle is the checker here

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="vit_b16")
a = ap.parse_args()
cfg = dict(eo.CONFIGS[a.model])
cfg["L"] = 1
sd = eo.synthetic_state_dict(cfg, nbit=64, nclass=200)
B = 2
x = eo.synthetic_images(B, cfg["image"])
dev = torch.device("cuda:0")
enc = ConceptHashEncoder(sd, heads=cfg["heads"], max_batch=B, device=dev)
lib = _lib.load()
hid = enc.hidden_states(x.to(dev), 1).cpu()
D, M, heads = cfg["D"], cfg["M"], cfg["heads"]
N = enc.ntok
rows = B * N
bpad = (cfg["b"] + 127) // 128 * 128


def buf(which, cols, dtype):
    t = torch.empty(rows, cols, dtype=dtype, device=dev)
    _lib.check(lib.ch_debug_copy_buffer(enc._h, which, _lib.ptr(t), t.numel() * t.element_size(), _lib.stream_ptr()), "copy")
    torch.cuda.synchronize()
    return t.float().cpu()


got = dict(H=buf(0, D, torch.float32), Xn=buf(1, D, torch.bfloat16), QKV=buf(2, 3 * D, torch.bfloat16),
           AO=buf(3, D, torch.bfloat16), A=buf(4, D, torch.bfloat16), AD=buf(5, max(bpad, 128), torch.bfloat16)[:, :cfg["b"]],
           F1=buf(6, M, torch.bfloat16))

# ---- the same layer with the oracle's fold-emulating arithmetic, keeping every stage ------------------------------------
VM = eo.VM
pre = VM + "encoder.layers.0."
st = {}
eo.encode(sd, x, heads=heads, with_pooled=False, stages=st, emulate_bf16=True)   # only for h0: bf16 patch-embed operands, equals the HIP tap to 2e-6
h = st["h0"]
bf = eo._bf16
hd = D // heads
xin = bf(eo.layer_norm(h, sd[pre + "layer_norm1.weight"].float(), sd[pre + "layer_norm1.bias"].float()))
qkv = [bf(xin @ bf(sd[pre + f"self_attn.{n}.weight"].float()).t() + sd[pre + f"self_attn.{n}.bias"].float()) for n in ("q_proj", "k_proj", "v_proj")]
want = {"QKV": torch.cat(qkv, -1).reshape(rows, 3 * D)}
q, k, v = [t.reshape(B, N, heads, hd).transpose(1, 2) for t in qkv]
s = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
e = torch.exp(s - s.max(dim=-1, keepdim=True).values)
o = bf(((bf(e) @ v) / e.sum(dim=-1, keepdim=True)).transpose(1, 2).reshape(B, N, D))
want["AO"] = o.reshape(rows, D)
a_b = bf(o @ bf(sd[pre + "self_attn.out_proj.weight"].float()).t() + sd[pre + "self_attn.out_proj.bias"].float())
h1 = h + a_b + eo._adapter_fold(sd, pre + "adapt_mlp_1.", a_b)
want["Xn"] = bf(h1).reshape(rows, D)
m = eo._fold_linear(bf(h1), sd[pre + "mlp.fc1.weight"].float(), sd[pre + "mlp.fc1.bias"].float(),
                    sd[pre + "layer_norm2.weight"].float(), sd[pre + "layer_norm2.bias"].float())
f1 = bf(eo.quick_gelu(m))
want["F1"] = f1.reshape(rows, M)
m_b = bf(f1 @ bf(sd[pre + "mlp.fc2.weight"].float()).t() + sd[pre + "mlp.fc2.bias"].float())
want["A"] = m_b.reshape(rows, D)
p2 = pre + "adapt_mlp_2."
down = eo._fold_linear(m_b, sd[p2 + "down_proj.weight"].float(), sd[p2 + "down_proj.bias"].float(),
                       sd[p2 + "adapter_layer_norm.weight"].float(), sd[p2 + "adapter_layer_norm.bias"].float())
want["AD"] = bf(F.gelu(down)).reshape(rows, cfg["b"])
h2 = h1 + m_b + eo._adapter_fold(sd, p2, m_b)
want["H"] = h2.reshape(rows, D)
print(f"# {a.model}, layer 0, {B} images: HIP buffer vs fold-emulating oracle, in chain order")
for key in ("QKV", "AO", "Xn", "F1", "A", "AD", "H"):
    g, w = got[key][:rows], want[key]
    d = (g - w)
    rms = w.pow(2).mean().sqrt()
    neq = float((g != w).float().mean())
    print(f"{key:4s} fraction of elements that differ {neq:.3e}; rms diff / rms {float(d.pow(2).mean().sqrt() / rms):.3e}; "
          f"max diff / rms {float(d.abs().max() / rms):.3e}")
print(f"hidden tap vs H buffer identical: {bool(torch.equal(hid.reshape(rows, D), got['H'][:rows]))}")
