#!/usr/bin/env python3
"""Run-to-run determinism probe of the encoder under concurrent launch chains, with per-stage localisation.

40 `encode` calls of 11 images (ViT-S/16 x 2 layers, chunks of 4 = two launch chains of 2 images) compared bit for bit with the
first; then 30 one-layer runs whose workspace buffers (H, Xn, QKV, AO, A, AD, F1: ch_debug_copy_buffer) are compared stage by stage,
printing the rows / columns that differ.  This is the tool that localised the `v_pk_fma_f32 ... op_sel:[0,1,0]` quarter-wave
corruption of round 3 (profiles/r03_pk_opsel_hazard.txt) to one accumulator column x 16 rows of the LN-fold
GEMM epilogues.   CH_LIB_TAG labels the output when several builds of the library are compared.

    python tools/determinism_probe.py          (needs a GPU; imports oracle/ for the synthetic model: test infrastructure)
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
from oracle import encoder_oracle as eo
from concepthash_amd.encoder import ConceptHashEncoder
from concepthash_amd import _lib
dev = torch.device("cuda", 0)
cfg = dict(eo.CONFIGS["vit_s16"]); cfg["L"] = 2
sd = eo.synthetic_state_dict(cfg, nbit=64, nclass=10)
x = eo.synthetic_images(11, cfg["image"]).to(dev)
lib = _lib.load()
D, M = cfg["D"], cfg["M"]
rows = 4 * 201
enc = ConceptHashEncoder(sd, heads=cfg["heads"], max_batch=4, device=dev)
def bufs():
    out = {}
    for name, which, cols, dt in (("H", 0, D, torch.float32), ("Xn", 1, D, torch.bfloat16), ("QKV", 2, 3 * D, torch.bfloat16), ("AO", 3, D, torch.bfloat16),
                                  ("A", 4, D, torch.bfloat16), ("AD", 5, 128, torch.bfloat16), ("F1", 6, M, torch.bfloat16)):
        t = torch.empty(rows, cols, dtype=dt, device=dev)
        _lib.check(lib.ch_debug_copy_buffer(enc._h, which, _lib.ptr(t), t.numel() * t.element_size(), _lib.stream_ptr()), "copy")
        out[name] = t
    torch.cuda.synchronize()
    return out
ref = enc.encode(x)["codes"].clone(); torch.cuda.synchronize()
bad = 0
for it in range(40):
    c = enc.encode(x)["codes"]; torch.cuda.synchronize()
    bad += int(not torch.equal(c, ref))
print(f"lib {os.environ.get('CH_LIB_TAG')}: {bad} / 40 encode calls differ from the first", flush=True)
for layer in (1,):
    enc.hidden_states(x[:4], layer); torch.cuda.synchronize()
    b0 = bufs()
    for it in range(30):
        enc.hidden_states(x[:4], layer); torch.cuda.synchronize()
        b = bufs()
        msg = []
        for k in b0:
            d = (b[k].float() - b0[k].float()).abs()
            if float(d.max()) > 0:
                r = torch.nonzero(d.amax(dim=1) > 0).flatten()
                cc = torch.nonzero(d.amax(dim=0) > 0).flatten()
                msg.append(f"{k}: rows {int(r[0])}..{int(r[-1])} ({r.numel()}), cols {int(cc[0])}..{int(cc[-1])} ({cc.numel()}), max {float(d.max()):.2e}")
        if msg:
            print(f"  layer {layer} iter {it}: " + " | ".join(msg), flush=True)
enc.close()
