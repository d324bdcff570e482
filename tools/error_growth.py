#!/usr/bin/env python3
"""Per-layer error growth of the HIP encoder's residual stream (the table behind the test tolerances).

For ViT-B/16 x 12 layers (seeded synthetic weights, 2 images) print, after every layer, max-abs / RMS and RMS / RMS of
  HIP vs fp32 oracle | HIP vs emu (rounds AFTER the LayerNorm) | HIP vs emu-fold (rounds where the default chain rounds) |
  emu vs fp32 | emu-fold vs fp32 | emu vs emu-fold
so that "how far is the kernel from an oracle that rounds at the same points" can be read next to "how far are two
rounding-emulating oracles from each other".      python tools/error_growth.py [--model vit_b16] [--batch 2]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from concepthash_amd.encoder import ConceptHashEncoder
from oracle import encoder_oracle as eo   # tools/ script run by hand: the oracle is the checker here, nothing is timed


def err(a, b):
    d = (a.double() - b.double())
    rms = b.double().pow(2).mean().sqrt()
    return float(d.abs().max() / rms), float(d.pow(2).mean().sqrt() / rms)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="vit_b16")
    ap.add_argument("--batch", type=int, default=2)
    a = ap.parse_args()
    cfg = dict(eo.CONFIGS[a.model])
    sd = eo.synthetic_state_dict(cfg, nbit=64, nclass=200)
    x = eo.synthetic_images(a.batch, cfg["image"])
    dev = torch.device("cuda:0")
    enc = ConceptHashEncoder(sd, heads=cfg["heads"], max_batch=a.batch, device=dev)
    st_ref, st_emu, st_fold = {}, {}, {}
    eo.encode(sd, x, heads=cfg["heads"], with_pooled=False, stages=st_ref)
    eo.encode(sd, x, heads=cfg["heads"], with_pooled=False, stages=st_emu, emulate_bf16=True)
    eo.encode(sd, x, heads=cfg["heads"], with_pooled=False, stages=st_fold, emulate_bf16=True, emulate_fold=True)
    print(f"# {a.model} x {cfg['L']} layers, batch {a.batch}: residual stream after layer l; each cell = max-abs/RMS (RMS/RMS)")
    print("| layer | HIP vs fp32 | HIP vs emu | HIP vs emu-fold | emu vs fp32 | emu-fold vs fp32 | emu vs emu-fold |")
    print("|---|---|---|---|---|---|---|")
    for l in range(cfg["L"] + 1):
        h = enc.hidden_states(x.to(dev), l).cpu()
        r, e, f = st_ref[f"h{l}"], st_emu[f"h{l}"], st_fold[f"h{l}"]
        cells = [err(h, r), err(h, e), err(h, f), err(e, r), err(f, r), err(e, f)]
        print(f"| {l} | " + " | ".join(f"{m:.1e} ({s:.1e})" for m, s in cells) + " |")


if __name__ == "__main__":
    main()
