#!/usr/bin/env python3
"""Same-process A/B of ch_model_set_option settings on the benchmark encode step (ViT-B/16, batch 256, codes + packed): variants are
interleaved over `--rounds` rounds of `--steps` steps each; prints median / min ms per step and images/s per variant.
    python tools/encode_ab.py "wide_kernel=-1" "wide_kernel=0" "wide_kernel=0,streams=1" [--batch 256] [--rounds 7] [--steps 10]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from concepthash_amd import synthetic as syn
from concepthash_amd.encoder import ConceptHashEncoder


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+", help='comma lists of key=value options, e.g. "wide_kernel=-1,streams=1"; "" = defaults')
    ap.add_argument("--model", default="vit_b16")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    cfg = syn.CONFIGS[a.model]
    dev = torch.device("cuda:0")
    sd = syn.synthetic_state_dict(cfg, nbit=64, nclass=200, seed=42)
    enc = ConceptHashEncoder(sd, heads=cfg["heads"], max_batch=a.batch, device=dev)
    x = syn.synthetic_images(a.batch, cfg["image"], seed=42).to(dev).to(torch.bfloat16)
    defaults = {k: enc.get_option(k) for k in ("streams", "wide_kernel", "ln_fold", "prune_last", "resid_nt", "nt_out", "pp_min_k", "group_n")}
    variants = []
    for v in a.variants:
        opts = dict(defaults)
        for kv in filter(None, v.split(",")):
            k, _, val = kv.partition("=")
            opts[k.strip()] = int(val)
        variants.append((v or "(defaults)", opts))
    times = {name: [] for name, _ in variants}
    ref = None
    for r in range(a.rounds + 1):
        for name, opts in variants:
            for k, val in opts.items():
                enc.set_option(k, val)
            out = enc.encode(x, want=("codes", "packed"))
            torch.cuda.synchronize()
            if ref is None:
                ref = out["codes"].clone()
            same = bool(torch.equal(out["codes"], ref))
            t0 = time.perf_counter()
            for _ in range(a.steps):
                enc.encode(x, want=("codes", "packed"))
            torch.cuda.synchronize()
            if r:
                times[name].append((time.perf_counter() - t0) / a.steps * 1e3)
            if r == 1:
                print(f"  [{name}] codes identical to the first variant: {same}", flush=True)
    for name, _ in variants:
        t = sorted(times[name])
        med = t[len(t) // 2]
        print(f"{name:40s} median {med:7.3f} ms/step (min {t[0]:7.3f})  {a.batch / med * 1e3:8.0f} images/s", flush=True)


if __name__ == "__main__":
    main()
