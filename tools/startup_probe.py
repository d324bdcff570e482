#!/usr/bin/env python3
"""Where the one-time seconds of an evaluation command go (tools/e2e_validation_demo.py reports ~0.85 s on the first batch): import,
GPU context, engine creation, first / second encode call, decoder + pre-processing objects and their first call."""
import io, os, sys, time
t00 = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
t_imp = time.perf_counter()


def main():
    marks = [("import torch + numpy", t_imp - t00)]

    def lap(name, t0):
        torch.cuda.synchronize()
        marks.append((name, time.perf_counter() - t0))

    t0 = time.perf_counter(); torch.zeros(1, device="cuda"); lap("GPU context (first torch.zeros on cuda)", t0)
    from concepthash_amd import synthetic as syn
    cfg = syn.CONFIGS["vit_b16"]
    t0 = time.perf_counter(); sd = syn.synthetic_state_dict(cfg, nbit=64, nclass=200, seed=42); lap("synthetic state dict (host)", t0)
    from concepthash_amd.encoder import ConceptHashEncoder
    t0 = time.perf_counter(); enc = ConceptHashEncoder(sd, heads=cfg["heads"], max_batch=64, device=torch.device("cuda:0")); lap("ConceptHashEncoder(...) incl. library load", t0)
    x = syn.synthetic_images(64, cfg["image"]).to("cuda", torch.bfloat16)
    for i in range(3):
        t0 = time.perf_counter(); enc.encode(x, want=("codes", "packed")); lap(f"encode call {i}", t0)
    from PIL import Image
    from concepthash_amd.jpeg import GpuJpegDecoder
    from concepthash_amd.preprocess import GpuPreprocess
    rng = np.random.default_rng(0)
    files = []
    for i in range(64):
        bio = io.BytesIO()
        Image.fromarray(rng.integers(0, 255, (375, 500, 3), dtype=np.uint8)).save(bio, "JPEG", quality=60)
        files.append(np.frombuffer(bio.getvalue(), dtype=np.uint8))
    t0 = time.perf_counter(); dec = GpuJpegDecoder(device=torch.device("cuda:0")); pre = GpuPreprocess(256, 224, device=torch.device("cuda:0")); lap("decoder + pre-processing objects", t0)
    for i in range(3):
        t0 = time.perf_counter(); px, sizes = dec.decode(files); lap(f"decode call {i}", t0)
        t0 = time.perf_counter(); pre(px, sizes); lap(f"pre-process call {i}", t0)
    for name, sec in marks:
        print(f"{sec * 1e3:9.1f} ms  {name}")


if __name__ == "__main__":
    main()
