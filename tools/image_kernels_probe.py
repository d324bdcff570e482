#!/usr/bin/env python3
"""The GPU half of the loader alone: 256 JPEG files of 500 x 375 (the common CUB-200 size) -> ch_jpeg_reconstruct -> ch_preprocess, a few
times, with nothing else on the GPU -- the command to put behind `rocprofv3 --kernel-trace --stats` for the per-kernel times of
`jpeg_idct`, `jpeg_color`, `resize_h*`, `resize_v*`.  Also checks one image against Pillow / the PIL chain.

    python3 tools/image_kernels_probe.py [--images 256] [--reps 12] [--train]"""
import argparse
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=256)
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--train", action="store_true", help="training geometry: a random crop box and flip per image")
    a = ap.parse_args()
    import numpy as np
    import torch
    from PIL import Image
    from concepthash_amd.jpeg import GpuJpegDecoder
    from concepthash_amd.preprocess import GpuPreprocess
    from utils import transforms as T
    files = []
    for i in range(a.images):
        rng = np.random.default_rng(i)
        low = rng.integers(0, 256, (12 + i % 13, 16 + i % 11, 3), dtype=np.uint8)
        h, w = (375, 500) if i % 5 else (500, 375)
        img = np.asarray(Image.fromarray(low).resize((w, h), Image.BICUBIC), dtype=np.int16)
        img = np.clip(img + rng.normal(0, 7, img.shape), 0, 255).astype(np.uint8)
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", quality=85)
        files.append(b.getvalue())
    dev = torch.device("cuda:0")
    dec = GpuJpegDecoder(device=dev)
    pre = GpuPreprocess(256, 224, device=dev)      # bf16 output, as the trainer uses it
    boxes = flips = None
    if a.train:
        rng = np.random.default_rng(1)
        boxes, flips = [], []
        for i in range(a.images):
            h, w = (375, 500) if i % 5 else (500, 375)
            bh, bw = int(rng.integers(h // 3, h + 1)), int(rng.integers(w // 3, w + 1))
            boxes.append((int(rng.integers(0, h - bh + 1)), int(rng.integers(0, w - bw + 1)), bh, bw))
            flips.append(bool(rng.integers(0, 2)))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for rep in range(a.reps):
        if rep == a.reps // 2:
            ev[0].record()
        pixels, sizes = dec.decode(files)
        out = pre(pixels, sizes, boxes=boxes, flips=flips)
    ev[1].record()
    torch.cuda.synchronize()
    print(f"decode + pre-process of {a.images} images: {ev[0].elapsed_time(ev[1]) / (a.reps - a.reps // 2):.3f} ms per batch "
          f"(host entropy decode included; max taps {(pre.plan(sizes) if boxes is None else pre.plan_boxes(sizes, boxes, flips))[4]})")
    i = 3
    ref = Image.open(io.BytesIO(files[i])).convert("RGB")
    if a.train:
        t, l, bh, bw = boxes[i]
        ref = ref.crop((l, t, l + bw, t + bh)).resize((224, 224), Image.BICUBIC)
        ref = ref.transpose(Image.FLIP_LEFT_RIGHT) if flips[i] else ref
        want = T.normalize_transform(3)(T.ToTensor()(ref))
    else:
        want = T.Compose([T.Resize(256, T.interpolation("bicubic")), T.CenterCrop(224), T.ToTensor(), T.normalize_transform(3)])(ref)
    print("bit-equal to the PIL chain (bf16 of it):", bool(torch.equal(out[i].cpu(), want.to(torch.bfloat16))))


if __name__ == "__main__":
    main()
