#!/usr/bin/env python3
"""The reference's TRAINING command, end to end, on a CUB-200-sized set of JPEG files (reference README.md:7-9, main_v2.py:17-19,
experiments/train_helper.py:47-304): `python main_v2.py exp=hashing dataset=cub200 optim=sgd ...` for a few epochs, once with the GPU
data path (`dataset.gpu_decode=true`: the workers' RandomResizedCrop / RandomHorizontalFlip draws, GPU decode + crop-box resize) and once
with the reference's loader arrangement (CPU workers decode and transform) -- separate processes, the per-epoch wall clock from each
run's own `train_history.json` timestamps.

    python tools/e2e_training_demo.py [--train 5994] [--epochs 3] [--batch-size 32] [--out out.json]

Synthetic images, random-init frozen backbone: throughput, not accuracy.  (With worker processes the CPU arrangement draws its crops from
per-worker seeds, so the two runs see different crops: losses are close, not equal.)"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train", type=int, default=5994)        # CUB-200-2011 training split
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--batch-size", type=int, default=32)     # the reference's training batch size
    ap.add_argument("--backbone", default="openai/clip-vit-base-patch16")
    ap.add_argument("--modes", default="gpu_decode,cpu_loader")
    ap.add_argument("--out", default="")
    ap.add_argument("--extra", default="", help="extra command-line overrides for every run, comma separated (e.g. meter_stream=false)")
    a = ap.parse_args()
    import numpy as np
    from PIL import Image
    from concepthash_amd.hostcpu import cpu_budget
    work = tempfile.mkdtemp(prefix="ch_e2e_train_")
    try:
        data = os.path.join(work, "data", "cub200_2011")
        os.makedirs(os.path.join(data, "images"))
        n = a.train

        def make(i):
            rng = np.random.default_rng(9000 + i)
            low = rng.integers(0, 256, (12 + i % 13, 16 + i % 11, 3), dtype=np.uint8)
            h, w = (375, 500) if i % 5 else (500, 375)
            img = np.asarray(Image.fromarray(low).resize((w, h), Image.BICUBIC), dtype=np.int16)
            img = np.clip(img + rng.normal(0, 7, img.shape), 0, 255).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(data, "images", f"{i}.jpg"), "JPEG", quality=85, progressive=(i % 6 == 0))

        with ThreadPoolExecutor(max_workers=min(16, cpu_budget())) as ex:
            list(ex.map(make, range(n)))
        lines = "".join(f"images/{i}.jpg {i % 200}\n" for i in range(n))
        for name in ("train.txt", "database.txt"):
            open(os.path.join(data, name), "w").write(lines)
        open(os.path.join(data, "test.txt"), "w").write("".join(f"images/{i}.jpg {i % 200}\n" for i in range(min(n, 256))))
        env = dict(os.environ, PYTHONPATH=ROOT)
        result = {"train_images": n, "epochs": a.epochs, "batch_size": a.batch_size, "backbone": a.backbone, "host_cpu_quota": cpu_budget(), "runs": {}}
        for mode in a.modes.split(","):
            logdir = os.path.join(work, "run_" + mode)
            flags = ["dataset.gpu_decode=true"] if mode == "gpu_decode" else ["dataset.gpu_decode=false", "dataset.gpu_preprocess=false"]
            t0 = time.perf_counter()
            subprocess.run([sys.executable, os.path.join(ROOT, "main_v2.py"), "exp=hashing", "dataset=cub200", "data_dir=" + work, "optim=sgd",
                            "model.backbone.name=" + a.backbone, "model.nbit=64", f"epochs={a.epochs}", "eval_interval=0",
                            f"batch_size={a.batch_size}", "logdir=" + logdir] + flags + [x for x in a.extra.split(",") if x],
                           check=True, env=env, cwd=work)
            sec = time.perf_counter() - t0
            hist = json.load(open(os.path.join(logdir, "train_history.json")))
            ep = [round(h.get("train_seconds", 0.0), 2) for h in hist]
            result["runs"][mode] = {"wall_s": round(sec, 2), "train_loss": [round(h["train_loss"], 4) for h in hist], "epoch_s": ep,
                                    "keys": sorted(hist[0])[:40]}
            print(f"[demo] {mode}: {sec:.2f} s for the whole command ({a.epochs} epochs of {n} images); losses "
                  f"{[round(h['train_loss'], 3) for h in hist]}; epoch seconds {ep}", flush=True)
        print(json.dumps(result), flush=True)
        if a.out:
            json.dump(result, open(a.out, "w"), indent=1)
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
