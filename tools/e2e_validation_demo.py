#!/usr/bin/env python3
"""The reference's evaluation COMMAND, end to end, on a CUB-200-sized set of JPEG files (reference README / main_v2.py:21-41,
experiments/test_hashing.py:54-181): `python main_v2.py --config-name val.yaml logdir=<run> dataset=cub200 ...` run three times on the
same files and the same seeded checkpoint --

  * `dataset.gpu_decode=true`      workers read files; host threads Huffman-decode; GPU: IDCT / colour / resize / encode / retrieve
  * `dataset.gpu_preprocess=true`  workers decode with PIL; GPU: resize / crop / normalise / encode / retrieve
  * neither                        the reference's arrangement: workers decode and transform on the CPU; GPU: encode / retrieve

-- as separate processes, wall clock around each command (Python start-up, model build, worker start, both splits, the mAP pass, JSON
written).  The three `history.json` must agree: same codes, same mAP, bit for bit.

    python tools/e2e_validation_demo.py [--queries 5794] [--database 5994] [--backbone openai/clip-vit-base-patch16] [--out out.json]

Synthetic images (500 x 375, JPEG quality 85, one in six progressive), random-init weights: the numbers are throughput, not accuracy."""
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
    ap.add_argument("--queries", type=int, default=5794)      # CUB-200-2011 test split
    ap.add_argument("--database", type=int, default=5994)     # ... and its training split, the retrieval database
    ap.add_argument("--backbone", default="openai/clip-vit-base-patch16")
    ap.add_argument("--batch-size", type=int, default=64)     # configs/val.yaml:10
    ap.add_argument("--modes", default="gpu_decode,gpu_preprocess,cpu_loader")
    ap.add_argument("--out", default="")
    ap.add_argument("--extra", default="", help="extra command-line overrides for every run, comma separated (e.g. meter_stream=false)")
    a = ap.parse_args()
    import numpy as np
    from PIL import Image
    from concepthash_amd.hostcpu import cpu_budget

    work = tempfile.mkdtemp(prefix="ch_e2e_")
    try:
        data = os.path.join(work, "data", "cub200_2011")
        os.makedirs(os.path.join(data, "images"))
        n = a.queries + a.database

        def make(i):
            rng = np.random.default_rng(7000 + i)
            low = rng.integers(0, 256, (12 + i % 13, 16 + i % 11, 3), dtype=np.uint8)
            h, w = (375, 500) if i % 5 else (500, 375)
            img = np.asarray(Image.fromarray(low).resize((w, h), Image.BICUBIC), dtype=np.int16)
            img = np.clip(img + rng.normal(0, 7, img.shape), 0, 255).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(data, "images", f"{i}.jpg"), "JPEG", quality=85, progressive=(i % 6 == 0))

        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=min(16, cpu_budget())) as ex:
            list(ex.map(make, range(n)))
        gen_s = time.perf_counter() - t0
        for name, lo, hi in (("test.txt", 0, a.queries), ("database.txt", a.queries, n), ("train.txt", a.queries, n)):
            with open(os.path.join(data, name), "w") as f:
                f.write("".join(f"images/{i}.jpg {i % 200}\n" for i in range(lo, hi)))
        print(f"[demo] {n} JPEG files in {gen_s:.1f} s under {data}", flush=True)
        env = dict(os.environ, PYTHONPATH=ROOT)
        logdir = os.path.join(work, "run")
        common = ["dataset=cub200", "data_dir=" + work, "model.nbit=64"]   # configs: root = ${data_dir}/${dataset.data_folder} = <work>/data/cub200_2011
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synthetic_logdir.py"), logdir,
                        "model.backbone.name=" + a.backbone] + common, check=True, env=env, cwd=work)
        result = {"queries": a.queries, "database": a.database, "backbone": a.backbone, "batch_size": a.batch_size,
                  "host_cpu_quota": cpu_budget(), "generate_s": round(gen_s, 1), "runs": {}}
        hist = {}
        for mode in a.modes.split(","):
            flags = {"gpu_decode": ["dataset.gpu_decode=true"], "gpu_preprocess": ["dataset.gpu_decode=false", "dataset.gpu_preprocess=true"],
                     "cpu_loader": ["dataset.gpu_decode=false", "dataset.gpu_preprocess=false"]}[mode]
            ev = os.path.join(work, "ev_" + mode)
            t0 = time.perf_counter()
            subprocess.run([sys.executable, os.path.join(ROOT, "main_v2.py"), "--config-name", "val.yaml", "logdir=" + logdir,
                            f"batch_size={a.batch_size}", "eval_logdir=" + ev] + common + flags + [x for x in a.extra.split(",") if x],
                           check=True, env=env, cwd=work)
            sec, t_end = time.perf_counter() - t0, time.time()
            hist[mode] = json.load(open(os.path.join(ev, "history.json")))
            tm = hist[mode].get("timing_s") or {}
            if "written_unix" in tm:    # what the command spends outside its own clock: interpreter + imports before, teardown after
                tm["after_results_s"] = round(t_end - tm.pop("written_unix"), 2)
                tm["before_start_s"] = round(sec - tm["since_start"] - tm["after_results_s"], 2)
            result["runs"][mode] = {"wall_s": round(sec, 2), "images_per_s_whole_command": round(n / sec, 1), "mAP": hist[mode]["mAP"],
                                    "precisions": hist[mode].get("precisions"), "timing_s": hist[mode].get("timing_s")}
            print(f"[demo] {mode}: {sec:.2f} s for the whole command = {n / sec:.0f} images/s; mAP {hist[mode]['mAP']:.6f}; "
                  f"phases {hist[mode].get('timing_s')}", flush=True)
        modes = list(hist)
        same = all(hist[m]["mAP"] == hist[modes[0]]["mAP"] and hist[m].get("precisions") == hist[modes[0]].get("precisions")
                   and hist[m].get("recalls") == hist[modes[0]].get("recalls") for m in modes)
        result["all_runs_report_identical_metrics"] = bool(same)
        print(json.dumps(result), flush=True)
        if a.out:
            with open(a.out, "w") as f:
                json.dump(result, f, indent=1)
        if not same:
            sys.exit("the runs disagree")
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
