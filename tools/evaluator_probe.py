#!/usr/bin/env python3
"""Where a batch of the evaluator loop (COOPTrainer.inference_one_epoch, bench.py `evaluator_inclusive`) spends its time:
the same 8 x 256 decoded images through (a) pre-process + ch_encode (codes, packed), (b) pre-process + the model's eval forward
(all of its outputs), (c) + LGHLoss + accuracies + device meters (= inference_one_batch), (d) inference_one_epoch itself.
    python tools/evaluator_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from concepthash_amd import synthetic as syn


def main():
    dev = torch.device("cuda", 0)
    cfg = syn.CONFIGS["vit_b16"]
    sd = syn.synthetic_state_dict(cfg, nbit=64, nclass=200, seed=42)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    from concepthash_amd import config as cfglib
    from concepthash_amd.preprocess import GpuPreprocess
    from models.arch.coop import LGHWithFixedPrompt
    from models.backbone.clip import CLIP
    from models.loss.coop import LGHLoss
    from trainers.coop import COOPTrainer
    from utils.datasets import DeviceRawLoader
    from utils.misc import DeviceMeters
    dims = dict(hidden_size=cfg["D"], num_hidden_layers=cfg["L"], num_attention_heads=cfg["heads"], intermediate_size=cfg["M"],
                patch_size=cfg["patch"], image_size=cfg["image"], projection_dim=cfg["P"], hidden_act="quick_gelu")
    upt = cfglib.DictConfig(multi=True, num_heads=8, dropout=0.1, ensemble_method="concat", single_hash_fc=True, hash_pe=True)
    C, cd = sd["center"].shape
    tp = torch.nn.Sequential(torch.nn.Linear(cd, cd), torch.nn.ReLU(), torch.nn.Linear(cd, 64))
    model = LGHWithFixedPrompt(CLIP(dims, allow_random_init=True), 64, C, 4, add_bn=True, upt_config=upt, fixed_center=torch.zeros(C, cd),
                               text_projection=tp, has_adapter=True, adapter_bottleneck_dim=cfg["b"], concept_reg=True, max_batch=B)
    model.load_state_dict(sd)
    conf = cfglib.DictConfig(device=str(dev), batch_size=B, model=cfglib.DictConfig(has_adapter=True),
                             dataset=cfglib.DictConfig(multiclass=False, resize=256, crop=224, norm=3, gpu_preprocess=True))
    tr = COOPTrainer(conf)
    tr.distributed = False
    tr.model = model.to(dev).eval()
    tr.criterion = LGHLoss(margin=0.2, scale=8, loss_scales=dict(bin_logits=1, cont_logits=1, concept_logits=1), ncontext=4).to(dev)
    nb, h, w = 8, 375, 500
    gen = torch.Generator(device=dev).manual_seed(11)
    pixels = torch.randint(0, 256, (nb * B, h, w, 3), dtype=torch.uint8, device=dev, generator=gen)
    labels = torch.randint(0, C, (nb * B,), device=dev, generator=gen)
    loader = DeviceRawLoader(pixels, labels, C, B)
    tr.dataset = {"test": [0], "db": []}
    tr.dataloader = {"test": loader}
    pre = GpuPreprocess(256, 224, device=dev)
    eng = model._ensure_engine(dev, 224)

    def timeit(name, fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps / nb * 1e3
        print(f"{name:70s} {ms:8.3f} ms per batch   {B / ms * 1e3:9.1f} images/s")

    def a():
        for raw, t, i in loader:
            eng.encode(pre(raw.pixels, raw.sizes), want=("codes", "packed"))

    def a0():
        x = pre(pixels[:B].reshape(-1), [(h, w)] * B)
        for _ in range(nb):
            eng.encode(x, want=("codes", "packed"))

    def b():
        with torch.no_grad():
            for raw, t, i in loader:
                model(pre(raw.pixels, raw.sizes))

    def c():
        m = DeviceMeters(dev)
        for data in loader:
            tr.inference_one_batch(data, m)

    def host_only():            # host time of one batch's Python, GPU idle: how close the loop is to being host-bound
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = DeviceMeters(dev)
        for data in loader:
            tr.inference_one_batch(data, m)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"{'host time to ENQUEUE one batch of (c) (GPU running behind)':70s} {(t1 - t0) / nb * 1e3:8.3f} ms per batch")

    timeit("encode only, input resident (what bench.py `value` times, minus top-k)", a0)
    timeit("(a) GPU pre-process + ch_encode(codes, packed)", a)
    timeit("(b) GPU pre-process + model.eval() forward (all outputs)", b)
    timeit("(c) inference_one_batch: (b) + LGHLoss + accuracies + device meters", c)
    timeit("(d) inference_one_epoch(return_codes=True)", lambda: tr.inference_one_epoch("test", True))
    for flag in (False, True, False, True):          # same-process A/B of the meters' side stream (config key `meter_stream`)
        tr.config["meter_stream"] = flag
        timeit(f"(d) inference_one_epoch, meter_stream={flag}", lambda: tr.inference_one_epoch("test", True))
    host_only()


if __name__ == "__main__":
    main()
