#!/usr/bin/env python3
"""Training-step benchmark of the ConceptHash adapters on one MI355X: encoder forward with saved
activations + backward in the HIP library, timed with HIP events around `ch_train_forward` / `ch_train_backward`; with --full also the
whole step through the drop-in surface (model.train() forward, LGHLoss, backward, SGD step).

    python tools/train_bench.py [--config vit_b16] [--batches 32,64,128,256] [--steps 10]
Prints one JSON line per batch size.  FLOPs: forward 2 * params-touched * tokens; the backward's dgrad products equal the
forward's, the adapters' weight-gradient products add 4 N D b per layer per adapter pair, attention backward 2.5x its forward.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from concepthash_amd import synthetic
from concepthash_amd.training import TrainEngine, adapters_from_state_dict, encoder_step_flops


def _measure(eng, x, ctx, dhf, warmup, steps):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for it in range(warmup + steps):
        eng.drop_grads()
        ev[0].record()
        eng.forward(x, ctx)
        ev[1].record()
        eng.backward(dhf)
        ev[2].record()
        torch.cuda.synchronize()
        if it >= warmup:
            tf += ev[0].elapsed_time(ev[1])
            tb += ev[1].elapsed_time(ev[2])
    return tf / steps, tb / steps


def ab(a):
    cfg = synthetic.CONFIGS[a.config]
    sd = synthetic.synthetic_state_dict(cfg, nbit=64, nclass=200)
    adapters = adapters_from_state_dict(sd, cfg["L"], cfg["D"], cfg["b"])
    batches = [int(x) for x in a.batches.split(",")]
    dev = torch.device("cuda", torch.cuda.current_device())
    variants = []
    for spec in a.ab.split(";"):
        name, _, opts = spec.partition(":")
        options = {k: int(v) for k, v in (kv.split("=") for kv in opts.split(",") if kv)}
        variants.append((name, options, TrainEngine(sd, adapters, heads=cfg["heads"], max_batch=max(batches), device=dev, options=options)))
    Q, D = 4, cfg["D"]
    ctx = torch.randn(Q, D, device="cuda") * 0.02
    grads = {}
    for B in batches:
        x = synthetic.synthetic_images(B, cfg["image"]).to("cuda", torch.bfloat16)
        dhf = torch.randn(B, Q, D, device="cuda") * 0.01
        for cycle in range(a.cycles):
            for name, options, eng in variants:
                tf, tb = _measure(eng, x, ctx, dhf, a.warmup, a.steps)
                g = eng.grads.float().clone()
                rec = {"variant": name, "options": options, "batch": B, "cycle": cycle, "forward_ms": round(tf, 3), "backward_ms": round(tb, 3),
                       "step_ms": round(tf + tb, 3)}
                if g is not None:
                    base = grads.setdefault(B, g)
                    rec["grad_rel_l2_vs_first_variant"] = float((g - base).norm() / base.norm().clamp_min(1e-30))
                print(json.dumps(rec), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="vit_b16")
    ap.add_argument("--batches", default="32,64,128,256")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--full", action="store_true", help="also time the whole step through the drop-in surface: LGHWithFixedPrompt in "
                    "train mode, LGHLoss, loss.backward(), torch.optim.SGD.step() (wall clock between synchronisations)")
    ap.add_argument("--ab", default="", help="A/B of model options in one process, interleaved: 'name:key=v,key=v;name2:key=v' "
                    "(ch_model_set_option keys; an empty option list = the defaults); one engine per variant, --cycles rounds over them")
    ap.add_argument("--cycles", type=int, default=1)
    a = ap.parse_args()
    if a.ab:
        return ab(a)
    cfg = synthetic.CONFIGS[a.config]
    sd = synthetic.synthetic_state_dict(cfg, nbit=64, nclass=200)
    adapters = adapters_from_state_dict(sd, cfg["L"], cfg["D"], cfg["b"])
    batches = [int(x) for x in a.batches.split(",")]
    eng = TrainEngine(sd, adapters, heads=cfg["heads"], max_batch=max(batches), device=torch.device("cuda", torch.cuda.current_device()))
    fwd_f, bwd_f = encoder_step_flops(eng.cfg)
    Q, D = 4, cfg["D"]
    ctx = torch.randn(Q, D, device="cuda") * 0.02
    for B in batches:
        x = synthetic.synthetic_images(B, cfg["image"]).to("cuda", torch.bfloat16)
        dhf = torch.randn(B, Q, D, device="cuda") * 0.01
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        for it in range(a.warmup + a.steps):
            eng.drop_grads()      # backward overwrites the arena: no accumulation clone + add inside the timed call
            ev[0].record()
            eng.forward(x, ctx)
            ev[1].record()
            eng.backward(dhf)
            ev[2].record()
            torch.cuda.synchronize()
            if it >= a.warmup:
                tf += ev[0].elapsed_time(ev[1])
                tb += ev[1].elapsed_time(ev[2])
        tf /= a.steps
        tb /= a.steps
        print(json.dumps({"config": a.config, "batch": B, "forward_ms": round(tf, 3), "backward_ms": round(tb, 3),
                          "step_ms": round(tf + tb, 3), "images_per_s": round(B / (tf + tb) * 1e3, 1),
                          "forward_tflops": round(fwd_f * B / tf / 1e9, 1), "backward_tflops": round(bwd_f * B / tb / 1e9, 1),
                          "trainer_gib": round(eng.device_bytes / 2 ** 30, 2)}))


    if a.full:
        eng.close()
        full_step(a, cfg, sd, batches)


def full_step(a, cfg, sd, batches):
    from concepthash_amd.training import benchmark_full_step
    for B, r in benchmark_full_step(cfg, sd, batches, a.steps, a.warmup).items():
        print(json.dumps(dict(config=a.config, batch=B, **r)))


if __name__ == "__main__":
    main()
