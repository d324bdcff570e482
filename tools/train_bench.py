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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="vit_b16")
    ap.add_argument("--batches", default="32,64,128,256")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--full", action="store_true", help="also time the whole step through the drop-in surface: LGHWithFixedPrompt in "
                    "train mode, LGHLoss, loss.backward(), torch.optim.SGD.step() (wall clock between synchronisations)")
    a = ap.parse_args()
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
    """The reference's train_one_batch (trainers/coop.py:107-131) on this path: concept-token generator, head, loss and optimizer on
    torch around the HIP encoder.  Wall clock per step, synchronised once per step."""
    import time

    from concepthash_amd import config as cfglib
    from models.arch.coop import LGHWithFixedPrompt
    from models.backbone.clip import CLIP
    from models.loss.coop import LGHLoss
    dims = dict(hidden_size=cfg["D"], num_hidden_layers=cfg["L"], num_attention_heads=cfg["heads"], intermediate_size=cfg["M"],
                patch_size=cfg["patch"], image_size=cfg["image"], projection_dim=cfg["P"], hidden_act="quick_gelu")
    upt = cfglib.DictConfig(multi=True, num_heads=8, dropout=0.1, ensemble_method="concat", single_hash_fc=True, hash_pe=True)
    C, cd = sd["center"].shape
    tp = torch.nn.Sequential(torch.nn.Linear(cd, cd), torch.nn.ReLU(), torch.nn.Linear(cd, 64))
    model = LGHWithFixedPrompt(CLIP(dims, allow_random_init=True), 64, C, 4, add_bn=True, upt_config=upt, fixed_center=torch.zeros(C, cd),
                               text_projection=tp, has_adapter=True, adapter_bottleneck_dim=cfg["b"], concept_reg=True)
    model.load_state_dict(sd)
    model = model.cuda().train()
    model.train_max_batch = max(batches)
    crit = LGHLoss(margin=0.2, scale=8, loss_scales=dict(bin_logits=1, cont_logits=1, concept_logits=1), ncontext=4)
    params = list(model.get_adapter().parameters()) + list(model.get_training_modules().parameters())
    model.requires_grad_(False)
    for p in params:
        p.requires_grad_(True)
    from concepthash_amd.training import fuse_adapter_sgd
    groups = [{"params": list(model.get_adapter().parameters())}, {"params": list(model.get_training_modules().parameters())}]
    # as trainers/base.py builds it: torch.optim.SGD with the adapters' group updated by one launch over the arena (ch_sgd_step)
    opt = fuse_adapter_sgd(torch.optim.SGD(groups, lr=1e-3, momentum=0.9, weight_decay=5e-4), model)
    for B in batches:
        x = synthetic.synthetic_images(B, cfg["image"]).to("cuda", torch.bfloat16)
        y = torch.randint(0, C, (B,), device="cuda")
        for it in range(a.warmup + a.steps):
            if it == a.warmup:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            opt.zero_grad()
            loss = crit(model(x)[1], y)
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        print(json.dumps({"config": a.config, "batch": B, "full_step_ms": round(ms, 3), "images_per_s": round(B / ms * 1e3, 1),
                          "fused_adapter_sgd_steps": opt.fused_adapter_steps["steps"],
                          "what": "model.train() forward + LGHLoss + backward + SGD step, wall clock"}))


if __name__ == "__main__":
    main()
