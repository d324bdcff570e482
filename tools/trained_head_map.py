#!/usr/bin/env python3
"""North-star tolerance on a TRAINED head: |mAP@all(HIP encode) - mAP@all(fp32 reference arithmetic)| from the SAME checkpoint.

No trained checkpoint exists offline, so one is made here through the training path of this library: the
ConceptHash model at ViT-B/16 x 12 layers, 64 bit, on `ncls` classes of class-structured synthetic images (a seeded prototype per
class + seeded noise), the reference's `LGHLoss` with the shipped terms (models/loss/coop.py:120-189: `concept_logits`,
`cont_logits`, `bin_logits` -- the last one is the margin-cosine loss against the SIGN of the class centres, i.e. the loss's
quantisation behaviour -- scale 8, margin 0.2), SGD with momentum (configs/optim/sgd.yaml), adapters + head + concept-token
generator trainable, backbone frozen (configs/model/concept_hash_final_v1_nosa_apt.yaml).  Then, from that one checkpoint:
  * HIP: `model.eval()` forward (ch_encode) on held-out queries / gallery (same prototypes, fresh noise);
  * fp32: oracle/encoder_oracle.py (the CPU restatement pinned by the reference's own outputs) on the same images;
  * mAP@all of both code sets from the integer oracle (oracle/hamming_oracle.c), bit-flip rate, where the flipped bits sit, and the
    histogram of |fp32 code| near zero (what decides how many bits an encode error of a given size can flip).

    python tools/trained_head_map.py [--steps 300] [--batch 128] [--mix 0.5,0.87] [--per-eval 24]

This is TEST / evidence infrastructure (it imports oracle/); tests/test_parity_r3_gpu.py asserts on its numbers.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def build_model(cfg, sd, nbit, ncls, max_batch):
    from concepthash_amd import config as cfglib
    from models.arch.coop import LGHWithFixedPrompt
    from models.backbone.clip import CLIP
    dims = dict(hidden_size=cfg["D"], num_hidden_layers=cfg["L"], num_attention_heads=cfg["heads"], intermediate_size=cfg["M"],
                patch_size=cfg["patch"], image_size=cfg["image"], projection_dim=cfg["P"], hidden_act="quick_gelu")
    upt = cfglib.DictConfig(multi=True, num_heads=8, dropout=0.0, ensemble_method="concat", single_hash_fc=True, hash_pe=True)
    cd = sd["center"].shape[1]
    tp = torch.nn.Sequential(torch.nn.Linear(cd, cd), torch.nn.ReLU(), torch.nn.Linear(cd, nbit))
    model = LGHWithFixedPrompt(CLIP(dims, allow_random_init=True), nbit, ncls, 4, add_bn=True, upt_config=upt,
                               fixed_center=torch.zeros(ncls, cd), text_projection=tp, has_adapter=True,
                               adapter_bottleneck_dim=cfg["b"], concept_reg=True, max_batch=max_batch)
    model.load_state_dict(sd)
    return model


def run(dev, steps=300, batch=128, ncls=16, per_eval=24, mix=(0.5, 0.87), lr=0.02, seed=2026, config="vit_b16", nbit=64, log=print):
    from concepthash_amd import synthetic as syn
    from concepthash_amd.training import fuse_adapter_sgd
    from models.loss.coop import LGHLoss
    from oracle import encoder_oracle as eo
    from oracle import hamming_oracle as ho
    cfg = dict(syn.CONFIGS[config])
    sd = syn.synthetic_state_dict(cfg, nbit=nbit, nclass=ncls, seed=42)
    model = build_model(cfg, sd, nbit, ncls, max_batch=max(batch, 128)).to(dev)
    model.train_max_batch = batch
    crit = LGHLoss(margin=0.2, scale=8, loss_scales=dict(bin_logits=1, cont_logits=1, concept_logits=1), ncontext=4).to(dev)
    groups = [{"params": list(model.get_adapter().parameters())}, {"params": list(model.get_training_modules().parameters())}]
    model.requires_grad_(False)
    for g in groups:
        for p in g["params"]:
            p.requires_grad_(True)
    opt = fuse_adapter_sgd(torch.optim.SGD(groups, lr=lr, momentum=0.9, weight_decay=5e-4), model)
    gcpu = torch.Generator().manual_seed(seed)
    proto = torch.randn(ncls, 3, cfg["image"], cfg["image"], generator=gcpu)
    proto_d = proto.to(dev)
    ggpu = torch.Generator(device=dev).manual_seed(seed + 1)
    model.train()
    crit.train()
    t0 = time.perf_counter()
    hist = []
    for it in range(steps):
        for g in opt.param_groups:                                   # linear warm-up, then cosine decay (configs/scheduler/csw.yaml)
            g["lr"] = lr * min(1.0, (it + 1) / 20.0) * (0.5 * (1 + np.cos(np.pi * it / steps)))
        labels = torch.randint(0, ncls, (batch,), device=dev, generator=ggpu)
        noise = torch.randn(batch, 3, cfg["image"], cfg["image"], device=dev, generator=ggpu)
        x = (mix[0] * proto_d[labels] + mix[1] * noise).to(torch.bfloat16)
        opt.zero_grad()
        loss = crit(model(x)[1], labels)
        loss.backward()
        opt.step()
        if it % 50 == 0 or it == steps - 1:
            hist.append((it, float(loss.detach()), float(crit.losses["quan"])))
            log(f"  step {it:4d} loss {hist[-1][1]:.4f} quan {hist[-1][2]:.4f}")
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    # ---- one checkpoint, two evaluations ---------------------------------------------------------------------------------------
    model.eval()
    ckpt = {k: v.detach().to("cpu", torch.float32).clone() for k, v in model.state_dict().items()
            if not k.startswith(("adapter_params.", "trainable_params."))}
    n = ncls * per_eval
    labels = torch.arange(ncls).repeat_interleave(per_eval)
    noise = torch.randn(n, 3, cfg["image"], cfg["image"], generator=gcpu)
    x = (mix[0] * proto[labels] + mix[1] * noise).to(torch.bfloat16).float()
    perm = torch.randperm(n, generator=gcpu)
    x, labels = x[perm], labels[perm]
    nq = n // 4
    with torch.no_grad():
        hip = torch.cat([model(x[i:i + 128].to(dev))[1]["codes"].cpu() for i in range(0, n, 128)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = torch.cat([eo.encode(ckpt, x[i:i + 32], heads=cfg["heads"], with_pooled=False)["codes"] for i in range(0, n, 32)])
    oracle_s = time.perf_counter() - t0
    lab = labels.numpy().astype(np.int32)
    res = {}
    for name, codes in (("hip", hip), ("fp32", ref)):
        pk = ho.pack(codes.numpy())
        res[name] = float(ho.mean_ap(pk[:nq], pk[nq:], lab[:nq], lab[nq:])["mAP"])
    flips = (hip > 0) != (ref > 0)
    rms = float(ref.pow(2).mean().sqrt())
    err = (hip - ref).abs()
    bins = [0.01, 0.02, 0.05, 0.1, 0.2, 0.5]
    near = {b: float((ref.abs() < b * rms).float().mean()) for b in bins}
    out = dict(mAP_hip=res["hip"], mAP_fp32=res["fp32"], delta=abs(res["hip"] - res["fp32"]), flip_rate=float(flips.float().mean()),
               flips=int(flips.sum()), bits=int(flips.numel()), code_rms=rms, err_max_over_rms=float(err.max()) / rms,
               err_rms_over_rms=float(err.pow(2).mean().sqrt()) / rms, near_zero=near,
               flipped_max_abs_over_rms=(float(ref[flips].abs().max()) / rms if flips.any() else 0.0),
               quan_first=hist[0][2], quan_last=hist[-1][2], loss_first=hist[0][1], loss_last=hist[-1][1],
               train_seconds=train_s, oracle_seconds=oracle_s, steps=steps, batch=batch, queries=nq, gallery=n - nq)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--mix", default="0.5,0.87")
    ap.add_argument("--per-eval", type=int, default=24)
    ap.add_argument("--lr", type=float, default=0.02)
    a = ap.parse_args()
    mix = tuple(float(v) for v in a.mix.split(","))
    dev = torch.device("cuda", 0)
    r = run(dev, steps=a.steps, batch=a.batch, per_eval=a.per_eval, mix=mix, lr=a.lr)
    print(f"trained head (ViT-B/16 x 12, 64 bit, 16 classes, mix {mix}, {a.steps} steps of batch {a.batch}, {r['train_seconds']:.1f} s): "
          f"loss {r['loss_first']:.3f} -> {r['loss_last']:.3f}, quantisation term 1 - cos(code, sign(code)) {r['quan_first']:.4f} -> {r['quan_last']:.4f}")
    print(f"mAP@all  HIP {r['mAP_hip']:.6f}   fp32 oracle {r['mAP_fp32']:.6f}   |delta| {r['delta']:.2e}   ({r['queries']} queries x {r['gallery']} gallery)")
    print(f"bit flips {r['flips']} / {r['bits']} = {r['flip_rate']:.3e};  codes: max err / rms {r['err_max_over_rms']:.2e}, rms err / rms "
          f"{r['err_rms_over_rms']:.2e};  largest |fp32 code| / rms among flipped bits {r['flipped_max_abs_over_rms']:.2e}")
    print("fraction of fp32 codes with |code| < f x rms:  " + "  ".join(f"f={b}: {v:.2e}" for b, v in r["near_zero"].items()))


if __name__ == "__main__":
    main()
