#!/usr/bin/env python3
"""ConceptHash encode-and-retrieve benchmark (BASELINE.json metric), one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One STEP (per rank) = the hot path over one batch of synthetic input already resident in HBM:
  encode 256 synthetic 224x224 bf16 images with the ViT-B/16 ConceptHash model (64-bit codes, Q=4 concept tokens,
  adapters b=384)  ->  pre-sign codes + packed uint64 codes  ->  [N>1: RCCL all_gather of the packed query codes]  ->
  exact top-10 Hamming search of all N*256 queries against this rank's gallery shard (5,994 rows, the CUB-200 database
  size)  ->  [N>1: all_gather of the per-shard lists]  ->  merge to the global top-10.
Encode shards by image (no collective); retrieval shards the gallery by rows.  `value` is whole-job
images/s = N * 256 * K / max-over-ranks time.  Weak scaling: per-rank batch and per-rank gallery shard are fixed.

Also reported (outside the timed region, same process): the Hamming scan on the 1M x 128-bit synthetic gallery
(BASELINE.json config 5 size), and -- rank 0, N == 1 only -- the CPU baseline: the oracle (PyTorch fp32 restatement of
the reference forward + the C Hamming oracle) on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0       # HBM3E spec
BATCH = 256
GALLERY_ROWS = 5994         # CUB-200 database size
NBIT = 64
NCLASS = 200
TOPK = 10


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="vit_b16")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--streams", type=int, default=1, help="2: two micro-batches on two HIP streams (DESIGN.md section 3)")
    ap.add_argument("--report-two-streams", action="store_true",
                    help="also time the optional two-stream mode after the timed region (extra launches: keep it out of profiled runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hamming-scan", action="store_true")
    args = ap.parse_args()

    os.environ["CH_STREAMS"] = str(args.streams)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"[bench] note: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    local_rank = local_rank % max(1, torch.cuda.device_count())   # rehearsal: several ranks may share the one visible GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("BENCH_BACKEND", "nccl")   # nccl = RCCL over xGMI; gloo only for rehearsals on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from concepthash_amd import retrieval as rt
    from concepthash_amd import synthetic as syn
    from concepthash_amd.encoder import ConceptHashEncoder

    cfg = syn.CONFIGS[args.model]
    B = args.batch
    t_setup = time.perf_counter()
    sd = syn.synthetic_state_dict(cfg, nbit=NBIT, nclass=NCLASS, seed=42)
    log(f"[bench r{rank}] synthetic weights ready ({time.perf_counter() - t_setup:.1f} s)")
    enc = ConceptHashEncoder(sd, heads=cfg["heads"], max_batch=B, device=dev)
    log(f"[bench r{rank}] model on device: {enc.device_bytes / 2**20:.0f} MiB ({time.perf_counter() - t_setup:.1f} s)")
    images = syn.synthetic_images(B, cfg["image"], seed=42 + rank).to(dev).to(torch.bfloat16)
    g_np, gl_np = syn.synthetic_codes(GALLERY_ROWS, NBIT, seed=1234 + rank, nclass=NCLASS)
    gallery = torch.from_numpy(g_np.view(np.int64)).to(dev)
    W = gallery.shape[1]

    def step():
        out = enc.encode(images, want=("codes", "packed"))
        q = out["packed"]
        if world > 1:
            allq = torch.empty(world * B, W, dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(allq, q)
            q = allq
        idx, dst = rt.hamming_topk(q, gallery, TOPK, g_index_base=rank * GALLERY_ROWS)
        if world > 1:
            nq = q.shape[0]        # output = concatenation along dim 0 (the form every backend accepts), viewed per shard
            li = torch.empty(world * nq, TOPK, dtype=torch.int64, device=dev)
            ld = torch.empty(world * nq, TOPK, dtype=torch.int32, device=dev)
            dist.all_gather_into_tensor(li, idx)
            dist.all_gather_into_tensor(ld, dst)
            idx, dst = rt.topk_merge(li.view(world, nq, TOPK), ld.view(world, nq, TOPK))
        return out["codes"], idx, dst

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    log(f"[bench r{rank}] warm-up done ({time.perf_counter() - t_setup:.1f} s)")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    enc.profile_begin(args.steps * (enc.launches_per_encode + 2) + 8)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        codes, idx, dst = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = enc.profile_end()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(codes).all()
    log(f"[bench r{rank}] timed region: {args.steps} steps in {elapsed:.3f} s")

    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        gemm_cats = [c for c in prof if c.startswith("gemm_")]
        gemm_ms = sum(prof[c]["ms"] for c in gemm_cats)
        gemm_flops = sum(prof[c]["flops"] for c in gemm_cats)
        gemm_launches = sum(prof[c]["launches"] for c in gemm_cats)
        achieved_all = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        total_ms = sum(p["ms"] for p in prof.values())
        traffic_tab = {}
        tpath = os.path.join(ROOT, "profiles", "gemm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic_tab = json.load(open(tpath)).get("per_kernel", {})
            except Exception:
                traffic_tab = {}
        # kernel instances as rocprofv3 names them (default chain, LayerNorm folded): (label, rocprof name, launch categories)
        rows_tok = B * enc.ntok
        D_, b_pad = enc.cfg["dim"], (enc.cfg["adapter_dim"] + 127) // 128 * 128
        up_bytes = rows_tok * (D_ * 4 * 2 + D_ * 2 * 2 + b_pad * 2) + D_ * b_pad * 2      # fp32 RMW + bf16 addend + bf16 copy + X, W
        down_bytes = rows_tok * (D_ * 2 + b_pad * 2) + D_ * b_pad * 2
        instances = [
            ("gemm_pp_kernel<EPI_BIAS_STATS> (out_proj + fc2, 256x256 ping-pong)", "gemm_pp_kernel<6, 0>", ["gemm_out", "gemm_fc2"], None),
            ("gemm_pp_kernel<EPI_FOLD_QUICKGELU> (fc1)", "gemm_pp_kernel<9, 0>", ["gemm_fc1"], None),
            ("gemm_pp_kernel<EPI_FOLD_BIAS> (qkv)", "gemm_pp_kernel<8, 0>", ["gemm_qkv"], None),
            ("gemm_bf16_kernel<EPI_SCALE_RESID_STATS> (adapter up, 128x128)", "gemm_bf16_kernel<7>", ["gemm_up"], up_bytes),
            ("gemm_bf16_kernel<EPI_FOLD_GELU> (adapter down, 128x128)", "gemm_bf16_kernel<10>", ["gemm_down"], down_bytes),
        ]
        per_kernel = []
        for label, rname, cats, hbm_bytes in instances:
            ms = sum(prof[c]["ms"] for c in cats if c in prof)
            n = sum(prof[c]["launches"] for c in cats if c in prof)
            fl = sum(prof[c]["flops"] for c in cats if c in prof)
            if n == 0 or ms <= 0:
                continue
            tf = fl / (ms * 1e-3) / 1e12
            row = {"kernel": label, "rocprof_name": rname, "bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_BF16_TFLOPS,
                   "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4), "avg_launch_us": round(ms * 1e3 / n, 2),
                   "launches_per_step": n // max(1, args.steps), "ms_per_step": round(ms / args.steps, 3),
                   "algorithmic_gflop_per_launch": round(fl / n / 1e9, 2),
                   "traffic": traffic_tab.get(rname, {}).get("hbm_bytes_per_launch")}
            if hbm_bytes is not None:   # short-K adapter GEMMs: priced against HBM (algorithmic bytes per launch / duration)
                gbs = hbm_bytes / (ms * 1e-3 / n) / 1e9
                row.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                            "frac": round(gbs / PEAK_HBM_GBS, 4), "algorithmic_bytes_per_launch": int(hbm_bytes),
                            "tflops": round(tf, 2)})
            per_kernel.append(row)
        dominant = max(per_kernel, key=lambda r: r["ms_per_step"]) if per_kernel else None
        result = {
            "metric": "images/s encode (ViT+hash) + Hamming top-10 retrieval, CUB-200 64-bit",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"CUB-200 64-bit concept_hash, {args.model} bf16 (CLIP-structured, Q=4 concept tokens, "
                                   f"adapters b=384), batch={B}/GPU, top-{TOPK} Hamming vs {GALLERY_ROWS}-row gallery shard/GPU",
                       "per_gpu_batch": B, "global_batch": world * B, "gallery_rows_per_gpu": GALLERY_ROWS,
                       "hip_streams": args.streams,
                       "parallelism": f"images x{world} (no collective), gallery rows x{world} (RCCL all_gather of packed "
                                      f"queries + lists)" if world > 1 else "single GPU"},
            # dominant kernel = the instance with the most time per step (first row of profiles/*_kernel_stats.csv)
            "roofline": ({k: dominant[k] for k in ("bound", "kernel", "rocprof_name", "achieved", "peak", "unit", "frac", "traffic",
                                                   "avg_launch_us", "launches_per_step", "algorithmic_gflop_per_launch")}
                         if dominant else None),
            "roofline_per_kernel": per_kernel,
            "roofline_all_gemm_launches": {"bound": "mfma", "achieved": round(achieved_all, 2), "peak": PEAK_BF16_TFLOPS,
                                           "unit": "TFLOP/s", "frac": round(achieved_all / PEAK_BF16_TFLOPS, 4),
                                           "avg_launch_us": round(gemm_ms * 1e3 / max(1, gemm_launches), 2),
                                           "launches_per_step": gemm_launches // max(1, args.steps),
                                           "algorithmic_gflop_per_step": round(gemm_flops / max(1, args.steps) / 1e9, 2),
                                           "note": "every GEMM launch of the step, HBM-bound adapter projections included"},
            "encode_tflops_end_to_end": round(enc.flops_per_image * B / (ms_per_step * 1e-3) / 1e12, 2),
            "kernel_ms_per_step": {c: round(p["ms"] / args.steps, 4) for c, p in prof.items()},
            "kernel_ms_per_step_total": round(total_ms / args.steps, 3),
        }

    # ---- PCIe-inclusive variant (outside the timed region): the same step fed from pinned host memory each time -----
    if rank == 0:
        host = images.cpu().pin_memory()
        dev_in = torch.empty_like(images)

        def step_h2d():
            dev_in.copy_(host, non_blocking=True)
            return enc.encode(dev_in, want=("codes", "packed"))

        step_h2d()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            step_h2d()
        torch.cuda.synchronize()
        pcie_s = (time.perf_counter() - t0) / 5
        # double-buffered feed: the copy of batch i+1 runs on a side stream while batch i is encoded
        bufs = [dev_in, torch.empty_like(images)]
        copy_stream = torch.cuda.Stream(device=dev)
        ready = [torch.cuda.Event(), torch.cuda.Event()]      # copy into buffer j finished
        freed = [torch.cuda.Event(), torch.cuda.Event()]      # encode of buffer j finished (it may be overwritten)
        main = torch.cuda.current_stream(dev)
        for e in freed:
            e.record(main)

        def feed(j):
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(freed[j])
                bufs[j].copy_(host, non_blocking=True)
                ready[j].record(copy_stream)

        nrep = 8
        feed(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(nrep):
            j = i & 1
            if i + 1 < nrep:
                feed(j ^ 1)
            main.wait_event(ready[j])
            enc.encode(bufs[j], want=("codes", "packed"))
            freed[j].record(main)
        torch.cuda.synchronize()
        pipe_s = (time.perf_counter() - t0) / nrep
        result["pcie_inclusive"] = {"images_per_s": round(B / pcie_s, 1), "ms_per_step": round(pcie_s * 1e3, 3),
                                    "note": f"encode only, batch copied from pinned host memory every step on the same stream "
                                            f"({host.numel() * 2 / 2**20:.0f} MiB bf16, no overlap); never used for `value`",
                                    "double_buffered": {"images_per_s": round(B / pipe_s, 1), "ms_per_step": round(pipe_s * 1e3, 3),
                                                        "note": "copy of batch i+1 on a side stream under the encode of batch i"}}
        del bufs
        del host, dev_in

    # ---- optional two-stream mode (outside the timed region): same step, two micro-batches on two HIP streams ---------
    if rank == 0 and world == 1 and args.streams == 1 and args.report_two_streams:
        os.environ["CH_STREAMS"] = "2"
        enc2 = ConceptHashEncoder(sd, heads=cfg["heads"], max_batch=B, device=dev)
        os.environ["CH_STREAMS"] = "1"
        c1 = enc.encode(images, want=("codes", "packed"))
        c2 = enc2.encode(images, want=("codes", "packed"))
        same = bool(torch.equal(c1["codes"], c2["codes"]) and torch.equal(c1["packed"], c2["packed"]))
        for _ in range(2):
            enc2.encode(images, want=("codes", "packed"))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            o2 = enc2.encode(images, want=("codes", "packed"))
            rt.hamming_topk(o2["packed"], gallery, TOPK)
        torch.cuda.synchronize()
        s2 = (time.perf_counter() - t0) / 10
        result["two_streams"] = {"images_per_s": round(B / s2, 1), "ms_per_step": round(s2 * 1e3, 3), "codes_identical": same,
                                 "note": "CH_STREAMS=2 / --streams 2: the tile-quantisation tails of one micro-batch's launches are "
                                         "filled by the other's; not the default because per-launch durations (the roofline above) "
                                         "stop describing single kernels once launches overlap; never used for `value`"}
        enc2.close()
        del enc2

    # ---- Hamming scan at BASELINE config-5 size (outside the timed region; per rank, reported by rank 0) -----------
    if rank == 0 and not args.no_hamming_scan:
        G5, Q5, NB5 = 1_000_000, 16384, 128
        g5 = torch.randint(-2 ** 63, 2 ** 63 - 1, (G5, NB5 // 64), dtype=torch.int64, device=dev)
        q5 = torch.randint(-2 ** 63, 2 ** 63 - 1, (Q5, NB5 // 64), dtype=torch.int64, device=dev)
        rt.hamming_topk(q5, g5, TOPK)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            rt.hamming_topk(q5, g5, TOPK)
        e1.record()
        torch.cuda.synchronize()
        sec = e0.elapsed_time(e1) * 1e-3 / reps
        log(f"[bench] hamming scan {Q5} x {G5} x {NB5} b: {sec * 1e3:.2f} ms")
        tq = 256
        alg_bytes = -(-Q5 // tq) * G5 * (NB5 // 64) * 8 + Q5 * ((NB5 // 64) + TOPK) * 8
        result["hamming"] = {
            "workload": f"{Q5} queries x {G5} gallery rows x {NB5} bit, exact top-{TOPK}, 1 GPU",
            "queries_per_s": round(Q5 / sec, 1), "comparisons_per_s": float(f"{Q5 * G5 / sec:.4g}"),
            "ms": round(sec * 1e3, 3),
            "roofline": {"bound": "hbm", "achieved": round(alg_bytes / sec / 1e9, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(alg_bytes / sec / 1e9 / PEAK_HBM_GBS, 5), "traffic": None,
                         "note": "algorithmic bytes = ceil(Qn/256)*G*W*8 + Qn*(W+k)*8; the scan is VALU-bound "
                                 "(xor+popcount+select), see DESIGN.md"},
            # the bound that applies (PMC-backed): 9.75 integer VALU instructions per 128-bit pair at
            # 4 cycles per wave64 instruction on 1024 SIMDs at 2.4 GHz, insertion passes not counted
            "valu_ceiling_comparisons_per_s": 4.0e12, "valu_frac": round(Q5 * G5 / sec / 4.0e12, 4),
        }
        del g5, q5
        # mAP@all + P@k/R@k at the CUB-200 size (5,794 queries x 5,994 gallery rows x 64 bit, real class-count statistics)
        qn_np, ql_np = syn.synthetic_codes(5794, NBIT, seed=77, nclass=NCLASS)
        qn = torch.from_numpy(qn_np.view(np.int64)).to(dev)
        qlab, glab = torch.from_numpy(ql_np).to(dev), torch.from_numpy(gl_np).to(dev)
        rt.evaluate(qn, gallery, qlab, glab)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ev = rt.evaluate(qn, gallery, qlab, glab)
        torch.cuda.synchronize()
        ev_s = (time.perf_counter() - t0) / 5
        result["hamming"]["map_eval"] = {"workload": "mAP@all + P/R@{1,5,10}: 5794 queries x 5994 gallery rows x 64 bit, 200 classes",
                                         "ms": round(ev_s * 1e3, 3), "queries_per_s": round(5794 / ev_s, 1),
                                         "mAP": round(ev["mAP"], 6)}

    # ---- CPU baseline: oracle on the host cores, bounded sample (rank 0, N == 1 only) ---------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import encoder_oracle as eo     # the ONLY use of oracle/ in this file: the CPU baseline being timed
        from oracle import hamming_oracle as ho
        # the box exposes all host cores to os.cpu_count() but grants a CPU share: use the affinity mask, capped at 16
        cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
        torch.set_num_threads(cores)
        bs = 8
        x = syn.synthetic_images(bs, cfg["image"], seed=42)
        eo.encode(sd, x[:1], heads=cfg["heads"], with_pooled=False)  # warm-up
        log(f"[bench] cpu baseline: {cores} threads, warm-up done")
        t0 = time.perf_counter()
        nb = 0
        while nb < BATCH // bs and (nb == 0 or time.perf_counter() - t0 < 12.0):   # 10-30 s of CPU work, at most one bench batch
            c = eo.encode(sd, x, heads=cfg["heads"], with_pooled=False)["codes"]
            pk = ho.pack(c.numpy())
            ho.topk(pk, g_np, TOPK)
            nb += 1
            log(f"[bench] cpu baseline batch {nb}: {time.perf_counter() - t0:.1f} s")
        cpu_s = time.perf_counter() - t0
        result["cpu_baseline"] = {"value": round(nb * bs / cpu_s, 2), "unit": "images/s", "cores": cores, "kind": "port",
                                  "sample": f"{nb} batches of {bs} images: oracle/encoder_oracle.py (PyTorch CPU fp32 "
                                            f"restatement of the reference forward) + oracle/hamming_oracle.c pack + "
                                            f"top-{TOPK} vs the same {GALLERY_ROWS}-row gallery; {cpu_s:.1f} s"}
        nhq = 4096
        hq, hg = syn.synthetic_codes(nhq, 128, seed=1)[0], syn.synthetic_codes(1_000_000, 128, seed=2)[0]
        log("[bench] cpu hamming baseline inputs ready")
        t0 = time.perf_counter()
        ho.bench_topk(hq, hg, TOPK)
        hs = time.perf_counter() - t0
        result["cpu_baseline_hamming"] = {"value": float(f"{nhq * 1_000_000 / hs:.4g}"), "unit": "comparisons/s", "cores": 1,
                                          "kind": "port", "sample": f"{nhq} queries x 1M x 128 bit, C oracle (popcount + "
                                                                    f"counting-sort ranking), {hs:.1f} s"}

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
