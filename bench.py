#!/usr/bin/env python3
"""ConceptHash encode-and-retrieve benchmark (BASELINE.json metric), one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without WORLD_SIZE: starts its own N ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One STEP (per rank) = the hot path over one batch of synthetic input already resident in HBM:
  encode 256 synthetic 224x224 bf16 images with the ViT-B/16 ConceptHash model (64-bit codes, Q=4 concept tokens,
  adapters b=384)  ->  pre-sign codes + packed uint64 codes  ->  [N>1: RCCL all_gather of the packed query codes]  ->
  exact top-10 Hamming search of all N*256 queries against this rank's gallery shard (5,994 rows, the CUB-200 database
  size)  ->  [N>1: all_gather of the per-shard lists]  ->  merge to the global top-10.
Encode shards by image (no collective); retrieval shards the gallery by rows.  `value` is whole-job
images/s = N * 256 * K / max-over-ranks time.  Weak scaling: per-rank batch and per-rank gallery shard are fixed.

`value` is measured in the library's default launch mode: the batch as TWO micro-batches whose launch chains run on two HIP
streams (bit-identical codes; `--streams 1` = one chain).  Per-kernel durations stop describing single kernels once launches
overlap, so the `roofline` numbers come from a second, single-stream pass of the same K steps (HIP events on the launch stream
around every launch), run right after the timed region; `roofline_pass` says so and gives that pass's ms/step.

Also reported (outside the timed region, same process): the Hamming scan and mAP@all on the 1M x 128-bit synthetic gallery
(BASELINE.json config 5 size; N > 1: the gallery sharded by rows across the ranks) and at the NABirds / CUB sizes, and --
rank 0, N == 1 only -- the CPU baselines on the host cores, on bounded samples of the same workloads.
"""
from __future__ import annotations

import argparse
import io
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0       # HBM3E spec
BATCH = 256
GALLERY_ROWS = 5994         # CUB-200 database size
NBIT = 64
NCLASS = 200
TOPK = 10


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="vit_b16")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--streams", type=int, default=2,
                    help="micro-batch launch chains on separate HIP streams (library default 2; 1 = a single chain)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hamming-scan", action="store_true")
    ap.add_argument("--no-train-step", action="store_true", help="skip the training-step block (SURVEY.md section 8 row f4)")
    ap.add_argument("--no-roofline-pass", action="store_true", help="skip the single-stream profiled pass")
    ap.add_argument("--no-evaluator", action="store_true", help="skip the evaluator-inclusive block (COOPTrainer.inference_one_epoch)")
    ap.add_argument("--no-loader", action="store_true", help="skip the loader-inclusive block (JPEG files -> DataLoader -> decode -> encode)")
    ap.add_argument("--loader-images", type=int, default=8192, help="JPEG files the loader-inclusive block generates and reads")
    ap.add_argument("--encode-only", action="store_true",
                    help="nothing but the timed encode + retrieve steps (and the roofline pass unless --no-roofline-pass): no "
                         "pcie / decode / evaluator / training / Hamming-scan / CPU-baseline blocks -- what the rocprofv3 passes of "
                         "tools/profile_round.sh trace, so that every kernel row of a summary is the encoder's")
    a = ap.parse_args()
    if a.encode_only:
        a.no_cpu_baseline = a.no_hamming_scan = a.no_train_step = a.no_evaluator = a.no_loader = True
    return a


# ---------------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` with no launcher environment starts N fresh ranks BEFORE any GPU call
# ---------------------------------------------------------------------------------------------------------------------
def self_launch(args) -> int:
    import torch   # device_count() does not initialise the GPU on this image; nothing else of torch.cuda is touched here
    ndev = torch.cuda.device_count()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    if ndev < args.gpus and "BENCH_BACKEND" not in env:
        log(f"[bench] {args.gpus} ranks requested but {ndev} GPU(s) visible: REHEARSAL -- ranks share GPUs, collectives over gloo")
        env["BENCH_BACKEND"] = "gloo"
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    lines = out0.decode().splitlines()
    js = [ln for ln in lines if ln.startswith("{")]
    for ln in lines:                      # library chatter on rank 0's stdout (e.g. gloo's connection notice) goes to stderr:
        if not js or ln is not js[-1]:    # stdout carries exactly ONE line, the JSON
            log(ln)
    if js:
        print(js[-1], flush=True)
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        log(f"[bench] ranks failed: {bad}")
        return 1
    return 0


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"[bench] note: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    from concepthash_amd.hostcpu import cpu_budget, limit_torch_threads
    host_threads = limit_torch_threads()    # torch's CPU pool inside the container's CPU quota (hostcpu.py: 128 spinning OpenMP threads otherwise)
    log(f"[bench] host: cpu budget {cpu_budget()} cores, torch intra-op threads {host_threads}")
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
    enc = ConceptHashEncoder(sd, heads=cfg["heads"], max_batch=B, device=dev, options={"streams": args.streams})
    log(f"[bench r{rank}] model on device: {enc.device_bytes / 2**20:.0f} MiB ({time.perf_counter() - t_setup:.1f} s)")
    images = syn.synthetic_images(B, cfg["image"], seed=42 + rank).to(dev).to(torch.bfloat16)
    g_np, gl_np = syn.synthetic_codes(GALLERY_ROWS, NBIT, seed=1234 + rank, nclass=NCLASS)
    gallery = torch.from_numpy(g_np.view(np.int64)).to(dev)
    W = gallery.shape[1]

    def make_step(encoder, images=images):
        B = images.shape[0]

        def step():
            out = encoder.encode(images, want=("codes", "packed"))
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
        return step

    step = make_step(enc)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    log(f"[bench r{rank}] warm-up done ({time.perf_counter() - t_setup:.1f} s)")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        codes, idx, dst = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    rank_info = None
    if world > 1:
        own_elapsed = elapsed
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # what each rank ran on, from the ranks themselves: the line proves its own topology (device, architecture, bus id, timing)
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local_rank, "device": props.name, "gcn_arch": getattr(props, "gcnArchName", ""),
                "pci_bus_id": f"{getattr(props, 'pci_domain_id', 0):04x}:{getattr(props, 'pci_bus_id', 0):02x}:"
                              f"{getattr(props, 'pci_device_id', 0):02x}",
                "ms_per_step": round(own_elapsed / args.steps * 1e3, 3), "pid": os.getpid()}
        rank_info = [None] * world
        dist.all_gather_object(rank_info, mine)
    assert torch.isfinite(codes).all()
    log(f"[bench r{rank}] timed region: {args.steps} steps in {elapsed:.3f} s")

    # ---- roofline pass: the same K steps as ONE launch chain on one stream, every launch bracketed by HIP events ----------
    prof, prof_ms_per_step, codes_same = None, None, None
    if not args.no_roofline_pass:
        enc1 = enc
        enc.set_option("streams", 1)          # state of this handle, not of the process (include/concepthash_hip.h: ch_model_set_option)
        step1 = make_step(enc1)
        c1 = step1()[0]
        codes_same = bool(torch.equal(c1, codes))     # micro-batching never changes a bit
        torch.cuda.synchronize()
        enc1.profile_begin(args.steps * (enc1.launches_per_encode + 2) + 8)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step1()
        torch.cuda.synchronize()
        prof_ms_per_step = (time.perf_counter() - t0) / args.steps * 1e3
        prof = enc1.profile_end()
        enc.set_option("streams", args.streams)

    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        result = {
            "metric": "images/s encode (ViT+hash) + Hamming top-10 retrieval, CUB-200 64-bit",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"CUB-200 64-bit concept_hash, {args.model} bf16 (CLIP-structured, Q=4 concept tokens, "
                                   f"adapters b=384), batch={B}/GPU, top-{TOPK} Hamming vs {GALLERY_ROWS}-row gallery shard/GPU",
                       "per_gpu_batch": B, "global_batch": world * B, "gallery_rows_per_gpu": GALLERY_ROWS,
                       "hip_streams": args.streams,
                       "parallelism": f"images x{world} (no collective), gallery rows x{world} ({backend_name(dist)} all_gather of packed "
                                      f"queries + lists)" if world > 1 else "single GPU"},
            "encode_tflops_end_to_end": round(enc.flops_per_image * B / (ms_per_step * 1e-3) / 1e12, 2),
            "encode_mfma_frac_end_to_end": round(enc.flops_per_image * B / (ms_per_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
        }
        if world > 1:
            # self-verifying topology: the backend torch.distributed reports, and one record per rank gathered from the ranks
            result["collective_backend"] = dist.get_backend()
            result["ranks_seen"] = len({r["rank"] for r in rank_info})
            result["ranks"] = rank_info
            result["distinct_devices"] = len({(r["pci_bus_id"], r["device"]) for r in rank_info})
        if prof is not None:
            result.update(roofline_block(prof, args, enc, B, prof_ms_per_step, codes_same))

    # ---- the batch sizes the reference evaluates / trains with (configs/val.yaml:10 batch_size 64; model yaml batch 32), outside the
    # timed region: the same step (encode + pack + top-10 vs the gallery shard) at per-GPU batch 8 / 32 / 64 / 256, default options
    if rank == 0 and world == 1 and not args.encode_only:      # (its step would enter collectives the other ranks are not in)
        result["value_by_batch"] = by_batch_block(torch, enc, make_step, images, B)

    # ---- Hamming blocks (outside the timed region).  FIRST of the extra blocks: every rank takes part in its collectives, and the
    # rank-0-only blocks below would otherwise keep the other ranks waiting inside them -----------------------------------------
    if not args.no_hamming_scan:
        hb = hamming_block(torch, np, rt, syn, dist, dev, rank, world)
        if rank == 0:
            result["hamming"] = hb

    # ---- CPU baselines: oracle on the host cores, bounded samples (rank 0, N == 1 only).  The value moves between 14 and 29 images/s from
    # box to box and run to run (same code, 16 threads, no cgroup throttling in either case: where the shared host places the threads);
    # the cgroup counters of the interval are in the object.
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result.update(cpu_baselines(torch, np, syn, sd, cfg, g_np))

    # ---- PCIe-inclusive variant (outside the timed region): the same step fed from pinned host memory each time -----
    if rank == 0 and not args.encode_only:
        result["pcie_inclusive"] = pcie_block(torch, enc, images, B)

    # ---- decode-inclusive variant (outside the timed region): decoded uint8 bytes -> GPU pre-processing -> encode ------------
    if rank == 0 and not args.encode_only:
        result["decode_inclusive"] = decode_block(torch, enc, B, dev)

    # ---- loader-inclusive variants (outside the timed region): JPEG FILES on disk -> DataLoader -> decode -> pre-process -> encode ----
    if rank == 0 and not args.no_loader:
        result["loader_inclusive"] = loader_block(torch, np, enc, B, dev, args.loader_images, value=result["value"] / world)

    # ---- evaluator-inclusive variant (outside the timed region): the reference's evaluation loop around the same model ------------
    if rank == 0 and not args.no_evaluator:
        result["evaluator_inclusive"] = evaluator_block(torch, syn, sd, cfg, B, dev, value=result["value"] / world)

    # ---- training step of the adapters (outside the timed region) -------------------------------
    if rank == 0 and not args.no_train_step:
        result["train_step"] = train_block(torch, syn, sd, cfg, B, dev)

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def backend_name(dist):
    """RCCL when torch.distributed's backend is nccl (= RCCL on ROCm), otherwise the backend's own name (gloo: CPU rehearsal)."""
    b = dist.get_backend()
    return "RCCL" if b == "nccl" else str(b)


def by_batch_block(torch, enc, make_step, images, B):
    out = {}
    for b in (8, 32, 64, 256):
        if b > B:
            continue
        stp = make_step(enc, images[:b])
        for _ in range(3):
            stp()
        torch.cuda.synchronize()
        reps = 30 if b <= 64 else 10
        t0 = time.perf_counter()
        for _ in range(reps):
            stp()
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / reps
        out[f"batch_{b}"] = {"images_per_s": round(b / sec, 1), "ms_per_step": round(sec * 1e3, 3),
                             "mfma_frac": round(enc.flops_per_image * b / sec / 1e12 / PEAK_BF16_TFLOPS, 4)}
    out["note"] = ("the timed step (encode + pack + top-10 vs the gallery shard) at smaller per-GPU batches, wall clock incl. host launch "
                   "time; configs/val.yaml:10 evaluates with batch_size 64; never used for `value`")
    return out


def roofline_block(prof, args, enc, B, prof_ms_per_step, codes_same):
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
    # kernel instances as rocprofv3 names them (default chain, LayerNorm folded).  ONE problem shape per row: out_proj and fc2 run
    # the same code under two symbol names (gemm_pp.hip, TAG), and the final layer's launches on the compact head rows have
    # categories of their own ("*_pruned", not listed: < 1 % of the step).  grid = work-items, as the rocprofv3 CSVs print it.
    rows_tok = B * enc.ntok
    D_, F_, b_pad = enc.cfg["dim"], enc.cfg["ffn"], (enc.cfg["adapter_dim"] + 127) // 128 * 128
    st = rows_tok * (D_ // 64) * 8                       # one (sum, sumsq) pair per 64-column slice of a D-wide row

    def gemm_bytes(N, K, out_bytes):                     # X + W + what the epilogue reads / writes
        return rows_tok * K * 2 + N * K * 2 + out_bytes

    pp_grid = lambda N: -(-rows_tok // 256) * (N // 256) * 512
    v1_grid = lambda N: -(-rows_tok // 128) * (N // 128) * 256
    # cache-policy instances the dispatcher picks by size (gemm_bf16.hip: resid_nt_choice / out_nt_choice; CH_RESID_NT / CH_NT_OUT force)
    def _pick(key, nbytes, limit):
        opt = enc.get_option(key)          # 0 = by tensor size (gemm_bf16.hip), 1 / -1 = forced on / off
        return opt > 0 if opt else nbytes >= limit
    nt_resid = _pick("resid_nt", rows_tok * D_ * 4, 48 << 20)
    pp_tag = lambda N: 2 if _pick("nt_out", rows_tok * N * 2, 128 << 20) else 0
    instances = [  # (label, rocprof name, category, grid, algorithmic bytes per launch, bound)
        ("gemm_pp_kernel<EPI_BIAS_STATS> out_proj (K = D, 256x256 ping-pong)", "gemm_pp_kernel<6, 0, 0, 0>", "gemm_out",
         pp_grid(D_), gemm_bytes(D_, D_, rows_tok * D_ * 2 + st), "mfma"),
        ("gemm_pp_kernel<EPI_BIAS_STATS, TAG 1> fc2 (K = 4 D)", "gemm_pp_kernel<6, 0, 0, 1>", "gemm_fc2",
         pp_grid(D_), gemm_bytes(D_, F_, rows_tok * D_ * 2 + st), "mfma"),
        ("gemm_pp_kernel<EPI_FOLD_QUICKGELU> fc1", f"gemm_pp_kernel<{9 if enc.cfg['act'] == 0 else 10}, 0, 0, {pp_tag(F_)}>", "gemm_fc1",
         pp_grid(F_), gemm_bytes(F_, D_, rows_tok * F_ * 2 + st), "mfma"),
        ("gemm_pp_kernel<EPI_FOLD_BIAS> qkv (layers >= 1; layer 0 runs <EPI_BIAS>)", f"gemm_pp_kernel<8, 0, 0, {pp_tag(3 * D_)}>", "gemm_qkv",
         pp_grid(3 * D_), gemm_bytes(3 * D_, D_, rows_tok * 3 * D_ * 2 + st), "mfma"),
        ("gemm_bf16_kernel<EPI_SCALE_RESID_STATS> adapter up (128x128)", "gemm_bf16_kernel<7, true>" if nt_resid else "gemm_bf16_kernel<7, false>", "gemm_up",
         v1_grid(D_), gemm_bytes(D_, b_pad, rows_tok * D_ * (4 * 2 + 2 + 2) + st), "hbm"),   # fp32 RMW + bf16 addend + bf16 copy
        ("gemm_bf16_kernel<EPI_FOLD_GELU> adapter down (128x128)", "gemm_bf16_kernel<10, false>", "gemm_down",
         v1_grid(b_pad), gemm_bytes(b_pad, D_, rows_tok * b_pad * 2 + st), "hbm"),
    ]
    per_kernel = []
    for label, rname, cat, grid, alg_bytes, bound in instances:
        if cat not in prof or prof[cat]["launches"] == 0 or prof[cat]["ms"] <= 0:
            continue
        ms, n, fl = prof[cat]["ms"], prof[cat]["launches"], prof[cat]["flops"]
        tf = fl / (ms * 1e-3) / 1e12
        # HBM-side bytes of this (kernel, grid) from the PMC passes (tools/make_traffic_json.py); a figure below the algorithmic
        # bytes of the launch cannot be one shape's counter and is refused (a mixed-shape average did that in round 2)
        traffic = traffic_tab.get(f"{rname}@{grid}", {}).get("hbm_bytes_per_launch")
        traffic_note = None
        if traffic is not None and traffic < 0.98 * alg_bytes:
            traffic_note, traffic = f"refused: counter bytes {traffic} < algorithmic bytes {int(alg_bytes)}", None
        row = {"kernel": label, "rocprof_name": rname, "grid": grid, "bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_BF16_TFLOPS,
               "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4), "avg_launch_us": round(ms * 1e3 / n, 2),
               "launches_per_step": n // max(1, args.steps), "ms_per_step": round(ms / args.steps, 3),
               "algorithmic_gflop_per_launch": round(fl / n / 1e9, 2), "algorithmic_bytes_per_launch": int(alg_bytes),
               "traffic": traffic}
        if traffic_note:
            row["traffic_note"] = traffic_note
        if bound == "hbm":   # short-K adapter GEMMs: priced against HBM (algorithmic bytes per launch / duration)
            gbs = alg_bytes / (ms * 1e-3 / n) / 1e9
            row.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(gbs / PEAK_HBM_GBS, 4), "tflops": round(tf, 2)})
        per_kernel.append(row)
    dominant = max(per_kernel, key=lambda r: r["ms_per_step"]) if per_kernel else None
    return {
        # dominant kernel = the instance with the most time per step (first row of profiles/*_kernel_stats.csv)
        "roofline": ({k: dominant[k] for k in ("bound", "kernel", "rocprof_name", "grid", "achieved", "peak", "unit", "frac", "traffic",
                                               "avg_launch_us", "launches_per_step", "algorithmic_gflop_per_launch",
                                               "algorithmic_bytes_per_launch")}
                     if dominant else None),
        "roofline_pass": {"hip_streams": 1, "ms_per_step": round(prof_ms_per_step, 3),
                          "images_per_s": round(B / (prof_ms_per_step * 1e-3), 1), "codes_identical_to_timed_region": codes_same,
                          "note": "same K steps as ONE launch chain on one stream, HIP events around every launch; "
                                  "`value` above is the default micro-batched mode" if args.streams != 1 else
                                  "the timed region itself ran single-stream; this pass repeats it with the launch profiler on"},
        "roofline_per_kernel": per_kernel,
        "roofline_all_gemm_launches": {"bound": "mfma", "achieved": round(achieved_all, 2), "peak": PEAK_BF16_TFLOPS,
                                       "unit": "TFLOP/s", "frac": round(achieved_all / PEAK_BF16_TFLOPS, 4),
                                       "avg_launch_us": round(gemm_ms * 1e3 / max(1, gemm_launches), 2),
                                       "launches_per_step": gemm_launches // max(1, args.steps),
                                       "algorithmic_gflop_per_step": round(gemm_flops / max(1, args.steps) / 1e9, 2),
                                       "note": "every GEMM launch of the step, HBM-bound adapter projections included"},
        "kernel_ms_per_step": {c: round(p["ms"] / args.steps, 4) for c, p in prof.items()},
        "kernel_ms_per_step_total": round(total_ms / args.steps, 3),
    }


def pcie_block(torch, enc, images, B):
    dev = images.device
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
    # pipelined feed: the copy of batch i+1 runs on a side stream while batch i is encoded (three device buffers; tools/h2d_overlap_probe.py:
    # a 74 MiB pinned copy at 53 GiB/s overlaps the encode completely on this chip -- together 11.90 ms vs 11.89 alone).  The work runs
    # on an explicit stream, not the legacy null stream: events recorded on / waited for by the null stream serialised the round-3 form
    # of this loop (13.47 vs 11.79 ms per step).
    nbuf = 3
    bufs = [dev_in] + [torch.empty_like(images) for _ in range(nbuf - 1)]
    copy_stream, work_stream = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    ready = [torch.cuda.Event() for _ in range(nbuf)]      # copy into buffer j finished
    freed = [torch.cuda.Event() for _ in range(nbuf)]      # encode of buffer j finished (it may be overwritten)
    torch.cuda.synchronize()
    for e in freed:
        e.record(work_stream)

    def feed(j):
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(freed[j])
            bufs[j].copy_(host, non_blocking=True)
            ready[j].record(copy_stream)

    nrep = 10
    feed(0)
    feed(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(work_stream):
        for i in range(nrep):
            j = i % nbuf
            if i + 2 < nrep + 2:
                feed((i + 2) % nbuf)
            work_stream.wait_event(ready[j])
            enc.encode(bufs[j], want=("codes", "packed"), stream=work_stream)
            freed[j].record(work_stream)
    torch.cuda.synchronize()
    pipe_s = (time.perf_counter() - t0) / nrep
    return {"images_per_s": round(B / pcie_s, 1), "ms_per_step": round(pcie_s * 1e3, 3),
            "note": f"encode only, batch copied from pinned host memory every step on the same stream "
                    f"({host.numel() * 2 / 2**20:.0f} MiB bf16, no overlap); never used for `value`",
            "double_buffered": {"images_per_s": round(B / pipe_s, 1), "ms_per_step": round(pipe_s * 1e3, 3),
                                "note": "copies of the next batches on a side stream under the encode of batch i (three device buffers, explicit work stream)"}}


def decode_block(torch, enc, B, dev):
    """the evaluation loader's Resize(256, bicubic) -> CenterCrop(224) -> ToTensor -> normalize on the
    GPU (ch_preprocess, Pillow-exact) in front of the encoder: B decoded 500 x 375 RGB images (the common CUB-200 size) resident
    in HBM as uint8 -> bf16 NCHW batch -> codes.  JPEG decoding itself is not part of it (CPU workers)."""
    from concepthash_amd.preprocess import GpuPreprocess
    h, w = 375, 500
    gen = torch.Generator(device=dev).manual_seed(5)
    pixels = torch.randint(0, 256, (B * h * w * 3,), dtype=torch.uint8, device=dev, generator=gen)
    sizes = [(h, w)] * B
    pre = GpuPreprocess(256, 224, out_dtype=torch.bfloat16, device=dev)
    s_pre, batch = _ev_time(torch, lambda: pre(pixels, sizes), 5)
    s_all, _ = _ev_time(torch, lambda: enc.encode(pre(pixels, sizes), want=("codes", "packed")), 5)
    return {"images_per_s": round(B / s_all, 1), "ms_per_step": round(s_all * 1e3, 3), "preprocess_ms": round(s_pre * 1e3, 3),
            "preprocess_images_per_s": round(B / s_pre, 1),
            "preprocess_gbs": round((B * h * w * 3 + B * 3 * 224 * 224 * 2) / s_pre / 1e9, 1),
            "note": f"{B} decoded {w}x{h} uint8 RGB images resident in HBM -> ch_preprocess (bit-equal to the PIL chain) -> "
                    f"ch_encode; host-side descriptor planning included; never used for `value`"}


def loader_block(torch, np, enc, B, dev, nimg, value):
    """SURVEY.md section 8 f1, the real front end: CUB-sized JPEG files (500 x 375, 4:2:0, quality 85) generated here with PIL, read back
    through `engine.dataloader` worker processes, three ways:
      cpu_loader   the reference's loader (engine.py:41-54, configs/dataset/cub200.yaml:31-47): workers PIL-decode AND run the
                   Resize / CenterCrop / ToTensor / normalize chain; the fp32 batch is copied to the GPU and encoded;
      gpu_preprocess  workers PIL-decode only; ch_preprocess + ch_encode on the GPU;
      gpu_decode   workers only READ the files; host threads Huffman-decode (ch_jpeg_entropy_decode), the GPU does inverse DCT,
                   upsampling, colour conversion (ch_jpeg_reconstruct), ch_preprocess, ch_encode.
    Each mode: one warm-up pass over a quarter of the files (worker start-up, page cache), then a timed pass over all of them."""
    import shutil
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    import engine
    from concepthash_amd.jpeg import GpuJpegDecoder, prefetch_decoded
    from concepthash_amd.preprocess import GpuPreprocess
    from utils import transforms as T
    from utils.datasets import HashingDataset, OneHot
    from concepthash_amd.hostcpu import cpu_budget
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    budget = cpu_budget()
    root = tempfile.mkdtemp(prefix="ch_loader_")
    try:
        os.makedirs(os.path.join(root, "img"))
        h, w = 375, 500

        def make(i):
            rng = np.random.default_rng(1000 + i)
            low = rng.integers(0, 256, (12 + i % 13, 16 + i % 11, 3), dtype=np.uint8)          # smooth structure at a random scale ...
            img = np.asarray(Image.fromarray(low).resize((w, h), Image.BICUBIC), dtype=np.int16)
            img = np.clip(img + rng.normal(0, 7, img.shape), 0, 255).astype(np.uint8)            # ... plus sensor-like noise
            p = os.path.join(root, "img", f"{i}.jpg")
            Image.fromarray(img).save(p, "JPEG", quality=85)
            return os.path.getsize(p)

        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=min(16, budget)) as ex:
            sizes = list(ex.map(make, range(nimg)))
        gen_s = time.perf_counter() - t0
        with open(os.path.join(root, "test.txt"), "w") as f:
            f.write("".join(f"img/{i}.jpg {i % NCLASS}\n" for i in range(nimg)))
        chain = [T.Resize(256, T.interpolation("bicubic")), T.CenterCrop(224), T.ToTensor(), T.normalize_transform(3)]
        pre = GpuPreprocess(256, 224, out_dtype=torch.bfloat16, device=dev)
        dec = GpuJpegDecoder(device=dev)
        out = {"files": nimg, "image_size": [h, w], "mean_file_kb": round(float(np.mean(sizes)) / 1024, 1), "host_cores_visible": cores,
               "host_cpu_quota": budget, "host_cores_used": min(16, budget), "torch_cpu_threads": torch.get_num_threads(), "generate_s": round(gen_s, 1)}
        log(f"[bench] loader: {nimg} JPEG files ({out['mean_file_kb']} KiB mean) in {gen_s:.1f} s; {cores} host cores")

        train_chain = [T.RandomResizedCrop(224, interpolation=T.interpolation("bicubic")), T.RandomHorizontalFlip(), T.ToTensor(),
                       T.normalize_transform(3)]

        def run(mode, n, epochs, train=False):
            ds = HashingDataset(root, "test.txt", transform=train_chain if train else chain, target_transform=OneHot(NCLASS),
                                gpu_preprocess=mode == "gpu_preprocess", gpu_decode=mode == "gpu_decode")
            ds.items = ds.items[:n]
            dl = engine.dataloader(ds, B, shuffle=False, drop_last=False)
            rates = []
            for epoch in range(epochs):         # the SAME loader object iterated again, as the trainers' epoch loops do
                t0 = time.perf_counter()
                done = 0
                it = prefetch_decoded(dl, dec) if mode == "gpu_decode" else dl      # host half of the decode one or two batches ahead
                for image, labels, index in it:
                    if mode == "gpu_decode":
                        boxes, flips = image.boxes, image.flips            # training chain: the workers' draws ride along
                        image = pre(*image.finish(), boxes=boxes, flips=flips)
                    elif mode == "gpu_preprocess":
                        image = image.to(dev, non_blocking=True)
                        image = pre(image.pixels, image.sizes, boxes=image.boxes, flips=image.flips)
                    else:
                        image = image.to(dev, non_blocking=True)
                    enc.encode(image, want=("codes", "packed"))
                    done += labels.shape[0]
                torch.cuda.synchronize()
                rates.append(done / (time.perf_counter() - t0))
            workers = dl.num_workers
            del it, dl
            return rates, workers

        for mode in ("cpu_loader", "gpu_preprocess", "gpu_decode"):
            run(mode, max(B, nimg // 4), 1)
            rates, workers = run(mode, nimg, 2)
            # images_per_s: the second epoch over the same loader (what every epoch after the first costs); first_epoch adds whatever the
            # arrangement pays once -- the forkserver itself and, for gpu_decode (persistent workers), the worker start
            out[mode] = {"images_per_s": round(rates[1], 1), "first_epoch_images_per_s": round(rates[0], 1), "vs_value": round(rates[1] / value, 4),
                         "loader_workers": workers}
            if mode == "gpu_decode":
                out[mode].update(decode_threads=dec.threads, pil_fallback=dec.stats["pil_fallback"])
            log(f"[bench] loader {mode}: {rates[1]:.0f} images/s (first epoch {rates[0]:.0f})")
        # the TRAINING transforms of the same configs (RandomResizedCrop(224, bicubic) -> RandomHorizontalFlip -> ToTensor -> normalize,
        # configs/dataset/cub200.yaml:13-23): the reference's arrangement runs them on PIL images in the workers; the GPU path lets the
        # workers draw the crop box and the flip (same random calls) and resizes the box on the GPU.  Same consumer as above (ch_encode),
        # so the number is the FEED rate an arrangement can sustain; the training step itself takes `train_step` ms per batch.
        out["train_transforms"] = {}
        for mode in ("cpu_loader", "gpu_decode"):
            run(mode, max(B, nimg // 4), 1, train=True)
            rates, workers = run(mode, nimg, 2, train=True)
            out["train_transforms"][mode] = {"images_per_s": round(rates[1], 1), "first_epoch_images_per_s": round(rates[0], 1),
                                             "loader_workers": workers}
            log(f"[bench] loader (training transforms) {mode}: {rates[1]:.0f} images/s (first epoch {rates[0]:.0f})")
        # the decoder split alone, files already in memory: host entropy decode -> H2D -> reconstruct (no loader, no encode)
        files = [np.fromfile(os.path.join(root, "img", f"{i}.jpg"), dtype=np.uint8) for i in range(B)]
        dec.decode(files)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            dec.decode(files)
        torch.cuda.synchronize()
        out["gpu_decode"]["decoder_alone_images_per_s"] = round(4 * B / (time.perf_counter() - t0), 1)
        t0 = time.perf_counter()
        for f in files[:64]:
            np.asarray(Image.open(io.BytesIO(f.tobytes())).convert("RGB"))
        out["pil_decode_images_per_s_per_core"] = round(64 / (time.perf_counter() - t0), 1)
        out["note"] = ("JPEG files on local disk (page cache warm) -> engine.dataloader workers -> ... -> ch_encode(codes, packed); wall clock "
                       "of a full pass; cpu_loader is the reference's loader arrangement; never used for `value`")
        return out
    finally:
        shutil.rmtree(root, ignore_errors=True)


def evaluator_block(torch, syn, sd, cfg, B, dev, value):
    """What `python main_v2.py exp=validation` runs per split: `COOPTrainer.inference_one_epoch(datakey, return_codes=True)`
    (reference trainers/base.py:275-307 -> trainers/coop.py:73-106) -- per batch GPU pre-processing of decoded uint8 images
    (`dataset.gpu_preprocess`), `LGHWithFixedPrompt.forward` (ch_encode: codes, both centre logits, concept logits, hash features),
    `LGHLoss` for the loss meters, the accuracy meters, and at the end ONE device -> host copy of codes / labels / meter sums.
    The dataset is synthetic and already resident in HBM (8 batches of decoded 500 x 375 images), so the number is the LOOP's:
    no JPEG decoding, no host -> device copies.  Reported next to `value` (which is encode + top-k only)."""
    from concepthash_amd import config as cfglib
    from models.arch.coop import LGHWithFixedPrompt
    from models.backbone.clip import CLIP
    from models.loss.coop import LGHLoss
    from trainers.coop import COOPTrainer
    from utils.datasets import DeviceRawLoader
    dims = dict(hidden_size=cfg["D"], num_hidden_layers=cfg["L"], num_attention_heads=cfg["heads"], intermediate_size=cfg["M"],
                patch_size=cfg["patch"], image_size=cfg["image"], projection_dim=cfg["P"], hidden_act="quick_gelu")
    upt = cfglib.DictConfig(multi=True, num_heads=8, dropout=0.1, ensemble_method="concat", single_hash_fc=True, hash_pe=True)
    C, cd = sd["center"].shape
    tp = torch.nn.Sequential(torch.nn.Linear(cd, cd), torch.nn.ReLU(), torch.nn.Linear(cd, NBIT))
    model = LGHWithFixedPrompt(CLIP(dims, allow_random_init=True), NBIT, C, 4, add_bn=True, upt_config=upt, fixed_center=torch.zeros(C, cd),
                               text_projection=tp, has_adapter=True, adapter_bottleneck_dim=cfg["b"], concept_reg=True, max_batch=B)
    model.load_state_dict(sd)
    conf = cfglib.DictConfig(device=str(dev), batch_size=B, model=cfglib.DictConfig(has_adapter=True),
                             dataset=cfglib.DictConfig(multiclass=False, resize=256, crop=cfg["image"], norm=3, gpu_preprocess=True))
    tr = COOPTrainer(conf)
    tr.distributed, tr.rank, tr.world_size = False, 0, 1       # a rank-0-only block: never enter a collective (N > 1 runs)
    tr.model = model.to(dev).eval()
    tr.criterion = LGHLoss(margin=0.2, scale=8, loss_scales=dict(bin_logits=1, cont_logits=1, concept_logits=1), ncontext=4).to(dev)
    nb, h, w = 8, 375, 500
    gen = torch.Generator(device=dev).manual_seed(11)
    pixels = torch.randint(0, 256, (nb * B, h, w, 3), dtype=torch.uint8, device=dev, generator=gen)
    labels = torch.randint(0, C, (nb * B,), device=dev, generator=gen)
    tr.dataset = {"test": [0], "db": []}
    tr.dataloader = {"test": DeviceRawLoader(pixels, labels, C, B)}
    tr.inference_one_epoch("test", True)          # warm-up: engine build, allocator
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    meters, out = tr.inference_one_epoch("test", True)
    torch.cuda.synchronize()
    sec = time.perf_counter() - t0
    ips = nb * B / sec
    del tr, model
    return {"images_per_s": round(ips, 1), "ms_per_batch": round(sec / nb * 1e3, 3), "vs_value": round(ips / value, 4),
            "batches": nb, "meters": {k: round(m.avg, 5) for k, m in meters.items()}, "codes_shape": list(out["codes"].shape),
            "note": f"COOPTrainer.inference_one_epoch over {nb} x {B} decoded {w}x{h} uint8 images resident in HBM: GPU pre-process + "
                    f"model forward (codes, centre / concept logits, hash features) + LGHLoss + loss / accuracy meters kept on the "
                    f"device + one device -> host copy of codes, labels and meter sums at the end; wall clock; never used for `value`"}


def train_block(torch, syn, sd, cfg, B, dev):
    """Encoder forward (activations saved) + backward of the training step, HIP library only (ch_train_forward /
    ch_train_backward): the reference's own batch size (32, configs/model/concept_hash_final_v1_nosa_apt.yaml:76) and the bench
    batch.  The head / loss / optimizer around it are a few hundred microseconds of small launches and are not in these times."""
    from concepthash_amd.training import TrainEngine, adapters_from_state_dict, encoder_step_flops
    eng = TrainEngine(sd, adapters_from_state_dict(sd, cfg["L"], cfg["D"], cfg["b"]), heads=cfg["heads"], max_batch=B, device=dev)
    fwd_f, bwd_f = encoder_step_flops(eng.cfg)
    ctx = torch.randn(4, cfg["D"], device=dev) * 0.02
    out = {}
    for b in sorted({min(32, B), B}):
        x = syn.synthetic_images(b, cfg["image"]).to(dev, torch.bfloat16)
        dhf = torch.randn(b, 4, cfg["D"], device=dev) * 0.01
        s_f, _ = _ev_time(torch, lambda: eng.forward(x, ctx), 5)
        # drop_grads: each backward OVERWRITES the gradient arena (no accumulation clone + add inside the timed calls)
        s_all, _ = _ev_time(torch, lambda: (eng.drop_grads(), eng.forward(x, ctx), eng.backward(dhf)), 5)
        s_b = s_all - s_f
        out[f"batch_{b}"] = {"images_per_s": round(b / s_all, 1), "ms_per_step": round(s_all * 1e3, 3), "forward_ms": round(s_f * 1e3, 3),
                             "backward_ms": round(s_b * 1e3, 3), "tflops": round((fwd_f + bwd_f) * b / s_all / 1e12, 1),
                             "mfma_frac": round((fwd_f + bwd_f) * b / s_all / 1e12 / PEAK_BF16_TFLOPS, 4)}
    out["trainer_gib"] = round(eng.device_bytes / 2 ** 30, 2)
    eng.close()
    eng = None
    # the whole step through the drop-in surface (model.train() forward, LGHLoss, backward, SGD step), wall clock
    from concepthash_amd.training import benchmark_full_step
    for b, r in benchmark_full_step(cfg, sd, sorted({min(32, B), B}), steps=5, warmup=2).items():
        out[f"batch_{b}"]["full_step_ms"] = r["full_step_ms"]
        out[f"batch_{b}"]["full_step_images_per_s"] = r["images_per_s"]
    out["note"] = ("adapters + concept tokens trained, backbone frozen; bf16 operands, fp32 accumulation / residual gradient / "
                   "parameter gradients; ms_per_step = the two C-ABI calls (HIP events), full_step_ms = the whole step through the "
                   "Python surface incl. head, loss and optimizer (wall clock); never used for `value`")
    return out


def _ev_time(torch, fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps, out


def hamming_block(torch, np, rt, syn, dist, dev, rank, world):
    """BASELINE.json config 5: 16,384 queries x 1M-row x 128-bit synthetic gallery, exact top-10.  N == 1: the whole gallery on
    this GPU.  N > 1: the gallery sharded by rows (rank r holds rows [r*1M/N, (r+1)*1M/N)), every rank scans all queries
    against its shard, the per-shard lists are all-gathered (RCCL) and merged -- comparisons/s is the whole job's."""
    G5, Q5, NB5 = 1_000_000, 16384, 128
    W5 = NB5 // 64
    gen = torch.Generator(device=dev).manual_seed(99)
    q5 = torch.randint(-2 ** 63, 2 ** 63 - 1, (Q5, W5), dtype=torch.int64, device=dev, generator=gen)   # same queries on every rank
    lo, hi = rank * G5 // world, (rank + 1) * G5 // world
    gen_g = torch.Generator(device=dev).manual_seed(1000 + rank)
    g5 = torch.randint(-2 ** 63, 2 ** 63 - 1, (hi - lo, W5), dtype=torch.int64, device=dev, generator=gen_g)

    def scan():
        idx, dst = rt.hamming_topk(q5, g5, TOPK, g_index_base=lo)
        if world > 1:
            li = torch.empty(world * Q5, TOPK, dtype=torch.int64, device=dev)
            ld = torch.empty(world * Q5, TOPK, dtype=torch.int32, device=dev)
            dist.all_gather_into_tensor(li, idx)
            dist.all_gather_into_tensor(ld, dst)
            idx, dst = rt.topk_merge(li.view(world, Q5, TOPK), ld.view(world, Q5, TOPK))
        return idx, dst

    scan()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        scan()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    sec = (time.perf_counter() - t0) / reps
    if world > 1:
        t = torch.tensor([sec], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sec = float(t.item())
    if rank == 0:
        log(f"[bench] hamming scan {Q5} x {G5} x {NB5} b on {world} GPU(s): {sec * 1e3:.2f} ms")
    tq = 256
    alg_bytes = -(-Q5 // tq) * G5 * W5 * 8 + Q5 * (W5 + TOPK) * 8
    out = {
        "workload": f"{Q5} queries x {G5} gallery rows x {NB5} bit, exact top-{TOPK}, {world} GPU(s)"
                    + (f" (gallery sharded by rows, {G5 // world} rows per GPU; all_gather of the per-shard lists + merge included)"
                       if world > 1 else ""),
        "queries_per_s": round(Q5 / sec, 1), "comparisons_per_s": float(f"{Q5 * G5 / sec:.4g}"),
        "comparisons_per_s_per_gpu": float(f"{Q5 * G5 / sec / world:.4g}"), "ms": round(sec * 1e3, 3),
        "roofline": {"bound": "hbm", "achieved": round(alg_bytes / sec / 1e9, 2), "peak": PEAK_HBM_GBS * world, "unit": "GB/s",
                     "frac": round(alg_bytes / sec / 1e9 / (PEAK_HBM_GBS * world), 5), "traffic": None,
                     "note": "algorithmic bytes = ceil(Qn/256)*G*W*8 + Qn*(W+k)*8; the scan is VALU-bound "
                             "(xor+popcount+select), see DESIGN.md"},
        # the bound that applies (PMC-backed): 9.75 integer VALU instructions per 128-bit pair at
        # 4 cycles per wave64 instruction on 1024 SIMDs at 2.4 GHz, insertion passes not counted
        "valu_ceiling_comparisons_per_s": 4.0e12 * world, "valu_frac": round(Q5 * G5 / sec / (4.0e12 * world), 4),
    }
    if world > 1:
        # mAP@all + P/R@{1,5,10} of the same 1M x 128-bit problem on the row-sharded gallery: per-shard histogram pass (recording
        # the shard's relevant rows), all_gather of the histograms, global "ranked before" bases, AP terms from the shard's own
        # records, integer all_reduce -- bit-identical to the single-GPU result for any shard count (tests)
        from concepthash_amd.distributed import ShardedRetrieval
        NC5 = 200
        ql5 = torch.randint(0, NC5, (Q5,), dtype=torch.int32, device=dev, generator=gen)       # same on every rank (gen is)
        gl5 = torch.randint(0, NC5, (hi - lo,), dtype=torch.int32, device=dev, generator=gen_g)
        sr = ShardedRetrieval(g5, gl5)
        sr.evaluate(q5, ql5, R=-1, ks=(1, 5, 10))
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            ev = sr.evaluate(q5, ql5, R=-1, ks=(1, 5, 10))
        torch.cuda.synchronize()
        dist.barrier()
        t = torch.tensor([(time.perf_counter() - t0) / 3], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        msec = float(t.item())
        if rank == 0:
            out["map_eval_1m_sharded"] = {
                "workload": f"mAP@all + P/R@{{1,5,10}}, {Q5} queries x {G5} gallery rows x {NB5} bit, {NC5} classes, gallery sharded "
                            f"by rows over {world} GPUs ({G5 // world} rows each)",
                "ms": round(msec * 1e3, 3), "queries_per_s": round(Q5 / msec, 1), "mAP": round(float(ev["mAP"]), 6),
                "comparisons_per_s": float(f"{Q5 * G5 / msec:.4g}"),
                "note": "per-shard recording scan + all_gather of the per-shard histograms + global prefix + record walk + integer "
                        "all_reduce; max over ranks"}
        return out if rank == 0 else None
    # ---- single GPU only: mAP@all (one-scan and two-scan forms, per-pass times) at three sizes ---------------------------
    def map_eval(name, qn, gn, nbit, ncls, reps):
        Wm = nbit // 64
        gg = torch.Generator(device=dev).manual_seed(7)
        gq = torch.randint(-2 ** 63, 2 ** 63 - 1, (qn, Wm), dtype=torch.int64, device=dev, generator=gg)
        ga = torch.randint(-2 ** 63, 2 ** 63 - 1, (gn, Wm), dtype=torch.int64, device=dev, generator=gg)
        ql = torch.randint(0, ncls, (qn,), dtype=torch.int32, device=dev, generator=gg)
        gl = torch.randint(0, ncls, (gn,), dtype=torch.int32, device=dev, generator=gg)
        seg = rt.map_seg_rows(qn, gn, Wm)
        s_h, hist = _ev_time(torch, lambda: rt.hamming_hist(gq, ga, ql, gl, 0, seg), reps)
        s_p, (base, _) = _ev_time(torch, lambda: rt.hist_prefix(hist), reps)
        limits, _ = rt.normalize_limits([-1, 1, 5, 10])
        s_a, _ = _ev_time(torch, lambda: rt.hamming_ap_multi(gq, ga, ql, gl, 0, seg, base, limits), reps)
        s_hr, (_, recs) = _ev_time(torch, lambda: rt.hamming_hist_rec(gq, ga, ql, gl, 0, seg), reps)
        s_ar, _ = _ev_time(torch, lambda: rt.hamming_ap_rec(gq, ga, ql, gl, 0, seg, base, recs, limits), reps)
        s_e2, _ = _ev_time(torch, lambda: rt.evaluate(gq, ga, ql, gl, R=-1, ks=(1, 5, 10), records=False), max(1, reps // 2))
        s_e, ev = _ev_time(torch, lambda: rt.evaluate(gq, ga, ql, gl, R=-1, ks=(1, 5, 10)), max(1, reps // 2))
        pairs = qn * gn
        return {"workload": f"{name}: mAP@all + P/R@{{1,5,10}}, {qn} queries x {gn} gallery rows x {nbit} bit, {ncls} classes",
                "ms": round(s_e * 1e3, 3), "queries_per_s": round(qn / s_e, 1), "mAP": round(ev["mAP"], 6),
                # `ms` = evaluate() in its default one-scan form: histogram pass that records the relevant rows + prefix + AP terms
                # from the records; the two-scan form (histogram pass, prefix, AP pass over the gallery again) beside it
                "one_scan": {"hist_and_records_ms": round(s_hr * 1e3, 3), "ap_from_records_ms": round(s_ar * 1e3, 3),
                             "record_list_entries": int(recs[1]), "workgroups_overflowed": int(recs[3].sum())},
                "two_scan_ms": round(s_e2 * 1e3, 3),
                "hist_pass_ms": round(s_h * 1e3, 3), "hist_prefix_ms": round(s_p * 1e3, 3), "ap_pass_ms": round(s_a * 1e3, 3),
                "hist_pass_comparisons_per_s": float(f"{pairs / s_h:.4g}"), "ap_pass_comparisons_per_s": float(f"{pairs / s_a:.4g}"),
                # VALU ceilings: instructions per pair of the row loop at 4 cycles per wave64 instruction
                "valu_frac_hist_pass": round(pairs / s_h / MAP_VALU_CEILING[nbit][0], 4),
                "valu_frac_ap_pass": round(pairs / s_a / MAP_VALU_CEILING[nbit][1], 4)}

    del g5
    out["map_eval"] = map_eval("CUB-200 size", 5794, 5994, 64, 200, 6)
    out["map_eval_nabirds"] = map_eval("NABirds size", 24633, 23929, 64, 555, 4)
    out["map_eval_1m"] = map_eval("BASELINE config 5 size", 16384, 1_000_000, 128, 200, 2)
    return out


# comparisons/s ceilings of the mAP row loops: (histogram pass, AP pass) at 4 cycles per wave64 VALU instruction on 1024 SIMDs
# at 2.4 GHz = 6.14e11 wave-instructions/s, 64 pairs per wave-instruction slot: 3.93e13 / (VALU instructions per row of the bare
# row loop: 8 at 64 bit, 12 at 128 bit -- per 64-bit word 2 x (xor with the DPP-broadcast gallery word + chained bcnt), plus label
# xor, compare, increment select, counter address; ISA listing, SQ_INSTS_VALU in profiles/r02_hamming_1m_pmc_*.txt;
# the AP pass's relevant-row work is not counted, so its fraction is against the same bare-loop ceiling)
MAP_VALU_CEILING = {64: (3.93e13 / 8.0, 3.93e13 / 8.0), 128: (3.93e13 / 12.0, 3.93e13 / 12.0)}


def cpu_baselines(torch, np, syn, sd, cfg, g_np):
    from oracle import encoder_oracle as eo     # the ONLY use of oracle/ in this file: the CPU baselines being timed
    from oracle import hamming_oracle as ho
    out = {}
    # the box exposes all host cores to os.cpu_count() but grants a CPU quota: use that (hostcpu.cpu_budget), capped at 16
    from concepthash_amd.hostcpu import cpu_budget
    cores = min(16, cpu_budget())
    torch.set_num_threads(cores)
    bs = 8
    x = syn.synthetic_images(bs, cfg["image"], seed=42)
    eo.encode(sd, x[:1], heads=cfg["heads"], with_pooled=False)  # warm-up
    log(f"[bench] cpu baseline: {cores} threads, warm-up done")

    def cpu_stat():     # cgroup-v2 bandwidth accounting of the container: a throttled baseline says so in the line
        try:
            kv = dict(line.split() for line in open("/sys/fs/cgroup/cpu.stat"))
            return int(kv.get("nr_periods", 0)), int(kv.get("nr_throttled", 0)), int(kv.get("usage_usec", 0))
        except OSError:
            return 0, 0, 0
    cs0 = cpu_stat()
    t0 = time.perf_counter()
    nb = 0
    while nb < BATCH // bs and (nb == 0 or time.perf_counter() - t0 < 12.0):   # 10-30 s of CPU work, at most one bench batch
        c = eo.encode(sd, x, heads=cfg["heads"], with_pooled=False)["codes"]
        pk = ho.pack(c.numpy())
        ho.topk(pk, g_np, TOPK)
        nb += 1
    cpu_s = time.perf_counter() - t0
    cs1 = cpu_stat()
    log(f"[bench] cpu baseline: {nb} batches in {cpu_s:.1f} s; cgroup: {cs1[1] - cs0[1]} of {cs1[0] - cs0[0]} periods throttled, "
        f"{(cs1[2] - cs0[2]) / 1e6 / max(cpu_s, 1e-9):.1f} cores busy on average")
    out["cpu_baseline"] = {"value": round(nb * bs / cpu_s, 2), "unit": "images/s", "cores": cores, "kind": "port",
                           "sample": f"{nb} batches of {bs} images: oracle/encoder_oracle.py (PyTorch CPU fp32 "
                                     f"restatement of the reference forward) + oracle/hamming_oracle.c pack + "
                                     f"top-{TOPK} vs the same {GALLERY_ROWS}-row gallery; {cpu_s:.1f} s",
                           "cgroup_periods_throttled": [cs1[1] - cs0[1], cs1[0] - cs0[0]],
                           "cores_busy_avg": round((cs1[2] - cs0[2]) / 1e6 / max(cpu_s, 1e-9), 1)}
    # ---- Hamming, SURVEY.md section 8(d): reference-style float matmul + topk on all granted cores, a packed numpy
    # XOR/popcount, and the C oracle; 3 warm-up + 5 timed iterations each, median
    G, nbit = 1_000_000, 128
    hg = syn.synthetic_codes(G, nbit, seed=2)[0]

    def median_time(fn, warm=3, timed=5):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(timed):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    nq_f = 128
    hq = syn.synthetic_codes(nq_f, nbit, seed=1)[0]
    bits_g = torch.from_numpy(np.unpackbits(hg.view(np.uint8), axis=1, bitorder="little").astype(np.float32) * 2 - 1)   # +-1 [G, nbit]
    bits_q = torch.from_numpy(np.unpackbits(hq.view(np.uint8), axis=1, bitorder="little").astype(np.float32) * 2 - 1)

    def ref_style():   # get_hd formula (trainers/orthohash.py:263-264, un-normalised) + topk, as the reference-style path ranks
        d = 0.5 * (nbit - bits_q @ bits_g.t())
        return torch.topk(d, TOPK, dim=1, largest=False)

    t_f = median_time(ref_style)
    log(f"[bench] cpu hamming: float matmul + topk {t_f:.2f} s per {nq_f} queries")

    def np_popcount():
        d = np.zeros((nq_f, G), dtype=np.uint8)
        for w in range(hg.shape[1]):
            d += np.bitwise_count(hq[:, w][:, None] ^ hg[:, w][None, :])
        return np.argpartition(d, TOPK, axis=1)[:, :TOPK]

    t_n = median_time(np_popcount, warm=1, timed=3)
    log(f"[bench] cpu hamming: numpy popcount + argpartition {t_n:.2f} s per {nq_f} queries")
    nq_c = 1024
    hq_c = syn.synthetic_codes(nq_c, nbit, seed=1)[0]
    t_c = median_time(lambda: ho.bench_topk(hq_c, hg, TOPK), warm=1, timed=3)
    out["cpu_baseline_hamming"] = {
        "value": float(f"{nq_f * G / t_f:.4g}"), "unit": "comparisons/s", "cores": cores, "kind": "port",
        "sample": f"{nq_f} queries x 1M x {nbit} bit: reference-style 0.5*(nbit - sign(q) sign(g)^T) float32 matmul + torch.topk "
                  f"(k = {TOPK}) on {cores} threads, 3 warm-up + 5 timed, median {t_f:.2f} s",
        "numpy_packed_popcount": {"value": float(f"{nq_f * G / t_n:.4g}"), "unit": "comparisons/s", "cores": 1,
                                  "sample": f"{nq_f} queries x 1M x {nbit} bit: numpy xor + bitwise_count + argpartition, "
                                            f"1 warm-up + 3 timed, median {t_n:.2f} s"},
        "c_oracle": {"value": float(f"{nq_c * G / t_c:.4g}"), "unit": "comparisons/s", "cores": 1,
                     "sample": f"{nq_c} queries x 1M x {nbit} bit: oracle/hamming_oracle.c (popcount + counting-sort ranking), "
                               f"1 warm-up + 3 timed, median {t_c:.2f} s"}}
    return out


if __name__ == "__main__":
    main()
