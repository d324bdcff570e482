#!/usr/bin/env python3
"""Where the time of the `gpu_decode` loader path goes (bench.py `loader_inclusive`): per batch, the wait for the pipeline, the host
entropy decode + uploads, the reconstruct / pre-process launches and the encode launch; cgroup throttling counters and CPU seconds per
thread class per pass.  `python tools/loader_probe.py [nimg] [--matrix] [--workers N] [--encode-constant]
[--no-thread-limit] [--pretouch] [--affinity N]` on a GPU box."""
import io, os, sys, tempfile, time, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from PIL import Image
from torch.utils.data import DataLoader
from concepthash_amd import synthetic as syn
from concepthash_amd.encoder import ConceptHashEncoder
from concepthash_amd.jpeg import GpuJpegDecoder
from concepthash_amd.preprocess import GpuPreprocess
from utils.datasets import HashingDataset, OneHot


def _cpu_stat():
    """cgroup-v2 CPU bandwidth accounting of this container: (periods, periods throttled, throttled microseconds, usage microseconds)."""
    try:
        kv = dict(line.split() for line in open("/sys/fs/cgroup/cpu.stat"))
        return tuple(int(kv.get(k, 0)) for k in ("nr_periods", "nr_throttled", "throttled_usec", "usage_usec"))
    except OSError:
        return (0, 0, 0, 0)


def _thread_cpu():
    """CPU seconds of every live thread of this process, by name, plus the process total and the reaped children's total."""
    import resource
    tick = os.sysconf("SC_CLK_TCK")
    live = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{tid}/stat").read()
            name = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            live[f"{name}:{tid}"] = (int(rest[11]) + int(rest[12])) / tick
        except OSError:
            pass
    ru, rc = resource.getrusage(resource.RUSAGE_SELF), resource.getrusage(resource.RUSAGE_CHILDREN)
    return live, ru.ru_utime + ru.ru_stime, rc.ru_utime + rc.ru_stime


def main():
    nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    if "--affinity" in sys.argv:        # confine this process (and the workers it starts) to the first N cores it may use
        ncore = int(sys.argv[sys.argv.index("--affinity") + 1])
        os.sched_setaffinity(0, sorted(os.sched_getaffinity(0))[:ncore])
    dev = torch.device("cuda:0")
    cfg = syn.CONFIGS["vit_b16"]
    enc = ConceptHashEncoder(syn.synthetic_state_dict(cfg, nbit=64, nclass=200, seed=42), heads=cfg["heads"], max_batch=256, device=dev)
    root = tempfile.mkdtemp(prefix="ch_probe_")
    os.makedirs(root + "/img")
    rng = np.random.default_rng(0)
    for i in range(nimg):
        low = rng.integers(0, 256, (12 + i % 13, 16 + i % 11, 3), dtype=np.uint8)
        img = np.asarray(Image.fromarray(low).resize((500, 375), Image.BICUBIC), dtype=np.int16)
        Image.fromarray(np.clip(img + rng.normal(0, 7, img.shape), 0, 255).astype(np.uint8)).save(f"{root}/img/{i}.jpg", "JPEG", quality=85)
    open(root + "/test.txt", "w").write("".join(f"img/{i}.jpg {i % 200}\n" for i in range(nimg)))
    pre = GpuPreprocess(256, 224, out_dtype=torch.bfloat16, device=dev)
    x_const = syn.synthetic_images(256, cfg["image"]).to(dev, torch.bfloat16)
    dec = GpuJpegDecoder(device=dev)
    ds = HashingDataset(root, "test.txt", target_transform=OneHot(200), gpu_decode=True)
    import engine
    from concepthash_amd.hostcpu import cpu_budget, limit_torch_threads
    if "--no-thread-limit" not in sys.argv:
        limit_torch_threads()
    print("cpu budget", cpu_budget(), "torch threads", torch.get_num_threads(), flush=True)
    combos = [(16, 6), (8, 6), (12, 2), (16, 2), (8, 2), (24, 3)] if "--matrix" in sys.argv else [(16, 6)]
    if "--workers" in sys.argv:            # e.g. `--workers 0` under rocprofv3 (no child processes)
        combos = [(16, int(sys.argv[sys.argv.index("--workers") + 1]))]
    for threads, nworkers in combos:
        dec.threads = threads
        dec.pretouch = "--pretouch" in sys.argv
        for rep in range(2):
            for k in ("plan_s", "ring_wait_s", "entropy_s", "enqueue_s", "touch_s"):
                dec.stats[k] = 0.0
            ds.file_workers = nworkers > 0 and ("--workers" in sys.argv or "--matrix" in sys.argv)   # DataLoader worker processes reading the files
            dl = engine.dataloader(ds, 256, shuffle=False, drop_last=False, workers=nworkers)
            if "--no-thread-limit" in sys.argv:
                torch.set_num_threads(128)
            workers, pin = dl.num_workers, False
            ctx = (f"forkserver, decode threads {threads}" if workers else f"none (in-process file reads), decode threads {threads}")
            tw = td = tp = te = 0.0
            t_first = None
            cs0 = _cpu_stat()
            th0 = _thread_cpu()
            t_all = time.perf_counter()
            from concepthash_amd.jpeg import prefetch_decoded
            it = iter(prefetch_decoded(dl, dec) if "--no-prefetch" not in sys.argv else dl)
            n = 0
            while True:
                t0 = time.perf_counter()
                try:
                    image, labels, index = next(it)
                except StopIteration:
                    break
                t1 = time.perf_counter()
                if t_first is None:
                    t_first = t1 - t_all
                if "--encode-constant" in sys.argv:      # diagnostic: the pipeline runs (reads, entropy decode, uploads) but the consumer
                    image.slot["busy"] = False           # encodes one resident batch: what do H2D traffic and the pipeline's threads cost?
                    ev = torch.cuda.Event(); ev.record(); image.slot["event"] = ev
                    t2 = time.perf_counter()
                    x = x_const
                else:
                    px, sizes = image.finish() if hasattr(image, "staged") else dec.decode(image.files)
                    t2 = time.perf_counter()
                    x = pre(px, sizes)
                t3 = time.perf_counter()
                enc.encode(x, want=("codes", "packed"))
                t4 = time.perf_counter()
                tw += t1 - t0; td += t2 - t1; tp += t3 - t2; te += t4 - t3
                n += labels.shape[0]
            torch.cuda.synchronize()
            tot = time.perf_counter() - t_all
            nb = -(-n // 256)
            cs1 = _cpu_stat()
            print(f"  cgroup cpu: {cs1[1] - cs0[1]} of {cs1[0] - cs0[0]} periods throttled, {(cs1[2] - cs0[2]) / 1e3:.0f} ms throttled, "
                  f"{(cs1[3] - cs0[3]) / 1e6 / tot:.1f} cores busy on average over {tot * 1e3:.0f} ms")
            print(f"workers {workers} pin {pin} ctx {ctx} pass {rep}: {n / tot:8.0f} images/s ({(n - 256) / max(tot - t_first, 1e-9):.0f} after the first batch, which took {t_first * 1e3:.0f} ms) | per batch ms: wait {tw / nb * 1e3:7.1f} decode {td / nb * 1e3:7.1f} "
                  f"(plan {dec.stats['plan_s'] / nb * 1e3:.1f} ring-wait {dec.stats['ring_wait_s'] / nb * 1e3:.1f} entropy {dec.stats['entropy_s'] / nb * 1e3:.1f} touch {dec.stats['touch_s'] / nb * 1e3:.1f} "
                  f"enqueue {dec.stats['enqueue_s'] / nb * 1e3:.1f}) preprocess {tp / nb * 1e3:6.1f} encode-launch {te / nb * 1e3:6.1f}", flush=True)
            del it, dl
            time.sleep(0.5)                 # let the workers exit so that RUSAGE_CHILDREN holds their time
            th1 = _thread_cpu()
            d_live = {k: v - th0[0].get(k, 0.0) for k, v in th1[0].items()}
            top = sorted(d_live.items(), key=lambda kv: -kv[1])[:8]
            self_d, child_d = th1[1] - th0[1], th1[2] - th0[2]
            print(f"  cpu seconds: process {self_d:.2f} (live threads {sum(d_live.values()):.2f}, exited threads {self_d - sum(d_live.values()):.2f}), "
                  f"reaped children {child_d:.2f}; top live threads: " + ", ".join(f"{k} {v:.2f}" for k, v in top), flush=True)
    shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":      # spawn / forkserver workers re-import this module
    main()
