#!/usr/bin/env python3
"""Does a large pinned host -> device copy on a side stream overlap the encoder's kernels on this box?  (bench.py `pcie_inclusive`
double-buffered feed and the gpu_decode loader both rely on it.)  Times, over the same N iterations: the encode alone, the copy alone,
and both enqueued together (copy on a side stream); overlap <=> together ~= max(alone), serialisation <=> together ~= sum."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from concepthash_amd import synthetic as syn
from concepthash_amd.encoder import ConceptHashEncoder


def main():
    dev = torch.device("cuda:0")
    cfg = syn.CONFIGS["vit_b16"]
    enc = ConceptHashEncoder(syn.synthetic_state_dict(cfg, nbit=64, nclass=200, seed=42), heads=cfg["heads"], max_batch=256, device=dev)
    x = syn.synthetic_images(256, cfg["image"], seed=1).to(dev).to(torch.bfloat16)
    side = torch.cuda.Stream(device=dev)
    N = 10
    for mb in (74, 151):
        host = torch.empty(mb << 20, dtype=torch.uint8, pin_memory=True)
        dst = torch.empty(mb << 20, dtype=torch.uint8, device=dev)

        def t(fn):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(N):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / N * 1e3

        def copy_side():
            with torch.cuda.stream(side):
                dst.copy_(host, non_blocking=True)

        def copy_main():
            dst.copy_(host, non_blocking=True)

        def both():
            copy_side()
            enc.encode(x, want=("codes", "packed"))

        for streams in (2, 1):
            enc.set_option("streams", streams)
            e = t(lambda: enc.encode(x, want=("codes", "packed")))
            c = t(copy_side)
            cm = t(copy_main)
            b = t(both)
            print(f"copy {mb} MiB, encode streams {streams}: encode alone {e:6.2f} ms | copy alone (side stream) {c:5.2f} ms = {mb / 1024 / c * 1e3:5.1f} GiB/s "
                  f"(main stream {cm:5.2f}) | together {b:6.2f} ms | sum {e + c:6.2f} max {max(e, c):6.2f}", flush=True)


if __name__ == "__main__":
    main()
