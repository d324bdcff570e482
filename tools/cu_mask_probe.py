#!/usr/bin/env python3
"""How many CUs do the HBM-bound launches of a layer need?  

The adapter up / down projections and attention hold every CU they are dispatched to while they wait on HBM; an MFMA-bound GEMM
of the other launch chain cannot share a CU with them (LDS and registers).  This probe runs them on streams created with
hipExtStreamCreateWithCUMask over the first n CUs of the mask (the driver deals mask bits round-robin over the 8 XCDs) and prints
  1. each kernel's duration vs n (alone on the chip);
  2. the makespan of an MFMA-bound GEMM (fc1 / fc2, unmasked stream) launched together with an HBM-bound one on n CUs, against the
     two launched on two unmasked streams and against their serial sum.
    python tools/cu_mask_probe.py [--rows 25728] [--cus 256,192,128,96,64,48,32]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from concepthash_amd import _lib


def hip():
    for name in ("libamdhip64.so", os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")):
        try:
            return ctypes.CDLL(name)
        except OSError:
            continue
    raise RuntimeError("libamdhip64.so not found")


def masked_stream(h, ncu, total=256):
    if ncu >= total:
        return torch.cuda.Stream()
    words = (total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(ncu):
        mask[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = h.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask({ncu}) -> {rc}")
    return torch.cuda.ExternalStream(s.value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=25728)
    ap.add_argument("--cus", default="256,192,128,96,64,48,32")
    ap.add_argument("--rounds", type=int, default=7)
    a = ap.parse_args()
    lib, h = _lib.load(), hip()
    dev = torch.device("cuda", 0)
    M = a.rows
    Mp = (M + 255) // 256 * 256
    B = M // 201
    scale = torch.tensor([0.5], device=dev)

    def gemm_case(N, K, epi):
        X = torch.randn(Mp, K, device=dev).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        out = torch.empty(Mp, N, dtype=torch.bfloat16, device=dev)
        resid = torch.zeros(Mp, N, device=dev)
        st_in = torch.rand(Mp, K // 64, 2, device=dev) * 64
        st_in[..., 1] += 64
        st_out = torch.zeros(Mp, N // 64, 2, device=dev)
        fold_c = torch.randn(N, device=dev)
        hb = torch.empty(Mp, N, dtype=torch.bfloat16, device=dev)
        addend = torch.zeros(Mp, N, dtype=torch.bfloat16, device=dev)
        keep = (X, W, bias, out, resid, st_in, st_out, fold_c, hb, addend)

        def run(stream):
            _lib.check(lib.ch_debug_gemm_ln(0, _lib.ptr(X), Mp, _lib.ptr(W), _lib.ptr(bias), M, N, K, epi, _lib.ptr(out), N,
                                            _lib.ptr(resid), N, _lib.ptr(scale), _lib.ptr(addend), _lib.ptr(st_in), _lib.ptr(fold_c),
                                            1e-5, _lib.ptr(st_out), _lib.ptr(hb), _lib.stream_ptr(stream)), "gemm")
        run.keep = keep
        return run

    qkv = torch.randn(B * 201, 2304, device=dev).to(torch.bfloat16)
    ao = torch.empty(B * 201, 768, dtype=torch.bfloat16, device=dev)

    def attention(stream):
        _lib.check(lib.ch_debug_attention(_lib.ptr(qkv), B, 201, 12, _lib.ptr(ao), _lib.stream_ptr(stream)), "attention")

    cases = {"up": gemm_case(768, 384, 7), "down": gemm_case(384, 768, 10), "attention": attention,
             "fc1": gemm_case(3072, 768, 9), "fc2": gemm_case(768, 3072, 6)}
    cus = [int(c) for c in a.cus.split(",")]
    streams = {n: masked_stream(h, n) for n in cus}
    main_s = torch.cuda.Stream()

    def timed(fn_pairs):
        """fn_pairs: [(fn, stream)] launched together; -> median makespan in us"""
        ts = []
        for r in range(a.rounds + 2):
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True)
            ends = [torch.cuda.Event(enable_timing=True) for _ in fn_pairs]
            e0.record(main_s)
            for _, s in fn_pairs:
                if s is not main_s:
                    s.wait_event(e0)
            for (fn, s), e in zip(fn_pairs, ends):
                fn(s)
                e.record(s)
            torch.cuda.synchronize()
            if r >= 2:
                ts.append(max(e0.elapsed_time(e) for e in ends) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]

    print(f"rows {M} ({B} images): duration alone vs CUs in the stream's mask (us, median of {a.rounds})")
    alone = {}
    for name, fn in cases.items():
        row = []
        for n in cus:
            alone[(name, n)] = timed([(fn, streams[n])])
            row.append(f"{n}: {alone[(name, n)]:7.1f}")
        print(f"  {name:10s} " + "  ".join(row), flush=True)
    print("makespan of an MFMA-bound GEMM (unmasked stream) launched together with an HBM-bound kernel on n CUs (us)")
    for big in ("fc1", "fc2"):
        for small in ("up", "down", "attention"):
            serial = alone[(big, cus[0])] + alone[(small, cus[0])]
            row = [f"serial {serial:7.1f}"]
            for n in cus:
                row.append(f"{n}: {timed([(cases[big], main_s), (cases[small], streams[n])]):7.1f}")
            print(f"  {big} || {small:10s} " + "  ".join(row), flush=True)


if __name__ == "__main__":
    main()
