#!/usr/bin/env python3
"""Stand-alone probe of the packed-fp32 op_sel form of DESIGN.md section 3.10 (tools/repro/pk_opsel_victim.hip).

A light victim kernel evaluates `v_pk_fma_f32 D, acc, ms, C op_sel:[0,1,0]` (form 1) or the operand-swapped `op_sel:[1,0,0]` spelling
(form 2) on operands read the way the LN-fold epilogue reads them and compares every result bit for bit with scalar FMAs.  It runs
  (a) alone on the chip,
  (b) on one stream while the encoder (the library's launch chains) runs on another,
  (c) next to single kernels of the library (GEMMs by shape, attention) and of torch,
  (d) next to synthetic co-tenants of ONE instruction class each (MFMA only, LDS-DMA only, ds_read only, packed VALU only, ...).
Prints mismatching lane-iterations per configuration.  Result on MI355X (profiles/r03_pk_opsel_hazard.txt): the SRC1-op_sel spellings of
v_pk_fma / v_pk_mul / v_pk_add_f32 return a wrong LOW lane only when a wave of another kernel executes MFMAs on the same SIMD.
    python tools/pk_opsel_repro.py [--seconds 1.0] [--form N | --form -1] [--synthetic]
"""
import argparse
import ctypes
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from concepthash_amd import _lib
from concepthash_amd import synthetic as syn
from concepthash_amd.encoder import ConceptHashEncoder


def build_victim():
    src = os.path.join(ROOT, "tools", "repro", "pk_opsel_victim.hip")
    out = "/tmp/libpkvictim.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", src, "-o", out], check=True)
    lib = ctypes.CDLL(out)
    lib.pk_victim_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.pk_cotenant_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.pk_class_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return lib


def measure(form, cotenant_mode=1, seconds=0.5, blocks=2048, iters=20000, vic=None):
    """Mismatching lane-iterations of one spelling (pk_opsel_victim.hip: FORM) next to the synthetic co-tenant `cotenant_mode`
    (0 = alone, 1 = MFMA only, ...) -> (mismatches, low-lane mismatches, lane-iterations).  Used by tests/test_isa_forms_gpu.py."""
    dev = torch.device("cuda", 0)
    vic = vic or build_victim()
    s_vic, s_agg = torch.cuda.Stream(), torch.cuda.Stream()
    src, dst = torch.randn(1 << 22, device=dev), torch.empty(1 << 22, device=dev)
    mism = torch.zeros(1, dtype=torch.int64, device=dev)
    low = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    t0, launches = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        assert vic.pk_victim_launch(form, blocks, iters, mism.data_ptr(), low.data_ptr(), s_vic.cuda_stream) == 0
        launches += 1
        if cotenant_mode:
            assert vic.pk_cotenant_launch(cotenant_mode, 1024, 20000, src.data_ptr(), dst.data_ptr(), src.numel(), s_agg.cuda_stream) == 0
        if launches % 4 == 0:
            s_vic.synchronize()
    torch.cuda.synchronize()
    checking = 0.5 if form > 10 else 1.0                     # MIX forms: half the waves check, the other half issue MFMAs
    return int(mism.item()), int(low.item()), int(launches * blocks * 256 * iters * checking)


CLASSES = {1: "v_pk_fma_f32 (no op_sel)", 2: "v_pk_mul_f32 op_sel_hi:[1,0] + v_pk_add_f32 op_sel_hi:[0,1]", 3: "v_cvt_pk_bf16_f32", 4: "v_exp_f32 + v_rcp_f32",
           5: "DPP: v_mov_b32_dpp quad_perm + v_xor_b32_dpp row_newbcast", 6: "v_bcnt_u32_b32 chain", 7: "ds_write_b64 + ds_read_b128 round trip",
           8: "integer VALU (add, shift-or, min / max select)", 9: "scalar v_fma_f32", 10: "POSITIVE CONTROL v_pk_fma_f32 op_sel:[0,1,0]"}


def class_exactness(op, repeats=6, blocks=2048, iters=20000, vic=None):
    """Checksums of one instruction class computed by the even waves, with the odd waves idle vs issuing MFMAs (class_kernel): same inputs,
    same instruction stream -> (lanes whose checksum differs in ANY of `repeats` MFMA-side launches, checking lanes, lane-iterations)."""
    dev = torch.device("cuda", 0)
    vic = vic or build_victim()
    s = torch.cuda.current_stream()
    ref = torch.zeros(blocks * 256, dtype=torch.int64, device=dev)
    assert vic.pk_class_launch(op, 0, blocks, iters, ref.data_ptr(), s.cuda_stream) == 0
    torch.cuda.synchronize()
    again = torch.zeros_like(ref)
    assert vic.pk_class_launch(op, 0, blocks, iters, again.data_ptr(), s.cuda_stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(ref, again), "the quiet launch is not reproducible"
    differing = torch.zeros(blocks * 256, dtype=torch.bool, device=dev)
    for _ in range(repeats):
        out = torch.zeros_like(ref)
        assert vic.pk_class_launch(op, 1, blocks, iters, out.data_ptr(), s.cuda_stream) == 0
        torch.cuda.synchronize()
        differing |= out != ref
    lanes = blocks * 128
    return int(differing.sum().item()), lanes, lanes * iters * repeats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--blocks", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=20000)
    ap.add_argument("--form", type=int, default=0, help="1..8: only that spelling (+10: with MFMA-issuing odd waves in the same workgroups); -1: all (default: 1 and 2)")
    ap.add_argument("--synthetic", action="store_true", help="only the synthetic single-instruction-class co-tenants")
    ap.add_argument("--classes", action="store_true", help="exactness of other instruction classes next to MFMA-issuing waves (checksum comparison)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    vic = build_victim()
    if a.classes:
        for op, label in CLASSES.items():
            bad, lanes, total = class_exactness(op, vic=vic)
            print(f"class {op:2d} | {label:62s} | {bad:8d} of {lanes} lanes differ with MFMA-issuing odd waves ({total:.2e} lane-iterations)", flush=True)
        return
    lib = _lib.load()
    cfg = syn.CONFIGS["vit_b16"]
    sd = syn.synthetic_state_dict(cfg, nbit=64, nclass=200, seed=42)
    images = syn.synthetic_images(256, cfg["image"], seed=42).to(dev).to(torch.bfloat16)
    s_vic, s_agg = torch.cuda.Stream(), torch.cuda.Stream()

    def run(form, aggressor, label):
        mism = torch.zeros(1, dtype=torch.int64, device=dev)
        first = torch.zeros(1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        launches = 0
        while time.perf_counter() - t0 < a.seconds:
            rc = vic.pk_victim_launch(form, a.blocks, a.iters, mism.data_ptr(), first.data_ptr(), s_vic.cuda_stream)
            assert rc == 0, rc
            launches += 1
            if aggressor is not None:
                with torch.cuda.stream(s_agg):
                    aggressor()
            if launches % 4 == 0:
                s_vic.synchronize()
        torch.cuda.synchronize()
        total = launches * a.blocks * 256 * a.iters
        print(f"form {form} | {label:44s} | {launches:4d} victim launches, {int(mism.item()):10d} mismatching lane-iterations of {total:.2e} "
              f"(low lane wrong in {int(first.item())})", flush=True)
        return int(mism.item())

    encs = {}
    for chains in (1, 2):
        os.environ["CH_STREAMS"] = str(chains)
        encs[chains] = ConceptHashEncoder(sd, heads=cfg["heads"], max_batch=256, device=dev)
    small_cfg = dict(syn.CONFIGS["vit_s16"])
    sd_s = syn.synthetic_state_dict(small_cfg, nbit=64, nclass=10, seed=1)
    os.environ["CH_STREAMS"] = "2"
    enc_small = ConceptHashEncoder(sd_s, heads=small_cfg["heads"], max_batch=4, device=dev)
    x_small = syn.synthetic_images(4, small_cfg["image"], seed=2).to(dev)

    aggressors = [(None, "alone"),
                  (lambda: encs[1].encode(images, want=("codes",)), "encoder ViT-B/16 x 256 images, one chain"),
                  (lambda: encs[2].encode(images, want=("codes",)), "encoder ViT-B/16 x 256 images, two chains"),
                  (lambda: [enc_small.encode(x_small, want=("codes",)) for _ in range(8)], "encoder ViT-S/16 x 4 images (128x128 GEMMs), two chains")]
    # ---- single-kernel co-tenants: which instruction mix does it take?
    M = 25728
    Mp = (M + 255) // 256 * 256
    scale = torch.tensor([0.5], device=dev)

    def gemm(N, K, epi, variant):
        X = torch.randn(Mp, K, device=dev).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        out = torch.empty(Mp, N, dtype=torch.bfloat16, device=dev)
        resid = torch.zeros(Mp, N, device=dev)
        addend = torch.zeros(Mp, N, dtype=torch.bfloat16, device=dev)
        keep = (X, W, bias, out, resid, addend)

        def fn():
            _lib.check(lib.ch_debug_gemm(variant, _lib.ptr(X), Mp, _lib.ptr(W), _lib.ptr(bias), M, N, K, epi, _lib.ptr(out), N, _lib.ptr(resid), N,
                                         _lib.ptr(scale), _lib.ptr(addend) if epi == 4 else None, _lib.stream_ptr(s_agg)), "gemm")
        fn.keep = keep
        return fn

    qkv = torch.randn(128 * 201, 2304, device=dev).to(torch.bfloat16)
    ao = torch.empty(128 * 201, 768, dtype=torch.bfloat16, device=dev)
    big = torch.randn(64 * 1024 * 1024, device=dev)
    ta, tb = torch.randn(8192, 4096, device=dev).to(torch.bfloat16), torch.randn(4096, 4096, device=dev).to(torch.bfloat16)
    from concepthash_amd import retrieval as rt
    import numpy as np
    g_np, _ = syn.synthetic_codes(200000, 128, seed=5, nclass=10)
    q_np, _ = syn.synthetic_codes(4096, 128, seed=6, nclass=10)
    gq, gg = torch.from_numpy(q_np.view(np.int64)).to(dev), torch.from_numpy(g_np.view(np.int64)).to(dev)
    singles = [(lambda: big.mul_(1.0001).add_(0.5), "elementwise fp32 multiply-add (no MFMA, HBM-bound)"),
               (lambda: rt.hamming_topk(gq, gg, 10), "Hamming top-k scan (no MFMA, VALU / DPP-bound)"),
               (lambda: torch.matmul(ta, tb), "torch.matmul bf16 8192x4096x4096 (vendor MFMA kernel)"),
               (gemm(3072, 768, 1, 2), "256x256 GEMM of this library, fc1 shape"),
               (gemm(768, 384, 4, 1), "128x128 GEMM of this library, adapter up shape"),
               (lambda: _lib.check(lib.ch_debug_attention(_lib.ptr(qkv), 128, 201, 12, _lib.ptr(ao), _lib.stream_ptr(s_agg)), "att"), "attention kernel (MFMA + ds_read_tr)")]
    # ---- synthetic co-tenants: one instruction class each (tools/repro/pk_opsel_victim.hip: cotenant_kernel)
    syn_src, syn_dst = torch.randn(32 * 1024 * 1024, device=dev), torch.empty(32 * 1024 * 1024, device=dev)
    names = {1: "MFMA only (v_mfma_f32_16x16x32_bf16, register operands)", 2: "LDS-DMA only (global_load_lds_dwordx4)", 3: "ds_read_b128 only",
             4: "packed-fp32 VALU only", 5: "global loads to VGPRs + stores", 6: "MFMA fed by ds_read_b128", 7: "s_barrier loop", 8: "ds_write_b64 + ds_read_b128"}
    synth = []
    for mode, label in names.items():
        def fn(mode=mode):
            rc = vic.pk_cotenant_launch(mode, 1024, 4000 if mode in (2, 5) else 20000, syn_src.data_ptr(), syn_dst.data_ptr(), syn_src.numel(), s_agg.cuda_stream)
            assert rc == 0, rc
        synth.append((fn, "synthetic: " + label))
    forms = (1, 2) if not a.form else ((a.form,) if a.form > 0 else tuple(range(1, 9)))
    todo = aggressors + singles + synth if not a.synthetic else [aggressors[0]] + synth
    if a.form < 0:
        todo = [aggressors[0], synth[0]]          # all spellings: alone and next to the MFMA-only co-tenant
        forms = tuple(range(1, 9)) + tuple(range(11, 19))   # ... and with MFMA-issuing waves inside the victim's own workgroups
    for form in forms:
        for fn, label in todo:
            run(form, fn, label)


if __name__ == "__main__":
    main()
