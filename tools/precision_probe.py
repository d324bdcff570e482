#!/usr/bin/env python3
"""How exact are the individual kernels, given EXACTLY the operands they are given?  

For the GEMM (fp32 accumulator through the residual epilogue), the LayerNorm-folded epilogue and the attention kernel: the
kernel's output against an fp64 evaluation of the same formula on the same bf16 operands --
  * fp32 results: RMS and max of the relative difference (accumulation order / MFMA-internal rounding);
  * bf16 results: the fraction of outputs that are NOT the correctly rounded bf16 of the fp64 value ("flips"; each one is a
    whole bf16 ulp, 2^-8 .. 2^-7 relative) and the RMS difference that causes.
Together with tools/error_growth.py this separates "different rounding realisation" from "different arithmetic".
    python tools/precision_probe.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from concepthash_amd import _lib

lib = _lib.load()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(3)


def bf16_round(x64):
    return x64.float().to(torch.bfloat16)


def report(tag, got_bf16, want64):
    want_b = bf16_round(want64)
    diff = got_bf16.float() != want_b.float()
    rms = want64.pow(2).mean().sqrt()
    print(f"{tag}: {float(diff.float().mean()):.3e} of the bf16 outputs differ from bf16(fp64 result); "
          f"rms(out - fp64)/rms {float((got_bf16.double() - want64).pow(2).mean().sqrt() / rms):.3e} vs "
          f"correctly rounded {float((want_b.double() - want64).pow(2).mean().sqrt() / rms):.3e}; "
          f"rms(out - bf16(fp64))/rms {float((got_bf16.double() - want_b.double()).pow(2).mean().sqrt() / rms):.3e}")


def gemm(variant, X, W, bias, M, epi, out=None, resid=None):
    N, K = W.shape
    _lib.check(lib.ch_debug_gemm(variant, _lib.ptr(X), X.shape[0], _lib.ptr(W), _lib.ptr(bias), M, N, K, epi, _lib.ptr(out),
                                 N if out is not None else 0, _lib.ptr(resid), N if resid is not None else 0, None, None,
                                 _lib.stream_ptr()), "ch_debug_gemm")


def gemm_ln(variant, X, W, bias, M, epi, out, stats_in, fold_c, eps):
    N, K = W.shape
    _lib.check(lib.ch_debug_gemm_ln(variant, _lib.ptr(X), X.shape[0], _lib.ptr(W), _lib.ptr(bias), M, N, K, epi, _lib.ptr(out), N,
                                    None, 0, None, None, _lib.ptr(stats_in), _lib.ptr(fold_c), eps, None, None,
                                    _lib.stream_ptr()), "ch_debug_gemm_ln")


M = 2048
for name, N, K in (("qkv-like", 2304, 768), ("fc2-like", 768, 3072), ("up-like", 768, 384)):
    X = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device=dev) * 0.02).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=dev) * 0.02
    want = X.double() @ W.double().t() + bias.double()
    for variant, vname in ((1, "128x128"), (2, "256x256 pp")):
        resid = torch.zeros(M, N, device=dev)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        gemm(variant, X, W, bias, M, 3, out=out, resid=resid)     # EPI_BIAS_RESID: resid += v, out = bf16(v)
        torch.cuda.synchronize()
        rel = (resid.double() - want).abs() / want.pow(2).mean().sqrt()
        print(f"[{name} K={K}, {vname}] fp32 accumulator vs fp64: rms {float(rel.pow(2).mean().sqrt()):.3e}, max {float(rel.max()):.3e}")
        report(f"[{name} K={K}, {vname}] bf16 output", out, want)
    # torch's own fp32 matmul on the same operands, for scale
    t32 = (X.float() @ W.float().t() + bias).double()
    rel = (t32 - want).abs() / want.pow(2).mean().sqrt()
    print(f"[{name} K={K}, torch fp32 matmul on the GPU] vs fp64: rms {float(rel.pow(2).mean().sqrt()):.3e}, max {float(rel.max()):.3e}")

# ---- LayerNorm-folded consumer: y = rstd * (x W'^T - mean * c) + d, statistics single pass from 64-column partial sums
K, N = 768, 768
x = torch.randn(M, K, generator=g, device=dev) * (0.5 + 2 * torch.rand(M, 1, generator=g, device=dev)) + 0.3 * torch.randn(M, 1, generator=g, device=dev)
X = x.to(torch.bfloat16)
W32 = torch.randn(N, K, generator=g, device=dev) * 0.02
gamma = 1 + 0.02 * torch.randn(K, generator=g, device=dev)
beta = 0.02 * torch.randn(K, generator=g, device=dev)
b = torch.randn(N, generator=g, device=dev) * 0.02
Wf = (W32 * gamma).to(torch.bfloat16)
c = Wf.float().sum(1)
d = b + W32 @ beta
xs = X.float().view(M, K // 64, 64)
stats = torch.stack([xs.sum(-1), xs.pow(2).sum(-1)], dim=-1).contiguous()
eps = 1e-5
xd = X.double()
mean = xd.mean(-1, keepdim=True)
var = (xd.pow(2).mean(-1, keepdim=True) - mean * mean).clamp_min(0)
rstd = (var + eps).rsqrt()
want = (xd @ Wf.double().t()) * rstd + (c.double() * (-mean * rstd) + d.double())       # the folded formula in fp64
for variant, vname in ((1, "128x128"), (2, "256x256 pp")):
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    gemm_ln(variant, X, Wf, d, M, 8, out, stats, c, eps)
    torch.cuda.synchronize()
    report(f"[LN-folded qkv-like, {vname}] bf16 output", out, want)

# ---- attention: softmax(q k^T / 8) v with bf16 probabilities into the PV product, normalised by the fp32 row sum
B, ntok, heads = 8, 201, 12
D = heads * 64
qkv = (torch.randn(B * ntok, 3 * D, generator=g, device=dev) * 0.8).to(torch.bfloat16)
pad = torch.zeros(256, 3 * D, dtype=torch.bfloat16, device=dev)
qkv_p = torch.cat([qkv, pad])
out = torch.empty(B * ntok + 256, D, dtype=torch.bfloat16, device=dev)
_lib.check(lib.ch_debug_attention(_lib.ptr(qkv_p), B, ntok, heads, _lib.ptr(out), _lib.stream_ptr()), "ch_debug_attention")
torch.cuda.synchronize()
q, k, v = [t.reshape(B, ntok, heads, 64).transpose(1, 2).double() for t in qkv.view(B * ntok, 3, D).unbind(1)]
s = (q @ k.transpose(-1, -2)) * 0.125
e = torch.exp(s - s.max(-1, keepdim=True).values)
o = ((e.float().to(torch.bfloat16).double() @ v) / e.sum(-1, keepdim=True)).transpose(1, 2).reshape(B * ntok, D)
report("[attention, 201 tokens, bf16 P into PV, fp32 row sum] bf16 output", out[:B * ntok], o)
o_exact = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * ntok, D)
print(f"   (for scale: exact softmax vs the bf16-P formula: rms {float((o - o_exact).pow(2).mean().sqrt() / o_exact.pow(2).mean().sqrt()):.3e})")
