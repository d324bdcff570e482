"""GPU: the kernels of the training step in isolation (C-ABI taps ch_debug_attention_bwd / ch_debug_wgrad / ch_debug_ln_bwd /
ch_debug_act), each against torch fp32 autograd of the same op on the same bf16-rounded operands.  Tolerances: the kernels keep
bf16 operands and fp32 accumulators, so a result differs from the fp32 reference by the bf16 rounding of intermediate operands
(P, dS: 2^-9 relative per term) and of the bf16 output (2^-9 relative)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from concepthash_amd import _lib as L
    return L, L.load()


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


@pytest.mark.parametrize("B,ntok,heads", [(2, 21, 2), (3, 41, 2), (2, 201, 4), (1, 64, 1), (2, 257, 2)])
def test_attention_backward_against_autograd(B, ntok, heads):
    L, lib = _lib()
    D = heads * 64
    g = torch.Generator(device="cuda").manual_seed(ntok)
    qkv = (torch.randn(B * ntok, 3 * D, generator=g, device="cuda") * 1.5).to(torch.bfloat16)
    dO = torch.randn(B * ntok, D, generator=g, device="cuda").to(torch.bfloat16)
    out = torch.full((B * ntok, 3 * D), float("nan"), dtype=torch.bfloat16, device="cuda")
    L.check(lib.ch_debug_attention_bwd(L.ptr(qkv), L.ptr(dO), B, ntok, heads, L.ptr(out), None, 0, L.stream_ptr()), "attention_bwd")
    torch.cuda.synchronize()
    x = qkv.float().view(B, ntok, 3, heads, 64).permute(2, 0, 3, 1, 4).contiguous().requires_grad_(True)   # [3, B, h, N, 64]
    q, k, v = x[0], x[1], x[2]
    p = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1)
    o = p @ v                                                                                                 # [B, h, N, 64]
    o.backward(dO.float().view(B, ntok, heads, 64).permute(0, 2, 1, 3))
    want = x.grad.permute(1, 3, 0, 2, 4).reshape(B * ntok, 3 * D)
    got = out.float()
    assert not bool(torch.isnan(got).any())
    for j, name in enumerate(("dq", "dk", "dv")):
        a, w = got[:, j * D:(j + 1) * D], want[:, j * D:(j + 1) * D]
        assert _rel(a, w) < 1.5e-2, (name, _rel(a, w))
        assert float((a - w).abs().max()) < 3e-2 * float(w.abs().max()) + 1e-3, name


@pytest.mark.parametrize("B,ntok,heads,ncon", [(2, 21, 2, 4), (2, 201, 4, 4), (1, 41, 1, 8)])
def test_attention_backward_with_a_cotangent_on_the_probabilities(B, ntok, heads, ncon):
    """dpext: gradient arriving at the softmax rows of the last `ncon` (concept) tokens over tokens 1 .. ntok-ncon-1 -- what the
    attention-diversity term of the loss produces -- on top of the usual dO."""
    L, lib = _lib()
    D = heads * 64
    npatch = ntok - ncon - 1
    g = torch.Generator(device="cuda").manual_seed(ntok + 1)
    qkv = (torch.randn(B * ntok, 3 * D, generator=g, device="cuda") * 1.5).to(torch.bfloat16)
    dO = torch.randn(B * ntok, D, generator=g, device="cuda").to(torch.bfloat16)
    dpext = torch.randn(B, heads, ncon, npatch, generator=g, device="cuda") * 3
    out = torch.full((B * ntok, 3 * D), float("nan"), dtype=torch.bfloat16, device="cuda")
    L.check(lib.ch_debug_attention_bwd(L.ptr(qkv), L.ptr(dO), B, ntok, heads, L.ptr(out), L.ptr(dpext), ncon, L.stream_ptr()),
            "attention_bwd")
    torch.cuda.synchronize()
    x = qkv.float().view(B, ntok, 3, heads, 64).permute(2, 0, 3, 1, 4).contiguous().requires_grad_(True)
    p = torch.softmax(x[0] @ x[1].transpose(-1, -2) * 0.125, dim=-1)
    o = p @ x[2]
    loss = (o * dO.float().view(B, ntok, heads, 64).permute(0, 2, 1, 3)).sum() + (p[:, :, -ncon:, 1:-ncon] * dpext).sum()
    loss.backward()
    want = x.grad.permute(1, 3, 0, 2, 4).reshape(B * ntok, 3 * D)
    got = out.float()
    for j, name in enumerate(("dq", "dk", "dv")):
        a, w = got[:, j * D:(j + 1) * D], want[:, j * D:(j + 1) * D]
        assert _rel(a, w) < 1.5e-2, (name, _rel(a, w))


@pytest.mark.parametrize("rows,N,K", [(1000, 128, 128), (4321, 768, 384), (4321, 384, 768), (51456, 256, 128), (31, 128, 256)])
def test_weight_gradient_product(rows, N, K):
    L, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(rows)
    ra = (rows + 31) // 32 * 32 + 64
    A = torch.zeros(ra, N, dtype=torch.bfloat16, device="cuda")
    Bm = torch.zeros(ra, K, dtype=torch.bfloat16, device="cuda")
    A[:rows] = torch.randn(rows, N, generator=g, device="cuda").to(torch.bfloat16)
    Bm[:rows] = torch.randn(rows, K, generator=g, device="cuda").to(torch.bfloat16)
    Bm[rows:] = 7.0                                            # operand B's padding rows need not be zero (A's are)
    out = torch.full((N, K), float("nan"), device="cuda")
    L.check(lib.ch_debug_wgrad(L.ptr(A), N, L.ptr(Bm), K, rows, ra, N, K, L.ptr(out), L.stream_ptr()), "wgrad")
    torch.cuda.synchronize()
    want = A[:rows].double().t() @ Bm[:rows].double()
    assert _rel(out, want) < 1e-5, _rel(out, want)
    out2 = torch.empty_like(out)
    L.check(lib.ch_debug_wgrad(L.ptr(A), N, L.ptr(Bm), K, rows, ra, N, K, L.ptr(out2), L.stream_ptr()), "wgrad")
    torch.cuda.synchronize()
    assert torch.equal(out, out2)                             # chunk partials summed in a fixed order: run-to-run identical


@pytest.mark.parametrize("rows,D", [(37, 128), (1000, 768), (513, 1280)])
def test_layernorm_backward_rows(rows, D):
    L, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(D)
    x = (torch.randn(rows, D, generator=g, device="cuda") * (0.5 + 2 * torch.rand(rows, 1, generator=g, device="cuda"))
         + torch.randn(rows, 1, generator=g, device="cuda")).to(torch.bfloat16)
    gamma = 1 + 0.3 * torch.randn(D, generator=g, device="cuda")
    dy = torch.randn(rows, D, generator=g, device="cuda")
    dyg = (dy * gamma).to(torch.bfloat16)
    dres = torch.randn(rows, D, generator=g, device="cuda")
    out = torch.empty_like(dres)
    out_b = torch.empty(rows, D, dtype=torch.bfloat16, device="cuda")
    xhat = torch.empty(rows, D, dtype=torch.bfloat16, device="cuda")
    eps = 1e-5
    L.check(lib.ch_debug_ln_bwd(L.ptr(dyg), L.ptr(x), rows, D, eps, L.ptr(dres), L.ptr(out), L.ptr(out_b), L.ptr(xhat),
                                L.stream_ptr()), "ln_bwd")
    torch.cuda.synchronize()
    xd = x.double().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xd, (D,), None, None, eps)        # gamma already folded into dyg
    y.backward(dyg.double())
    want = dres.double() + xd.grad
    assert _rel(out, want) < 1e-5, _rel(out, want)
    assert torch.equal(out_b, out.to(torch.bfloat16))
    assert _rel(xhat.float(), y.detach()) < 4e-3


@pytest.mark.parametrize("act", [0, 1])
def test_activation_forward_and_derivative(act):
    L, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(act)
    n = 8 * 4099
    pre = (torch.randn(n, generator=g, device="cuda") * 3).to(torch.bfloat16)
    up = torch.randn(n, generator=g, device="cuda").to(torch.bfloat16)
    scale = torch.tensor([0.7], device="cuda")
    out = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    L.check(lib.ch_debug_act(None, L.ptr(pre), n, act, None, 0, L.ptr(out), L.stream_ptr()), "act_fwd")
    f = (lambda v: v * torch.sigmoid(1.702 * v)) if act == 0 else torch.nn.functional.gelu
    xd = pre.double().requires_grad_(True)
    y = f(xd)
    torch.cuda.synchronize()
    assert torch.allclose(out.double(), y.detach(), atol=1e-3, rtol=2 ** -8)
    y.backward(up.double() * 0.7)
    L.check(lib.ch_debug_act(L.ptr(up), L.ptr(pre), n, act, L.ptr(scale), 1, L.ptr(out), L.stream_ptr()), "act_bwd")
    torch.cuda.synchronize()
    assert torch.allclose(out.double(), xd.grad, atol=2e-3, rtol=2 ** -7), float((out.double() - xd.grad).abs().max())
