"""GPU: the fused-epilogue bf16 GEMM kernels in isolation (C-ABI test tap `ch_debug_gemm`).
  * each kernel against a plain PyTorch fp32 reference of the same op on the same bf16-rounded operands
    (tolerance: fp32 accumulation-order differences only, atol 2e-3 on O(1) outputs; bf16 output rounding 2^-8 rel);
  * the 256x256 ping-pong kernel against the 128x128 two-phase kernel BIT FOR BIT (same MFMA instruction and the same
    k order per output element, so any difference is a staging race), repeated as a race screen with an L2-warm and an
    L2-cold pattern."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

EPI_BIAS, EPI_QGELU, EPI_GELU, EPI_BIAS_RESID, EPI_SCALE_RESID = range(5)


def _need_experiments():
    """Variants 3 / 5 / 6 (gemm_dp / gemm_ppp / gemm_pq: measured, lost) exist only in an experiments
    build (CH_BUILD_EXPERIMENTS=1); the product library does not carry them."""
    from concepthash_amd import _lib
    if not _lib.load().ch_debug_experiments_built():
        pytest.skip("experiment kernels are not part of the product build (CH_BUILD_EXPERIMENTS=1 builds them)")


def _gemm(variant, X, W, bias, M, epi, out=None, resid=None, scale=None, addend=None):
    from concepthash_amd import _lib
    lib = _lib.load()
    N, K = W.shape
    _lib.check(lib.ch_debug_gemm(variant, _lib.ptr(X), X.shape[0], _lib.ptr(W), _lib.ptr(bias), M, N, K, epi, _lib.ptr(out),
                                 N if out is not None else 0, _lib.ptr(resid), N if resid is not None else 0,
                                 _lib.ptr(scale), _lib.ptr(addend), _lib.stream_ptr()), "ch_debug_gemm")


def _ref(X, W, bias, M, epi, resid0=None, scale=None):
    v = X[:M].float() @ W.float().t() + bias
    if epi == EPI_QGELU:
        v = v * torch.sigmoid(1.702 * v)
    if epi == EPI_GELU:
        v = torch.nn.functional.gelu(v)
    if epi == EPI_BIAS_RESID:
        return v, resid0[:M] + v
    if epi == EPI_SCALE_RESID:
        return None, resid0[:M] + scale * v
    return v, None


def _inputs(M, N, K, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    Mp = (M + 255) // 256 * 256
    X = torch.zeros(Mp, K, dtype=torch.bfloat16, device="cuda")
    X[:M] = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device="cuda")
    resid = torch.randn(Mp, N, generator=g, device="cuda")
    return X, W, bias, resid


@pytest.mark.parametrize("variant", [1, 2, 3, 7])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (1000, 768, 768), (2011, 2304, 768), (513, 768, 3072), (700, 3072, 768),
                                   (300, 768, 384), (257, 256, 640)])
def test_gemm_against_torch_fp32(variant, M, N, K):
    if variant == 3:
        _need_experiments()
    X, W, bias, resid0 = _inputs(M, N, K)
    scale = torch.tensor([0.7], device="cuda")
    for epi in (EPI_BIAS, EPI_QGELU, EPI_GELU, EPI_BIAS_RESID, EPI_SCALE_RESID):
        out = torch.full((X.shape[0], N), float("nan"), dtype=torch.bfloat16, device="cuda")
        resid = resid0.clone()
        _gemm(variant, X, W, bias, M, epi, out=out if epi != EPI_SCALE_RESID else None,
              resid=resid if epi >= EPI_BIAS_RESID else None, scale=scale)
        torch.cuda.synchronize()
        v, r = _ref(X, W, bias, M, epi, resid0, 0.7)
        if v is not None:
            assert torch.allclose(out[:M].float(), v, atol=2e-3 + 1e-2, rtol=2 ** -7), (epi, float((out[:M].float() - v).abs().max()))
            assert bool(torch.isnan(out[M:].float()).all())                      # rows >= M are never written
        if r is not None:
            assert torch.allclose(resid[:M], r, atol=2e-3, rtol=1e-4), (epi, float((resid[:M] - r).abs().max()))
            assert torch.equal(resid[M:], resid0[M:])
    # EPI_SCALE_RESID with the deferred bf16 addend: resid += addend + scale * (acc + bias)
    addend = torch.randn(X.shape[0], N, device="cuda").to(torch.bfloat16)
    resid = resid0.clone()
    _gemm(variant, X, W, bias, M, EPI_SCALE_RESID, resid=resid, scale=scale, addend=addend)
    torch.cuda.synchronize()
    _, r = _ref(X, W, bias, M, EPI_SCALE_RESID, resid0, 0.7)
    assert torch.allclose(resid[:M], r + addend[:M].float(), atol=2e-3, rtol=1e-4)
    assert torch.equal(resid[M:], resid0[M:])


@pytest.mark.parametrize("M,N,K", [(51456, 768, 768), (4096, 2304, 768), (3000, 3072, 768), (2500, 768, 3072), (1500, 768, 384)])
def test_pingpong_equals_two_phase_bitwise_and_is_race_free(M, N, K):
    X, W, bias, resid0 = _inputs(M, N, K, seed=1)
    ref = torch.empty(X.shape[0], N, dtype=torch.bfloat16, device="cuda")
    r_ref = resid0.clone()
    _gemm(1, X, W, bias, M, EPI_BIAS_RESID, out=ref, resid=r_ref)
    junk = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda")   # 256 MiB: evicts L2 + MALL when written
    from concepthash_amd import _lib
    # 2: ping-pong, four phases per K-tile; 4: the same kernel's coarse schedule (two phases per K-tile); 3: dual-WG ring (experiment)
    variants = [2] + ([4, 3] if _lib.load().ch_debug_experiments_built() else [])
    for variant in variants:
        for it in range(4):
            out = torch.zeros_like(ref)
            r = resid0.clone()
            if it % 2:
                junk.fill_(float(it))
            _gemm(variant, X, W, bias, M, EPI_BIAS_RESID, out=out, resid=r)
            torch.cuda.synchronize()
            assert torch.equal(out[:M].view(torch.int16), ref[:M].view(torch.int16)), f"variant {variant} iteration {it}"
            assert torch.equal(r[:M], r_ref[:M]), f"variant {variant} iteration {it}"


@pytest.mark.parametrize("M,N,K", [(50432, 384, 768), (50432, 768, 384), (1500, 768, 96), (777, 128, 3072), (300, 2304, 160)])
def test_ring_kernel_equals_two_phase_bitwise(M, N, K):
    """gemm_r4.hip (variant 7): 128x128x32 tiles behind a four-stage LDS-DMA ring with three K-steps in flight.  Same MFMA and the
    same k order per output element as the 128x128x64 two-phase kernel -> identical bits, for every epilogue, with the L2
    cold and warm (a stage overwritten before its readers finished, or read before it landed, shows up here).  The dispatcher sends
    the forward epilogues of small grids (<= 8,192 rows) to it; at the benchmark's size it lost to the two-phase kernel."""
    X, W, bias, resid0 = _inputs(M, N, K, seed=2)
    scale = torch.tensor([0.3], device="cuda")
    addend = torch.randn(X.shape[0], N, device="cuda").to(torch.bfloat16)
    junk = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda")
    base = 1 if K % 64 == 0 else None
    for epi in (EPI_BIAS, EPI_QGELU, EPI_GELU, EPI_BIAS_RESID, EPI_SCALE_RESID):
        outs = []
        for variant, it in ((base, 0), (7, 0), (7, 1), (7, 2)):
            if variant is None:
                continue
            out = torch.zeros(X.shape[0], N, dtype=torch.bfloat16, device="cuda")
            r = resid0.clone()
            if it == 1:
                junk.fill_(1.0)
            _gemm(variant, X, W, bias, M, epi, out=out if epi != EPI_SCALE_RESID else None,
                  resid=r if epi >= EPI_BIAS_RESID else None, scale=scale, addend=addend if epi == EPI_SCALE_RESID else None)
            torch.cuda.synchronize()
            outs.append((out, r))
        for out, r in outs[1:]:
            assert torch.equal(out.view(torch.int16), outs[0][0].view(torch.int16)), epi
            assert torch.equal(r, outs[0][1]), epi
    if base is None:       # K % 64 != 0: no two-phase kernel to compare with -> fp32 reference
        out = torch.zeros(X.shape[0], N, dtype=torch.bfloat16, device="cuda")
        _gemm(7, X, W, bias, M, EPI_BIAS, out=out)
        v, _ = _ref(X, W, bias, M, EPI_BIAS, resid0, 0.3)
        assert torch.allclose(out[:M].float(), v, atol=2e-3 + 1e-2, rtol=2 ** -7)


@pytest.mark.parametrize("M,N,K", [(51456, 2304, 768), (51456, 768, 768), (4000, 3072, 768), (2500, 768, 3072), (256, 256, 128),
                                   (70000, 256, 256)])
def test_persistent_pingpong_equals_two_phase_bitwise(M, N, K):
    """gemm_ppp.hip (variant 5): tiles streamed by persistent workgroups with cross-tile prefetch; bf16-output epilogues.
    Same MFMA / k order as the 128x128 kernel -> bit-identical outputs; repeated with cold and warm caches as a race screen."""
    _need_experiments()
    X, W, bias, _ = _inputs(M, N, K, seed=2)
    junk = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda")
    for epi in (EPI_BIAS, EPI_QGELU):
        ref = torch.empty(X.shape[0], N, dtype=torch.bfloat16, device="cuda")
        _gemm(1, X, W, bias, M, epi, out=ref)
        for it in range(4):
            out = torch.zeros_like(ref)
            if it % 2:
                junk.fill_(float(it))
            _gemm(5, X, W, bias, M, epi, out=out)
            torch.cuda.synchronize()
            assert torch.equal(out[:M].view(torch.int16), ref[:M].view(torch.int16)), (epi, it)
            assert not bool(out[M:].any())                    # padding rows untouched
    resid = torch.zeros(X.shape[0], N, device="cuda")
    with pytest.raises(RuntimeError, match="gemm_ppp"):
        _gemm(5, X, W, bias, M, EPI_BIAS_RESID, out=ref, resid=resid)


# ---- LayerNorm folded into the consumer GEMM (kernels.h EPI_BIAS_STATS .. EPI_FOLD_GELU) -----------
EPI_BIAS_STATS, EPI_SCALE_RESID_STATS, EPI_FOLD_BIAS, EPI_FOLD_QGELU, EPI_FOLD_GELU = 6, 7, 8, 9, 10


def _gemm_ln(variant, X, W, bias, M, epi, out=None, resid=None, scale=None, addend=None, stats_in=None, fold_c=None,
             eps=1e-5, stats_out=None, hb_out=None):
    from concepthash_amd import _lib
    lib = _lib.load()
    N, K = W.shape
    _lib.check(lib.ch_debug_gemm_ln(variant, _lib.ptr(X), X.shape[0], _lib.ptr(W), _lib.ptr(bias), M, N, K, epi,
                                    _lib.ptr(out), N if out is not None else 0, _lib.ptr(resid),
                                    N if resid is not None else 0, _lib.ptr(scale), _lib.ptr(addend), _lib.ptr(stats_in),
                                    _lib.ptr(fold_c), eps, _lib.ptr(stats_out), _lib.ptr(hb_out), _lib.stream_ptr()),
               "ch_debug_gemm_ln")


def _slice_stats(x_bf16, M):
    """[M, N/64, 2] (sum, sum of squares) over 64-column slices of the bf16 values, in fp64."""
    x = x_bf16[:M].double().view(M, -1, 64)
    return torch.stack([x.sum(-1), (x * x).sum(-1)], dim=-1)


@pytest.mark.parametrize("variant", [1, 2, 7])
@pytest.mark.parametrize("M,N,K", [(1000, 768, 768), (513, 768, 3072), (300, 768, 384), (2011, 256, 640)])
def test_statistics_producers(variant, M, N, K):
    X, W, bias, resid0 = _inputs(M, N, K, seed=3)
    # bias + statistics: the output is bit-identical to the plain bias epilogue, statistics describe the ROUNDED output
    ref = torch.zeros(X.shape[0], N, dtype=torch.bfloat16, device="cuda")
    _gemm(variant, X, W, bias, M, EPI_BIAS, out=ref)
    out = torch.zeros_like(ref)
    stats = torch.full((X.shape[0], N // 64, 2), float("nan"), device="cuda")
    _gemm_ln(variant, X, W, bias, M, EPI_BIAS_STATS, out=out, stats_out=stats)
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))
    want = _slice_stats(out, M)
    assert torch.allclose(stats[:M].double(), want, rtol=1e-5, atol=1e-4), float((stats[:M].double() - want).abs().max())
    assert bool(torch.isnan(stats[M:]).all())
    # residual update + bf16 copy + statistics of the copy
    scale = torch.tensor([0.7], device="cuda")
    addend = torch.randn(X.shape[0], N, device="cuda").to(torch.bfloat16)
    r_ref = resid0.clone()
    _gemm(variant, X, W, bias, M, EPI_SCALE_RESID, resid=r_ref, scale=scale, addend=addend)
    r = resid0.clone()
    hb = torch.zeros(X.shape[0], N, dtype=torch.bfloat16, device="cuda")
    stats = torch.full((X.shape[0], N // 64, 2), float("nan"), device="cuda")
    _gemm_ln(variant, X, W, bias, M, EPI_SCALE_RESID_STATS, resid=r, scale=scale, addend=addend, stats_out=stats, hb_out=hb)
    torch.cuda.synchronize()
    assert torch.equal(r, r_ref)
    assert torch.equal(hb[:M].view(torch.int16), r[:M].to(torch.bfloat16).view(torch.int16))
    assert not bool(hb[M:].any())
    want = _slice_stats(hb, M)
    assert torch.allclose(stats[:M].double(), want, rtol=1e-5, atol=1e-4)
    assert bool(torch.isnan(stats[M:]).all())


@pytest.mark.parametrize("M,N,K", [(51456, 768, 768), (4096, 2304, 768), (3000, 3072, 768), (2500, 768, 3072), (700, 256, 128), (515, 512, 256)])
def test_free_tail_schedule_is_bit_identical_and_race_free(M, N, K):
    """gemm_pp_kernel<EPI, 0, 2> (variant 8 of the taps, experiments build): the last K-tile's phases run without barriers and the
    epilogue stages through the even operand buffer in two 64-row passes, so that a wave that is done stores while its SIMD
    partner still issues MFMAs.  Same MFMA order: bit-identical to the dispatched schedule for every bf16-output epilogue --
    plain, quick_gelu, with row statistics, LayerNorm-folded -- with L2 / MALL warm and evicted, K = 128 (a single iteration)
    included."""
    _need_experiments()
    X, W, bias, _ = _inputs(M, N, K, seed=9)
    junk = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda")
    Mp = X.shape[0]
    stats = torch.zeros(Mp, K // 64, 2, device="cuda")
    stats[:M] = _slice_stats(X, M).float()
    fold_c = torch.randn(N, device="cuda")
    for epi in (EPI_BIAS, 1, EPI_BIAS_STATS, 9):
        outs = []
        for variant, it in ((2, 0), (8, 0), (8, 1), (8, 2), (8, 3)):
            out = torch.full((Mp, N), float("nan"), dtype=torch.bfloat16, device="cuda")
            st = torch.zeros(Mp, N // 64, 2, device="cuda")
            if it % 2:
                junk.fill_(float(it))
            if epi in (EPI_BIAS_STATS, 9):
                if epi == 9 and K > 1280:
                    break
                _gemm_ln(variant, X, W, bias, M, epi, out=out, stats_in=stats, fold_c=fold_c, eps=1e-5, stats_out=st)
            else:
                _gemm(variant, X, W, bias, M, epi, out=out)
            torch.cuda.synchronize()
            outs.append((out, st))
        for out, st in outs[1:]:
            assert torch.equal(out[:M].view(torch.int16), outs[0][0][:M].view(torch.int16)), (epi, M, N, K)
            assert torch.equal(st[:M], outs[0][1][:M]), (epi, M, N, K)


@pytest.mark.parametrize("M,K", [(51456, 768), (25728, 768), (1000, 768), (130, 1024), (515, 384), (77, 64)])
def test_whole_row_kernel_is_bit_identical_and_race_free(M, K):
    """gemm_rows_kernel (experiments/gemm_rows.hip, variant 9 of the taps, experiments build; measured no faster than the 128x128
    kernel, not dispatched): a workgroup owns 128 whole rows x N = 384, a wave 32 full rows.  Same MFMA and the same ascending-k
    order as the 128x128 kernel: bit-identical to it for every epilogue it takes -- bias, exact GELU, LayerNorm-folded (+ GELU /
    quick_gelu) -- with caches warm and evicted, ragged row counts, K = 64 (two K-steps) included."""
    _need_experiments()
    N = 384
    X, W, bias, _ = _inputs(M, N, K, seed=11)
    junk = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda")
    Mp = X.shape[0]
    stats = torch.zeros(Mp, max(1, K // 64), 2, device="cuda")
    stats[:M] = _slice_stats(X, M).float()
    fold_c = torch.randn(N, device="cuda")
    epis = [EPI_BIAS, EPI_GELU] + ([8, 9, 10] if K % 128 == 0 and K <= 1280 else [])
    for epi in epis:
        outs = []
        for variant, it in ((1, 0), (9, 0), (9, 1), (9, 2), (9, 3)):
            out = torch.full((Mp, N), float("nan"), dtype=torch.bfloat16, device="cuda")
            if it % 2:
                junk.fill_(float(it))
            if epi >= 8:
                _gemm_ln(variant, X, W, bias, M, epi, out=out, stats_in=stats, fold_c=fold_c, eps=1e-5)
            else:
                _gemm(variant, X, W, bias, M, epi, out=out)
            torch.cuda.synchronize()
            outs.append(out)
        assert not torch.isnan(outs[1][:M].float()).any()
        for out in outs[1:]:
            assert torch.equal(out[:M].view(torch.int16), outs[0][:M].view(torch.int16)), (epi, M, K)
            assert torch.isnan(out[M:].float()).all()          # rows past M are never written


@pytest.mark.parametrize("M,N,K", [(51456, 384, 768), (25728, 384, 768), (1000, 384, 768), (130, 384, 1024), (515, 768, 384), (77, 384, 64),
                                   (300, 384, 192), (2049, 1152, 320)])
def test_wide_kernel_is_bit_identical_and_race_free(M, N, K):
    """gemm_wide_kernel (experiments/gemm_wide.hip, variant 10 of the taps, experiments build; 9-14 % faster than the 128x128 kernel as
    an isolated launch at 51k rows, neutral end to end, not dispatched -- profiles/r04_gemm_wide_ab.txt): 256 x 384 x
    32 tiles, 64 x 192 wave tiles, three-stage LDS-DMA ring behind counted `vmcnt(5)`, two-phase ping-pong with waves 4-7 one barrier
    behind.  Same MFMA and the same ascending-k order per output element as the 128x128 kernel: bit-identical to it for every epilogue
    the taps reach -- bias, exact GELU, LayerNorm-folded + GELU -- so any difference is a staging race; repeated with caches warm and
    evicted, ragged row counts, K = 64 (two K-tiles: prologue only), K = 192 / 320 (the ring wraps), N = 768 / 1,152 (two / three
    column tiles)."""
    _need_experiments()
    X, W, bias, _ = _inputs(M, N, K, seed=13)
    junk = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda")
    Mp = X.shape[0]
    stats = torch.zeros(Mp, max(1, K // 64), 2, device="cuda")
    if K % 64 == 0:
        stats[:M] = _slice_stats(X, M).float()
    fold_c = torch.randn(N, device="cuda")
    epis = [EPI_BIAS, EPI_GELU] + ([10] if K % 128 == 0 and K <= 1280 else [])
    for epi in epis:
        outs = []
        for variant, it in ((1, 0), (10, 0), (10, 1), (10, 2), (10, 3)):
            out = torch.full((Mp, N), float("nan"), dtype=torch.bfloat16, device="cuda")
            if it % 2:
                junk.fill_(float(it))
            if epi >= 8:
                _gemm_ln(variant, X, W, bias, M, epi, out=out, stats_in=stats, fold_c=fold_c, eps=1e-5)
            else:
                _gemm(variant, X, W, bias, M, epi, out=out)
            torch.cuda.synchronize()
            outs.append(out)
        assert not torch.isnan(outs[1][:M].float()).any()
        for out in outs[1:]:
            assert torch.equal(out[:M].view(torch.int16), outs[0][:M].view(torch.int16)), (epi, M, N, K)
            assert torch.isnan(out[M:].float()).all()          # rows past M are never written


@pytest.mark.parametrize("variant", [1, 2, 4, 7])
@pytest.mark.parametrize("M,N,K", [(1000, 2304, 768), (700, 3072, 768), (515, 384, 768), (300, 256, 128), (257, 512, 1280)])
def test_layernorm_folded_consumers(variant, M, N, K):
    """y = act(LN(x) W^T + b) computed as act(rstd * (x W'^T - mean * c) + d) with W' = bf16(W * gamma)."""
    if variant == 4:
        _need_experiments()
    if variant in (2, 4) and N % 256:
        pytest.skip("the 256x256 kernel needs N % 256 == 0")
    g = torch.Generator(device="cuda").manual_seed(5)
    Mp = (M + 255) // 256 * 256
    x = torch.randn(M, K, generator=g, device="cuda") * (0.5 + 3 * torch.rand(M, 1, generator=g, device="cuda")) \
        + 2 * torch.randn(M, 1, generator=g, device="cuda")            # per-row scale and a non-zero mean
    x[:, 5] *= 20                                                       # an outlier channel
    X = torch.zeros(Mp, K, dtype=torch.bfloat16, device="cuda")
    X[:M] = x.to(torch.bfloat16)
    W32 = torch.randn(N, K, generator=g, device="cuda") * K ** -0.5
    gamma = 1 + 0.3 * torch.randn(K, generator=g, device="cuda")
    beta = 0.2 * torch.randn(K, generator=g, device="cuda")
    b = torch.randn(N, generator=g, device="cuda")
    Wf = (W32 * gamma).to(torch.bfloat16)
    c = Wf.float().sum(1)
    d = b + W32 @ beta
    stats = torch.zeros(Mp, K // 64, 2, device="cuda")
    stats[:M] = _slice_stats(X, M).float()
    eps = 1e-5
    xn = torch.nn.functional.layer_norm(X[:M].double(), (K,), gamma.double(), beta.double(), eps)
    pre = xn @ W32.double().t() + b.double()
    for epi, f in ((EPI_FOLD_BIAS, lambda v: v), (EPI_FOLD_QGELU, lambda v: v * torch.sigmoid(1.702 * v)),
                   (EPI_FOLD_GELU, torch.nn.functional.gelu)):
        out = torch.full((Mp, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        _gemm_ln(variant, X, Wf, d, M, epi, out=out, stats_in=stats, fold_c=c, eps=eps)
        torch.cuda.synchronize()
        want = f(pre).float()
        err = (out[:M].float() - want).abs()
        # bf16 rounding of W' (2^-9 relative per term, random signs over K terms) + bf16 output rounding
        assert torch.allclose(out[:M].float(), want, atol=3e-2, rtol=2 ** -7), (epi, float(err.max()))
        assert float(err.pow(2).mean().sqrt()) < 6e-3, (epi, float(err.pow(2).mean().sqrt()))
        assert bool(torch.isnan(out[M:].float()).all())


@pytest.fixture
def splitk():
    from concepthash_amd import _lib
    lib = _lib.load()
    lib.ch_debug_set_gemm_splitk(1)
    yield
    lib.ch_debug_set_gemm_splitk(0)


@pytest.mark.parametrize("M,N,K", [(51456, 768, 768), (51456, 2304, 768), (6000, 768, 3072), (20000, 768, 3072), (300, 256, 256)])
def test_splitk_tail_of_the_pingpong_kernel(splitk, M, N, K):
    """Tiles of the last partial round are cut along K; the last-arriving slice sums all partial tiles in slice order.
    Against the 128x128 kernel (different summation order: fp32 rounding only), deterministic from launch to launch, and the
    tickets reset themselves (repeated launches)."""
    X, W, bias, resid0 = _inputs(M, N, K, seed=4)
    ref = torch.zeros(X.shape[0], N, dtype=torch.bfloat16, device="cuda")
    _gemm(1, X, W, bias, M, EPI_QGELU, out=ref)
    outs = []
    junk = torch.empty(32 * 1024 * 1024, dtype=torch.float32, device="cuda")
    for it in range(4):
        out = torch.zeros_like(ref)
        if it % 2:
            junk.fill_(float(it))
        _gemm(2, X, W, bias, M, EPI_QGELU, out=out)
        torch.cuda.synchronize()
        outs.append(out)
        d = (out[:M].float() - ref[:M].float()).abs()
        assert float(d.max()) <= 2 ** -7 * float(ref[:M].float().abs().max()) + 1e-3, (it, float(d.max()))
        assert float((d > 0).float().mean()) < 0.02          # a different bf16 rounding only where fp32 sums differ in the last bits
        assert not bool(out[M:].any())
    for o in outs[1:]:
        assert torch.equal(o.view(torch.int16), outs[0].view(torch.int16))
    # fp32 residual epilogue and the LayerNorm-fold producers go through the same fix-up path
    r_ref, r = resid0.clone(), resid0.clone()
    scale = torch.tensor([0.7], device="cuda")
    _gemm(1, X, W, bias, M, EPI_SCALE_RESID, resid=r_ref, scale=scale)
    _gemm(2, X, W, bias, M, EPI_SCALE_RESID, resid=r, scale=scale)
    stats = torch.zeros(X.shape[0], N // 64, 2, device="cuda")
    out = torch.zeros_like(ref)
    _gemm_ln(2, X, W, bias, M, EPI_BIAS_STATS, out=out, stats_out=stats)
    torch.cuda.synchronize()
    assert torch.allclose(r[:M], r_ref[:M], atol=2e-4, rtol=1e-5) and torch.equal(r[M:], r_ref[M:])
    want = _slice_stats(out, M)
    assert torch.allclose(stats[:M].double(), want, rtol=1e-5, atol=1e-4)



@pytest.mark.parametrize("M,N,K", [(51456, 768, 768), (4096, 2304, 768), (3000, 384, 768), (2500, 768, 3072), (300, 128, 128),
                                   (70000, 256, 256), (700, 640, 1280)])
def test_256x128_pingpong_equals_two_phase_bitwise_and_is_race_free(M, N, K):
    """gemm_pq.hip (variant 6): 256x128 tile, two phases per K-tile, three 16 KB units per K-tile.  Same MFMA and the same k
    order per output element as the 128x128 kernel -> bit-identical for every epilogue; repeated with cold and warm caches."""
    _need_experiments()
    X, W, bias, resid0 = _inputs(M, N, K, seed=9)
    scale = torch.tensor([0.7], device="cuda")
    addend = torch.randn(X.shape[0], N, device="cuda").to(torch.bfloat16)
    junk = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda")
    for epi in (EPI_BIAS, EPI_QGELU, EPI_GELU, EPI_BIAS_RESID, EPI_SCALE_RESID):
        ref = torch.zeros(X.shape[0], N, dtype=torch.bfloat16, device="cuda")
        r_ref = resid0.clone()
        _gemm(1, X, W, bias, M, epi, out=ref if epi != EPI_SCALE_RESID else None, resid=r_ref if epi >= EPI_BIAS_RESID else None,
              scale=scale, addend=addend if epi == EPI_SCALE_RESID else None)
        for it in range(3):
            out = torch.zeros_like(ref)
            r = resid0.clone()
            if it % 2:
                junk.fill_(float(it))
            _gemm(6, X, W, bias, M, epi, out=out if epi != EPI_SCALE_RESID else None, resid=r if epi >= EPI_BIAS_RESID else None,
                  scale=scale, addend=addend if epi == EPI_SCALE_RESID else None)
            torch.cuda.synchronize()
            assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), (epi, it)
            assert torch.equal(r, r_ref), (epi, it)
    if K <= 1280:   # LayerNorm-fold consumers and the statistics producers through the same kernel
        stats_in = torch.zeros(X.shape[0], K // 64, 2, device="cuda")
        stats_in[:M] = _slice_stats(X, M).float()
        fold_c = torch.randn(N, device="cuda")
        for epi in (EPI_BIAS_STATS, EPI_FOLD_BIAS, EPI_FOLD_QGELU, EPI_FOLD_GELU):
            kw = dict(stats_in=stats_in, fold_c=fold_c) if epi != EPI_BIAS_STATS else {}
            ref = torch.zeros(X.shape[0], N, dtype=torch.bfloat16, device="cuda")
            out = torch.zeros_like(ref)
            st_ref = torch.zeros(X.shape[0], N // 64, 2, device="cuda")
            st = torch.zeros_like(st_ref)
            _gemm_ln(1, X, W, bias, M, epi, out=ref, stats_out=st_ref if epi == EPI_BIAS_STATS else None, **kw)
            _gemm_ln(6, X, W, bias, M, epi, out=out, stats_out=st if epi == EPI_BIAS_STATS else None, **kw)
            torch.cuda.synchronize()
            assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), epi
            assert torch.equal(st, st_ref), epi
