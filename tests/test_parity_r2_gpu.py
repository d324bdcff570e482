"""GPU parity, second tier: the evidence VERDICT.md (round 1) found missing.

  (a) reference-generated fixture at 201 tokens that reaches the DISPATCHED kernels of the real configs (256x256 ping-pong
      GEMM, KB = 7 attention with its masked tail, LN-fold chain, final-layer pruning);
  (c) full depth: ViT-L/14 x 24 layers and ViT-S/16 x 12 layers against the fp32 oracle;
  (d) batch-256 ViT-B/16 ch_encode: images of the big batch are bit-equal to an oracle-checked small batch;
  (e) north-star number: |mAP@all(HIP codes) - mAP@all(fp32-oracle codes)| and the bit-flip rate on labelled,
      class-structured sets, in three margin regimes (< 1e-3 asserted where the ranking has a trained model's margin);
  (f) outlier-heavy residual channels (x100, LayerNorm gamma to match) through 12 layers;
  (g) the bound against the rounding-emulating oracle: measured against an oracle that rounds at the SAME points as the
      default chain (emulate_fold) and against the one that rounds after the LayerNorm -- both sit at the same distance from
      the HIP path as from each other (error-growth table): what is left is not the placement of the
      rounding points but bf16 roundings that flip on 1e-6 differences in accumulation order, summed over ~9 rounded GEMM
      outputs per layer.
Errors are max-abs / RMS of the compared tensor unless stated; every measured value is printed.
"""
import os

import numpy as np
import pytest
import torch

from conftest import fixture_images, load_fixture

pytestmark = pytest.mark.gpu


def _rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.double() - b.double()).abs().max() / b.double().pow(2).mean().sqrt().clamp_min(1e-12))


def _rms_err(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.double() - b.double()).pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt().clamp_min(1e-12))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _encoder(sd, heads, **kw):
    from concepthash_amd.encoder import ConceptHashEncoder
    return ConceptHashEncoder(sd, heads=heads, **kw)


# ---- (a) ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pp_min_k", [None, 256])
def test_reference_golden_at_201_tokens(dev, monkeypatch, pp_min_k):
    """tests/golden/encode_n201.npz = the reference's LGHWithFixedPrompt (models/arch/coop.py:524-598) at image 224 /
    patch 16, D = 256, 4 heads, 2 layers, bf16-representable weights and images.  Default dispatch: 402 rows are a small
    grid (fewer than 128 tiles of 256x256), so every GEMM runs on the 128x128 kernel; CH_GEMM_PP_MIN_K=256 pins the choice by K
    alone: every GEMM of the chain on the ping-pong kernel (qkv / fc1 / down with the LN-fold epilogues, up with the residual +
    statistics epilogue)."""
    from concepthash_amd import _lib
    sd, z = load_fixture("encode_n201")
    heads = int(z["meta/heads"])
    if pp_min_k:
        monkeypatch.setenv("CH_GEMM_PP_MIN_K", str(pp_min_k))
    monkeypatch.setenv("CH_STREAMS", "1")        # one launch chain, so that the dispatch counts below are per encode call
    enc = _encoder(sd, heads, max_batch=2)
    lib = _lib.load()
    n_small0, n_pp0 = lib.ch_debug_gemm_dispatch_count(0), lib.ch_debug_gemm_dispatch_count(1)
    x = fixture_images(z).to(dev)
    want = ("codes", "packed", "logits_cont", "logits_bin", "logits_concept", "hash_features", "image_features", "concept_attn")
    out = enc.encode(x, want=want)
    torch.cuda.synchronize()
    n_small, n_pp = lib.ch_debug_gemm_dispatch_count(0) - n_small0, lib.ch_debug_gemm_dispatch_count(1) - n_pp0
    print(f"pp_min_k={pp_min_k}: {n_pp} GEMMs on the 256x256 ping-pong kernel, {n_small} on the 128x128 kernel")
    assert n_pp + n_small == 1 + 2 * 8           # patch + 8 GEMMs per layer
    if pp_min_k:
        assert n_small == 0                      # the whole chain ran on the ping-pong kernel
    else:
        assert n_pp == 0 and n_small == 17       # small grid: the dispatcher prefers 128x128 tiles (gemm_bf16.hip)
    for key, tol in (("codes", 2e-2), ("hash_features", 2e-2), ("logits_cont", 2e-2), ("logits_bin", 3e-2),
                     ("logits_concept", 2e-2), ("image_features", 2e-2)):
        got, ref = out[key].cpu(), torch.from_numpy(z["out/" + key])
        e = _rel_err(got, ref)
        print(f"n201 {key}: rel err vs reference golden {e:.2e}")
        assert got.shape == ref.shape and e < tol, key
    for layer, key, tol in ((0, "h0", 1e-2), (1, "h1", 2e-2), (2, "h_last", 2e-2)):
        h = enc.hidden_states(x, layer).cpu()
        e = _rel_err(h, torch.from_numpy(z["out/" + key]))
        print(f"n201 hidden state after {layer} layers: {e:.2e}")
        assert e < tol, key
    # KB = 7 attention (201 keys padded to 224, boundary masking): the concept tokens' last-layer attention rows
    ca, ref = out["concept_attn"].cpu(), torch.from_numpy(z["out/concept_attn_last"])
    assert ca.shape == ref.shape == (2, heads, 4, 196)
    assert float((ca - ref).abs().max()) < 2e-3 and torch.allclose(ca.sum(-1), ref.sum(-1), atol=2e-3)
    from oracle import hamming_oracle as ho
    assert np.array_equal(out["packed"].cpu().numpy().view(np.uint64), ho.pack(out["codes"].cpu().numpy()))
    refc = torch.from_numpy(z["out/codes"])
    flips = (out["codes"].cpu() > 0) != (refc > 0)
    print(f"n201: {int(flips.sum())} / {flips.numel()} bits differ from the reference")
    assert bool((refc[flips].abs() < 2e-2 * refc.pow(2).mean().sqrt()).all())


@pytest.mark.parametrize("pp_min_k", [None, 256])
def test_cache_policy_instances_are_bit_identical(dev, monkeypatch, pp_min_k):
    """The dispatcher picks, by tensor size, GEMM instances whose fp32-residual read-modify-write (gemm_bf16_kernel<EPI, true>) or bf16
    output store (gemm_pp_kernel<EPI, 0, 0, 2>) is non-temporal (gemm_bf16.hip: resid_nt_choice / out_nt_choice; at BASELINE
    sizes they are the default: 158 MB residual, 316 / 237 MB fc1 / qkv outputs).  A cache policy is not arithmetic: forced on
    (CH_RESID_NT=1, CH_NT_OUT=1) on the reference's 201-token fixture every output is BIT-identical to the default instances', under
    both dispatches -- all 128x128 (the adapter up-projection's residual instance) and all 256x256 (qkv / fc1 output instance)."""
    from concepthash_amd import _lib
    sd, z = load_fixture("encode_n201")
    heads = int(z["meta/heads"])
    if pp_min_k:
        monkeypatch.setenv("CH_GEMM_PP_MIN_K", str(pp_min_k))
    monkeypatch.setenv("CH_STREAMS", "1")
    monkeypatch.setenv("CH_RESID_NT", "0")
    monkeypatch.setenv("CH_NT_OUT", "0")
    enc = _encoder(sd, heads, max_batch=2)
    lib = _lib.load()
    x = fixture_images(z).to(dev)
    want = ("codes", "packed", "logits_cont", "logits_bin", "logits_concept", "hash_features", "image_features", "concept_attn")
    count = lambda: (lib.ch_debug_gemm_dispatch_count(2), lib.ch_debug_gemm_dispatch_count(3))
    c0 = count()
    base = {k: v.clone() for k, v in enc.encode(x, want=want).items()}
    h_base = enc.hidden_states(x, 2).clone()
    torch.cuda.synchronize()
    assert count() == c0                                     # forced off: default instances only
    enc.set_option("resid_nt", 1)
    enc.set_option("nt_out", 1)
    out = enc.encode(x, want=want)
    h = enc.hidden_states(x, 2)
    torch.cuda.synchronize()
    c1 = count()
    n_resid, n_out = c1[0] - c0[0], c1[1] - c0[1]
    print(f"pp_min_k={pp_min_k}: {n_resid} launches of the non-temporal residual instance, {n_out} of the non-temporal output instance")
    if pp_min_k:
        assert n_out >= 2 * 2 * 2                            # qkv + fc1 of 2 layers, two calls
    else:
        assert n_resid >= 2 * 2 * 2 and n_out == 0           # two adapters of 2 layers, two calls; no 256x256 launch
    for k in want:
        assert torch.equal(out[k], base[k]), k
    assert torch.equal(h, h_base)
    enc.close()


def test_non_pretrain_resolution_against_reference_golden(dev):
    """inputs other than the pretrain resolution.  tests/golden/encode_interp.npz = the reference model
    pretrained at 64 px evaluated at 96 px (interpolate_pos_encoding, models/arch/coop.py:429-450); here the interpolated table
    is folded when the engine is built (ConceptHashEncoder(image_size=96)), and models.arch.coop picks the engine by input size."""
    sd, z = load_fixture("encode_interp")
    heads = int(z["meta/heads"])
    x = torch.from_numpy(z["in/images"]).to(dev)
    enc = _encoder(sd, heads, max_batch=2, image_size=96)
    assert enc.ntok == 1 + 36 + 4 and enc.pretrain_image_size == 64
    out = enc.encode(x, want=("codes", "hash_features", "logits_cont", "image_features"))
    torch.cuda.synchronize()
    for key in ("codes", "hash_features", "logits_cont", "image_features"):
        e = _rel_err(out[key].cpu(), torch.from_numpy(z["out/" + key]))
        print(f"96 px on a 64 px model, {key}: rel err vs reference golden {e:.2e}")
        assert e < 2e-2, key
    h0 = enc.hidden_states(x, 0).cpu()
    assert _rel_err(h0, torch.from_numpy(z["out/h0"])) < 1e-2
    with pytest.raises(ValueError):
        _encoder(sd, heads, max_batch=2).encode(x)              # the 64 px engine refuses 96 px inputs loudly
    with pytest.raises(ValueError):
        _encoder(sd, heads, image_size=100)                     # not a multiple of the patch size


# ---- (c) + (g) ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg_name,batch,nbit,nclass", [("vit_l14", 1, 128, 555), ("vit_s16", 2, 16, 200), ("vit_b16", 2, 64, 200)])
def test_full_depth_against_fp32_and_fold_emulating_oracle(dev, cfg_name, batch, nbit, nclass):
    """All layers of the BASELINE.json model sizes (L/14: 24, S/16: 12, B/16: 12).  Against the fp32 oracle: bf16 operand
    rounding through every layer (bound 4e-2 max-abs / RMS, 1e-2 RMS / RMS).  Against the oracle that rounds at the SAME
    points as the default chain (emulate_fold): 2.5e-2 max-abs / RMS and 1e-2 RMS / RMS -- measured up to 2.2e-2 / 7.8e-3 at
    24 layers, the same as the distance between the two rounding-emulating oracles themselves (tools/error_growth.py)."""
    from oracle import encoder_oracle as eo
    cfg = dict(eo.CONFIGS[cfg_name])
    sd = eo.synthetic_state_dict(cfg, nbit=nbit, nclass=nclass)
    x = eo.synthetic_images(batch, cfg["image"])
    enc = _encoder(sd, cfg["heads"], max_batch=2)
    out = enc.encode(x.to(dev), want=("codes", "hash_features", "logits_cont"))
    torch.cuda.synchronize()
    ref = eo.encode(sd, x, heads=cfg["heads"], with_pooled=False)
    emu = eo.encode(sd, x, heads=cfg["heads"], with_pooled=False, emulate_bf16=True, emulate_fold=True)
    for key in ("codes", "hash_features", "logits_cont"):
        got = out[key].cpu()
        e_ref, e_emu, r_ref = _rel_err(got, ref[key]), _rel_err(got, emu[key]), _rms_err(got, ref[key])
        print(f"{cfg_name} x {cfg['L']} layers, {key}: vs fp32 oracle {e_ref:.2e} (rms {r_ref:.2e}), "
              f"vs fold-emulating oracle {e_emu:.2e}; emulating vs fp32 {_rel_err(emu[key], ref[key]):.2e}")
        assert e_ref < 4e-2 and r_ref < 1e-2, key
        assert e_emu < 2.5e-2 and _rms_err(got, emu[key]) < 1e-2, key
    flips = (out["codes"].cpu() > 0) != (ref["codes"] > 0)
    assert bool((ref["codes"][flips].abs() < 4e-2 * ref["codes"].pow(2).mean().sqrt()).all())


# ---- (d) ---------------------------------------------------------------------------------------------------------
def test_batch_256_rows_are_bit_equal_to_an_oracle_checked_small_batch(dev):
    """ch_encode at the bench shape (ViT-B/16 x 12, batch 256: 201 row tiles, n-grouped tile order, 113 launches): the first,
    middle and last image of the big batch equal, bit for bit, the same three images encoded as a batch of 3, and that small
    batch is checked against the fp32 oracle."""
    from oracle import encoder_oracle as eo
    cfg = dict(eo.CONFIGS["vit_b16"])
    sd = eo.synthetic_state_dict(cfg, nbit=64, nclass=200)
    x = eo.synthetic_images(256, cfg["image"], seed=42)
    pick = [0, 127, 255]
    enc = _encoder(sd, cfg["heads"], max_batch=256)
    want = ("codes", "packed", "logits_cont", "hash_features")
    big = enc.encode(x.to(dev).to(torch.bfloat16), want=want)
    small = enc.encode(x[pick].to(dev).to(torch.bfloat16), want=want)
    torch.cuda.synchronize()
    assert torch.isfinite(big["codes"]).all()
    for k in want:
        assert torch.equal(big[k][pick], small[k]), k
    ref = eo.encode(sd, x[pick].to(torch.bfloat16).float(), heads=cfg["heads"], with_pooled=False)
    e = _rel_err(small["codes"].cpu(), ref["codes"])
    print(f"batch 256 vs batch 3: bit-equal; batch 3 vs fp32 oracle {e:.2e}")
    assert e < 4e-2
    # the batch is not 256 copies of one answer (seeded noise images through random weights still give many distinct codes)
    assert len({tuple(r) for r in big["packed"].cpu().numpy().tolist()}) > 32


# ---- (e) ---------------------------------------------------------------------------------------------------------
def test_map_delta_and_bit_flip_rate_against_fp32_oracle_codes(dev):
    """North-star tolerance (BASELINE.json): mAP@all of the HIP path's codes vs the fp32 reference restatement's on identical
    inputs; ViT-B/16 x 12 layers, 64-bit codes, 16 classes of class-structured synthetic images (prototype + noise), a quarter
    of the images = queries, the rest = gallery; mAP from the integer oracle on both code sets.

    What a bf16-operand encode does to mAP is decided by (i) how much pre-sign code mass sits within its error of zero (those
    bits flip) and (ii) how much margin the ranking has.  No trained checkpoint exists offline, so three regimes are measured:
      A  high margin: clean classes, hash_fc + BatchNorm FITTED (closed-form ridge regression of random class codewords on the
         gallery's fp32 hash features -- what training does to the head).  mAP ~ 1: asserted |delta mAP@all| < 1e-3.
      B  moderate margin: noisier classes, fitted head, mAP ~ 0.85-0.9 (the range of the paper's CUB numbers).  The linear
         probe's codes are unimodal around zero (1.4 % of them within 2 % of zero -- a trained model's quantisation loss,
         models/loss/coop.py, empties exactly that region), so this OVERSTATES a trained model's flips: reported, bounded at
         twice the measured value (3e-3).  The TRAINED head on the same noisy classes is tests/test_parity_r3_gpu.py.
      C  the untrained random head on the same images: near-chance ranking, the worst case: reported, bounded.
    Every flipped bit has |fp32 code| below the measured code error (checked), i.e. no bit flips for any other reason."""
    from oracle import encoder_oracle as eo
    from oracle import hamming_oracle as ho
    cfg = dict(eo.CONFIGS["vit_b16"])
    ncls, nbit, Q = 16, 64, 4
    sd = eo.synthetic_state_dict(cfg, nbit=nbit, nclass=ncls)
    torch.set_num_threads(min(16, torch.get_num_threads()))

    def dataset(per, mix, seed):
        g = torch.Generator().manual_seed(seed)
        proto = torch.randn(ncls, 3, cfg["image"], cfg["image"], generator=g)
        labels = torch.arange(ncls).repeat_interleave(per)
        noise = torch.randn(ncls * per, 3, cfg["image"], cfg["image"], generator=g)
        x = (mix[0] * proto[labels] + mix[1] * noise).to(torch.bfloat16).float()
        perm = torch.randperm(ncls * per, generator=g)
        x, labels = x[perm], labels[perm]
        hf = torch.cat([eo.encode(sd, x[i:i + 32], heads=cfg["heads"], with_pooled=False)["hash_features"]
                        for i in range(0, x.shape[0], 32)])                               # fp32 oracle, [n, Q, D]
        return x, labels, hf

    def fitted_head(hf, labels, nq, lam_f=0.1):
        codeword = torch.randn(ncls, nbit, generator=torch.Generator().manual_seed(7)).sign()
        feats = (hf + sd["hash_pe"].float())[nq:].reshape(-1, cfg["D"]).double()                          # [(gallery*Q), D]
        target = codeword[labels[nq:]].reshape(-1, Q, nbit // Q).reshape(-1, nbit // Q).double()          # concept-major bits
        A = feats - feats.mean(0, keepdim=True)
        lam = lam_f * float(A.pow(2).sum()) / cfg["D"]                   # lam_f x the mean diagonal of A^T A
        Wt = torch.linalg.solve(A.t() @ A + lam * torch.eye(cfg["D"], dtype=torch.float64),
                                A.t() @ (target - target.mean(0, keepdim=True)))
        sd_fit = dict(sd)
        sd_fit["hash_fc.weight"] = Wt.t().float().contiguous()                                            # [nbit/Q, D]
        v = ((hf + sd["hash_pe"].float()) @ sd_fit["hash_fc.weight"].t()).reshape(hf.shape[0], -1)[nq:]  # pre-BN, gallery
        sd_fit["hash_bn.running_mean"], sd_fit["hash_bn.running_var"] = v.mean(0), v.var(0, unbiased=False)
        sd_fit["hash_bn.weight"], sd_fit["hash_bn.bias"] = torch.ones(nbit), torch.zeros(nbit)
        return sd_fit

    def measure(tag, sd_head, x, labels, hf, nq):
        enc = _encoder(sd_head, cfg["heads"], max_batch=256)
        hip = enc.encode(x.to(dev), want=("codes",))["codes"].cpu()
        torch.cuda.synchronize()
        enc.close()
        ref = eo.hash_head(sd_head, hf)            # the head of the fp32 oracle on its own hash features (coop.py:544-559)
        lab = labels.numpy().astype(np.int32)
        flips = (hip > 0) != (ref > 0)
        res = {}
        for name, codes in (("hip", hip), ("fp32", ref)):
            pk = ho.pack(codes.numpy())
            res[name] = ho.mean_ap(pk[:nq], pk[nq:], lab[:nq], lab[nq:])["mAP"]
        d = abs(res["hip"] - res["fp32"])
        rms = ref.pow(2).mean().sqrt()
        near0 = float((ref.abs() < 0.02 * rms).float().mean())
        err = float((hip - ref).abs().max())
        print(f"[{tag}] mAP@all hip {res['hip']:.6f} vs fp32 oracle {res['fp32']:.6f}: |delta| {d:.2e}; bit-flip rate "
              f"{float(flips.float().mean()):.3e} ({int(flips.sum())} / {flips.numel()}); codes max err / rms {err / float(rms):.2e}; "
              f"fraction of fp32 codes within 2 % of zero {near0:.2e}")
        assert bool((ref[flips].abs() <= err).all())
        return d, res["fp32"], float(flips.float().mean()), err / float(rms)

    # ---- A: high margin
    x, labels, hf = dataset(12, (0.8, 0.6), 2024)
    d, m, rate, e = measure("A fitted head, clean classes", fitted_head(hf, labels, 48), x, labels, hf, 48)
    assert m > 0.95 and d < 1e-3 and e < 0.15
    # ---- B, C: moderate margin / untrained head on 512 noisier images
    x, labels, hf = dataset(32, (0.5, 0.87), 2025)
    d, m, rate, e = measure("B fitted head, noisy classes", fitted_head(hf, labels, 128), x, labels, hf, 128)
    assert 0.6 < m < 0.99 and d < 3e-3 and rate < 1e-2 and e < 0.15      # measured 1.5e-3 (round 2): bound = 2x, not 17x
    d, m, rate, e = measure("C random head, noisy classes", sd, x, labels, hf, 128)
    assert 1.0 / ncls < m < 1.0 and d < 6e-3 and rate < 5e-3 and e < 4e-2  # measured 2.8e-3 (round 2): bound = 2x


# ---- (f) ---------------------------------------------------------------------------------------------------------
def test_outlier_residual_channels_through_full_depth(dev):
    """Real CLIP towers carry a few residual channels two orders of magnitude above the rest.  Here 6 channels of the
    residual stream are driven x100 (pre-LN gamma, the rows of out_proj / fc2 / up_proj that write them) with the LayerNorm
    gammas of every consumer set to 1/100 on those channels, through 12 ViT-B/16 layers: the single-pass folded statistics
    (E[x^2] - mean^2 in fp32 on the raw bf16 rows) stay within the encode tolerance of the fp32 oracle."""
    from oracle import encoder_oracle as eo
    cfg = dict(eo.CONFIGS["vit_b16"])
    sd = eo.synthetic_state_dict(cfg, nbit=64, nclass=10)
    VM = "backbone.vision_model."
    ch = torch.tensor([5, 77, 200, 391, 512, 760])
    sd = {k: v.clone() for k, v in sd.items()}
    sd[VM + "pre_layrnorm.weight"][ch] *= 100.0
    for i in range(cfg["L"]):
        pre = VM + f"encoder.layers.{i}."
        for name in ("self_attn.out_proj", "mlp.fc2", "adapt_mlp_1.up_proj", "adapt_mlp_2.up_proj"):
            sd[pre + name + ".weight"][ch, :] *= 100.0
            sd[pre + name + ".bias"][ch] *= 100.0
        for name in ("layer_norm1", "layer_norm2", "adapt_mlp_1.adapter_layer_norm", "adapt_mlp_2.adapter_layer_norm"):
            sd[pre + name + ".weight"][ch] *= 0.01
    sd["hash_fc.weight"][:, ch] *= 0.01
    x = eo.synthetic_images(2, cfg["image"])
    st = {}
    ref = eo.encode(sd, x, heads=cfg["heads"], with_pooled=False, stages=st)
    hl = st[f"h{cfg['L']}"]
    ratio = float(hl[..., ch].abs().mean() / hl.abs().mean())
    print(f"outlier channels carry {ratio:.1f}x the mean magnitude of the final residual stream")
    assert ratio > 15
    enc = _encoder(sd, cfg["heads"], max_batch=2)
    out = enc.encode(x.to(dev), want=("codes", "hash_features"))
    hid = enc.hidden_states(x.to(dev), cfg["L"]).cpu()
    torch.cuda.synchronize()
    e_codes, r_codes = _rel_err(out["codes"].cpu(), ref["codes"]), _rms_err(out["codes"].cpu(), ref["codes"])
    keep = torch.ones(cfg["D"], dtype=torch.bool)
    keep[ch] = False
    e_rest = _rel_err(hid[..., keep], hl[..., keep])
    e_out = _rel_err(hid[..., ch], hl[..., ch])          # max-abs error / RMS of the outlier channels themselves
    print(f"outlier weights: codes {e_codes:.2e} (rms {r_codes:.2e}); final residual, ordinary channels {e_rest:.2e}, "
          f"outlier channels {e_out:.2e}")
    assert e_codes < 4e-2 and r_codes < 1e-2
    assert e_rest < 4e-2 and e_out < 4e-2


# ---- (h) the headline model against REFERENCE output (weights rebuilt from a seed) ----------------------------------------------
@pytest.mark.parametrize("config", ["vit_b16", "vit_s16", "vit_l14"])
def test_full_depth_against_the_reference_itself(dev, config):
    """tests/golden/seeded_<config>.npz (oracle/gen_seeded_golden.py): the reference's own LGHWithFixedPrompt at ViT-B/16 x 12 layers
    (201 tokens), ViT-S/16 x 12 and ViT-L/14 x 24 (261 tokens), on weights rebuilt here from the seed (checksums in the fixture).  Encode: the HIP path vs reference output, same
    bounds as against the fp32 oracle.  Training: gradient signatures (norm, projection on a seeded direction) of all 168 adapter
    tensors for a seeded cotangent on hash_features -- the reference's autograd vs ch_train_backward."""
    from concepthash_amd.training import ADAPTER_FIELDS, TrainEngine, adapters_from_state_dict
    from conftest import GOLDEN
    from oracle import seeded as gen
    z = np.load(os.path.join(GOLDEN, f"seeded_{config}.npz"))
    cfg, sd, x, cot = gen.seeded_inputs(config)
    for k in ("backbone.vision_model.encoder.layers.7.mlp.fc1.weight", "hash_fc.weight"):
        assert abs(float(sd[k].double().sum()) - z["chk/" + k][0]) <= 1e-9 * max(1.0, abs(z["chk/" + k][0]))
    enc = _encoder(sd, cfg["heads"], max_batch=2)
    out = enc.encode(x.to(dev), want=("codes", "hash_features", "logits_cont", "logits_bin", "logits_concept", "image_features"))
    torch.cuda.synchronize()
    for key in ("codes", "hash_features", "logits_cont", "logits_bin", "logits_concept", "image_features"):
        ref = torch.from_numpy(z["out/" + key])
        e, r = _rel_err(out[key].cpu(), ref), _rms_err(out[key].cpu(), ref)
        print(f"seeded {config} {key}: vs REFERENCE output {e:.2e} (rms {r:.2e})")
        assert e < 4e-2 and r < 1e-2, key
    ref_codes = torch.from_numpy(z["out/codes"])
    flips = (out["codes"].cpu() > 0) != (ref_codes > 0)
    assert bool((ref_codes[flips].abs() < 4e-2 * ref_codes.pow(2).mean().sqrt()).all())
    enc.close()
    eng = TrainEngine(sd, adapters_from_state_dict(sd, cfg["L"], cfg["D"], cfg["b"]), heads=cfg["heads"], max_batch=2, device=dev)
    from oracle import encoder_oracle as eo
    ctx = eo.concept_tokens(sd, 8)[0]
    eng.forward(x.to(dev), ctx.to(dev))
    eng.backward(cot.to(dev))
    torch.cuda.synchronize()
    it = iter(eng._views)
    worst, worst_scale = (0.0, ""), (0.0, "")
    # `scale` gradients are scalars, ds = sum over every row and column of dH * up(g): a sum of ~300k terms of both signs, whose bf16
    # operand noise does not shrink with the result -- judged against the largest |ds| of the 24 adapters, like the sibling rule of
    # tests/test_train_gpu.py; every other tensor by its own norm
    scale_ref = max(abs(float(z[k][0])) for k in z.files if k.startswith("sig/") and k.endswith(".scale"))
    for l in range(cfg["L"]):
        for a in (1, 2):
            for field in ADAPTER_FIELDS:
                k = f"backbone.vision_model.encoder.layers.{l}.adapt_mlp_{a}.{field}"
                norm, proj = gen.signature(k, next(it)[1])
                rn, rp = z["sig/" + k]
                if field == "scale":
                    err = abs(norm - rn) / scale_ref
                    worst_scale = max(worst_scale, (err, k))
                    assert err < 3e-2, (k, norm, rn, scale_ref)
                    continue
                err = max(abs(norm - rn), abs(proj - rp)) / rn
                worst = max(worst, (err, k))
                assert err < 5e-2, (k, norm, rn, proj, rp)
    print(f"seeded {config} training: worst scale-gradient error / largest |ds|: %.2e (%s)" % worst_scale)
    print(f"seeded {config} training: worst adapter gradient signature error vs the reference's autograd: %.2e (%s)" % worst)
    eng.close()
