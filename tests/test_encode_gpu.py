"""GPU parity: the HIP encoder (through the C-ABI) against (a) golden vectors produced by the reference's own model
code and (b) the CPU oracle on seeded synthetic weights.

Tolerances (bf16 GEMM operands, fp32 accumulation, fp32 residual stream; stated per comparison):
  * vs the bf16-rounding-emulating oracle: what is left is accumulation order and exp/erf implementation;
  * vs the fp32 oracle / reference golden: bf16 operand rounding through L layers.
Errors are measured relative to the RMS of the compared tensor, and the measured value is printed.
"""
import numpy as np
import pytest
import torch

from conftest import load_fixture

pytestmark = pytest.mark.gpu


def _rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.double() - b.double()).abs().max() / b.double().pow(2).mean().sqrt().clamp_min(1e-12))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _encoder(sd, heads, **kw):
    from concepthash_amd.encoder import ConceptHashEncoder
    return ConceptHashEncoder(sd, heads=heads, **kw)


def test_golden_reference_vectors(dev):
    """tests/golden/encode_hd64.npz was produced by the reference's LGHWithFixedPrompt (oracle/gen_golden.py)."""
    from oracle import encoder_oracle as eo
    sd, z = load_fixture("encode_hd64")
    heads = int(z["meta/heads"])
    enc = _encoder(sd, heads, max_batch=4)
    x = torch.from_numpy(z["in/images"]).to(dev)
    out = enc.encode(x, want=("codes", "packed", "logits_cont", "logits_bin", "logits_concept", "hash_features",
                              "image_features"))
    torch.cuda.synchronize()
    emu = eo.encode(sd, torch.from_numpy(z["in/images"]), heads=heads, emulate_bf16=True)
    for key, tol_ref, tol_emu in (("codes", 2e-2, 5e-3), ("hash_features", 2e-2, 5e-3), ("logits_cont", 2e-2, 5e-3),
                                  ("logits_bin", 3e-2, 1e-2), ("logits_concept", 2e-2, 5e-3),
                                  ("image_features", 2e-2, 1e-2)):
        got = out[key].cpu()
        ref = torch.from_numpy(z["out/" + key])
        e_ref = _rel_err(got, ref)
        e_emu = _rel_err(got, emu[key]) if key in emu else float("nan")
        print(f"{key}: rel err vs reference golden {e_ref:.2e}, vs bf16-emulating oracle {e_emu:.2e}")
        assert got.shape == ref.shape
        assert e_ref < tol_ref, key
        if key in emu:
            assert e_emu < tol_emu, key
    # hidden-state taps: layer 0 (embeddings + concept tokens + pre-LN) is fp32 except the bf16 patch GEMM
    h0 = enc.hidden_states(x, 0).cpu()
    assert _rel_err(h0, torch.from_numpy(z["out/h0"])) < 1e-2
    h1 = enc.hidden_states(x, 1).cpu()
    assert _rel_err(h1, torch.from_numpy(z["out/h1"])) < 2e-2
    hl = enc.hidden_states(x, 2).cpu()
    assert _rel_err(hl, torch.from_numpy(z["out/h_last"])) < 2e-2
    # packed bits == sign of the HIP codes, bit for bit (integer contract)
    from oracle import hamming_oracle as ho
    assert np.array_equal(out["packed"].cpu().numpy().view(np.uint64), ho.pack(out["codes"].cpu().numpy()))
    # bits that differ from the fp32 reference are only those with |code| within the numeric error
    refc = torch.from_numpy(z["out/codes"])
    flips = (out["codes"].cpu() > 0) != (refc > 0)
    assert bool((refc[flips].abs() < 2e-2 * refc.pow(2).mean().sqrt()).all())


@pytest.mark.parametrize("cfg_name,L,batch,nbit,nclass", [("vit_s16", 2, 3, 16, 200), ("vit_b16", 12, 2, 64, 200),
                                                         ("vit_b32", 3, 5, 64, 196), ("vit_l14", 2, 2, 128, 555)])
def test_synthetic_configs_against_oracle(dev, cfg_name, L, batch, nbit, nclass):
    """BASELINE.json configs (dims), seeded random weights, depth cut where the CPU oracle would be slow."""
    from oracle import encoder_oracle as eo
    cfg = dict(eo.CONFIGS[cfg_name])
    cfg["L"] = L
    sd = eo.synthetic_state_dict(cfg, nbit=nbit, nclass=nclass)
    x = eo.synthetic_images(batch, cfg["image"])
    enc = _encoder(sd, cfg["heads"], max_batch=4)
    out = enc.encode(x.to(dev), want=("codes", "packed", "logits_cont", "logits_bin", "logits_concept", "hash_features"))
    torch.cuda.synchronize()
    ref = eo.encode(sd, x, heads=cfg["heads"], with_pooled=False)
    emu = eo.encode(sd, x, heads=cfg["heads"], emulate_bf16=True, with_pooled=False)
    for key in ("codes", "hash_features", "logits_cont", "logits_concept"):
        e_ref, e_emu = _rel_err(out[key].cpu(), ref[key]), _rel_err(out[key].cpu(), emu[key])
        print(f"{cfg_name} {key}: rel err vs fp32 oracle {e_ref:.2e}, vs bf16-emulating oracle {e_emu:.2e}")
        assert e_ref < 4e-2, key     # max-abs error / RMS, bf16 operand rounding through up to 12 layers
        assert e_emu < 2.5e-2, key   # rounding-emulating oracle: residual = accumulation order, exp/erf, chaotic growth
        rms_err = float((out[key].cpu() - ref[key]).pow(2).mean().sqrt() / ref[key].pow(2).mean().sqrt())
        assert rms_err < 1e-2, (key, rms_err)
    flips = (out["codes"].cpu() > 0) != (ref["codes"] > 0)
    print(f"{cfg_name}: {int(flips.sum())} / {flips.numel()} bits differ from the fp32 oracle")
    assert bool((ref["codes"][flips].abs() < 4e-2 * ref["codes"].pow(2).mean().sqrt()).all())
    assert enc.flops_per_image > 0


def test_batching_dtype_and_determinism(dev):
    from oracle import encoder_oracle as eo
    cfg = dict(eo.CONFIGS["vit_s16"])
    cfg["L"] = 2
    sd = eo.synthetic_state_dict(cfg, nbit=64, nclass=10)
    x = eo.synthetic_images(11, cfg["image"]).to(dev)
    enc = _encoder(sd, cfg["heads"], max_batch=4)            # 11 images -> chunks 4,4,3
    a = enc.encode(x)
    b = enc.encode(x)
    torch.cuda.synchronize()
    assert torch.equal(a["codes"], b["codes"]) and torch.equal(a["packed"], b["packed"])     # run-to-run identical
    one = torch.cat([enc.encode(x[i:i + 1])["codes"] for i in range(11)])
    assert torch.equal(one, a["codes"])                      # an image's code does not depend on its batch-mates
    xb = x.to(torch.bfloat16)
    c = enc.encode(xb)
    d = enc.encode(xb.float())                               # bf16 images == the same values given as fp32
    assert torch.equal(c["codes"], d["codes"])
    # a state dict that already sits on the GPU (what `model.to("cuda")` leaves) is ingested device -> device: the same engine
    on_gpu = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sd.items()}
    enc_gpu = _encoder(on_gpu, cfg["heads"], max_batch=4)
    e = enc_gpu.encode(x)
    assert all(torch.equal(a[k], e[k]) for k in a)
    enc_gpu.close()
    enc.close()


@pytest.mark.parametrize("cfg_name,pp_min_k", [("vit_s16", None), ("vit_b16", 256)])
def test_forty_encodes_over_two_launch_chains_are_bit_identical(dev, monkeypatch, cfg_name, pp_min_k):
    """Run-to-run determinism under CONCURRENT launch chains (two streams share the CUs: the condition under which the
    `v_pk_fma_f32 ... op_sel:[0,1,0]` form of the LN-fold epilogue returned wrong quarter-wave lanes in round 3 -- 40 of 40 calls
    differed then).  Two layers, 11 images in chunks of 4 (two chains of 2 images each), codes and the
    hidden state after every layer, 40 calls each, all bit-identical to the first: ViT-S/16 on the 128x128 GEMM kernel (the
    dispatcher's choice at this size) and ViT-B/16 with CH_GEMM_PP_MIN_K=256, which sends qkv / out_proj / fc1 / fc2 / the adapter
    up-projections to the 256x256 ping-pong kernel."""
    from oracle import encoder_oracle as eo
    monkeypatch.setenv("CH_STREAMS", "2")
    if pp_min_k:
        monkeypatch.setenv("CH_GEMM_PP_MIN_K", str(pp_min_k))
    cfg = dict(eo.CONFIGS[cfg_name])
    cfg["L"] = 2
    sd = eo.synthetic_state_dict(cfg, nbit=64, nclass=10)
    x = eo.synthetic_images(11, cfg["image"]).to(dev)
    enc = _encoder(sd, cfg["heads"], max_batch=4)
    ref = enc.encode(x)["codes"].clone()
    hid = [enc.hidden_states(x[:4], layer).clone() for layer in (1, 2)]
    torch.cuda.synchronize()
    bad = 0
    for _ in range(40):
        bad += int(not torch.equal(enc.encode(x)["codes"], ref))
        for i, layer in enumerate((1, 2)):
            bad += int(not torch.equal(enc.hidden_states(x[:4], layer), hid[i]))
    torch.cuda.synchronize()
    assert bad == 0, f"{bad} of 120 calls differ from the first"
    enc.close()


def test_the_benchmark_batch_is_bit_identical_across_policies_chains_and_repeats(dev, monkeypatch):
    """BASELINE.json's measured configuration at its full size -- ViT-B/16 x 12 layers, 64 bit, 256 images -- through size-independent
    properties (an oracle run of this size takes minutes): the dispatcher's default here is the non-temporal residual instance for
    the 24 up-projections and the non-temporal output instance for qkv / fc1 (counted through the dispatch taps); the same batch
    with both policies forced off, as one launch chain instead of two, and twice more under the default must give the SAME codes,
    packed words and pooled outputs bit for bit; and the packed words are the signs of the codes."""
    from concepthash_amd import _lib
    from oracle import encoder_oracle as eo
    from oracle import hamming_oracle as ho
    cfg = dict(eo.CONFIGS["vit_b16"])
    sd = eo.synthetic_state_dict(cfg, nbit=64, nclass=200)
    x = eo.synthetic_images(256, cfg["image"], seed=3).to(dev).to(torch.bfloat16)
    lib = _lib.load()
    want = ("codes", "packed", "logits_cont", "hash_features")
    count = lambda: (lib.ch_debug_gemm_dispatch_count(2), lib.ch_debug_gemm_dispatch_count(3))
    enc = _encoder(sd, cfg["heads"], max_batch=256)
    c0 = count()
    ref = {k: v.clone() for k, v in enc.encode(x, want=want).items()}
    torch.cuda.synchronize()
    c1 = count()
    # two chains x (2 up-projections x 12 layers, the last layer's on compact rows falls below the size rule) ; qkv + fc1 per layer
    assert c1[0] - c0[0] >= 2 * 22 and c1[1] - c0[1] >= 2 * 22, (c0, c1)
    for _ in range(2):
        again = enc.encode(x, want=want)
        for k in want:
            assert torch.equal(again[k], ref[k]), k
    enc.set_option("resid_nt", -1)                           # per-handle options, no environment: both policies forced off
    enc.set_option("nt_out", -1)
    assert enc.get_option("resid_nt") == -1 and enc.get_option("streams") == 2
    c2 = count()
    off = enc.encode(x, want=want)
    torch.cuda.synchronize()
    assert count() == c2                                     # forced off: the default instances only
    for k in want:
        assert torch.equal(off[k], ref[k]), k
    enc.close()
    one = _encoder(sd, cfg["heads"], max_batch=256, options={"streams": 1})
    single = one.encode(x, want=want)
    torch.cuda.synchronize()
    for k in want:
        assert torch.equal(single[k], ref[k]), k
    one.close()
    assert np.array_equal(ref["packed"].cpu().numpy().view(np.uint64), ho.pack(ref["codes"].cpu().numpy()))


def test_errors_are_loud(dev):
    from oracle import encoder_oracle as eo
    cfg = dict(eo.CONFIGS["vit_s16"])
    cfg["L"] = 1
    sd = eo.synthetic_state_dict(cfg, nbit=16, nclass=10)
    enc = _encoder(sd, cfg["heads"], max_batch=2)
    with pytest.raises(ValueError):
        enc.encode(torch.zeros(1, 3, 64, 64, device=dev))
    with pytest.raises(TypeError):
        enc.encode(torch.zeros(1, 3, 224, 224, device=dev, dtype=torch.float16))
    with pytest.raises(KeyError):
        enc.encode(torch.zeros(1, 3, 224, 224, device=dev), want=("nope",))
    bad = dict(sd)
    del bad["hash_fc.weight"]
    with pytest.raises((RuntimeError, KeyError)):
        _encoder(bad, cfg["heads"])
    bad = dict(sd)
    bad["hash_bn.weight"] = torch.zeros(3)
    with pytest.raises(RuntimeError, match="hash_bn.weight"):
        _encoder(bad, cfg["heads"])


def test_fused_adapter_kernel_matches_unfused_chain(dev, monkeypatch):
    """adapter_fused.hip (opt-in, CH_FUSED_ADAPTER=1): LayerNorm folded into the down projection, bottleneck in LDS.
    Same inputs, different rounding points -> compare against the oracle with the encode tolerances, and against the
    default three-launch chain."""
    from concepthash_amd import _lib
    from oracle import encoder_oracle as eo
    cfg = dict(eo.CONFIGS["vit_b16"])
    cfg["L"] = 3
    sd = eo.synthetic_state_dict(cfg, nbit=64, nclass=10)
    x = eo.synthetic_images(3, cfg["image"])
    if not _lib.load().ch_debug_experiments_built():
        with pytest.raises(RuntimeError, match="not part of this build"):     # the product library refuses loudly
            _encoder(sd, cfg["heads"], max_batch=4, options={"fused_adapter": 1})
        pytest.skip("adapter_fused.hip is an experiment kernel (CH_BUILD_EXPERIMENTS=1 builds it)")
    base = _encoder(sd, cfg["heads"], max_batch=4).encode(x.to(dev))["codes"].cpu()
    fused = _encoder(sd, cfg["heads"], max_batch=4, options={"fused_adapter": 1}).encode(x.to(dev))["codes"].cpu()
    ref = eo.encode(sd, x, heads=cfg["heads"], with_pooled=False)["codes"]
    assert not torch.equal(fused, base)                       # really a different code path
    assert _rel_err(fused, ref) < 4e-2 and _rel_err(fused, base) < 2e-2


def test_concept_token_attention_maps(dev):
    """Optional interpretability output: last-layer attention of the Q concept tokens over the patch tokens, i.e. the slice
    attn_cache[-1][:, :, -Q:, 1:-Q] the reference's visualisation / attention-diversity code consumes."""
    from oracle import encoder_oracle as eo
    sd, z = load_fixture("encode_hd64")
    heads = int(z["meta/heads"])
    x = torch.from_numpy(z["in/images"])
    enc = _encoder(sd, heads, max_batch=4)
    out = enc.encode(x.to(dev), want=("codes", "concept_attn"))
    st = {}
    eo.encode(sd, x, heads=heads, stages=st)
    Q = 4
    ref = st["attn1"][:, :, -Q:, 1:-Q]                      # last of the 2 layers
    got = out["concept_attn"].cpu()
    assert got.shape == ref.shape == (4, heads, Q, 16)
    assert float((got - ref).abs().max()) < 2e-3            # probabilities; bf16 q/k operands
    assert torch.allclose(got.sum(-1), ref.sum(-1), atol=2e-3)
    base = enc.encode(x.to(dev))["codes"]
    assert torch.equal(base, out["codes"])                  # the tap does not perturb the encode


def test_two_stream_micro_batches_give_identical_codes(dev, monkeypatch):
    """CH_STREAMS=n splits the batch into n micro-batches on n HIP streams (default 2); rows are independent, so every output
    is bit-equal to the single-chain result."""
    from oracle import encoder_oracle as eo
    cfg = dict(eo.CONFIGS["vit_s16"])
    cfg["L"] = 2
    sd = eo.synthetic_state_dict(cfg, nbit=64, nclass=10)
    x = eo.synthetic_images(7, cfg["image"]).to(dev)
    monkeypatch.setenv("CH_STREAMS", "1")
    one = _encoder(sd, cfg["heads"], max_batch=8).encode(x, want=("codes", "packed", "concept_attn"))
    for ns in ("2", "3", "4"):       # 2 = the library default
        monkeypatch.setenv("CH_STREAMS", ns)
        encn = _encoder(sd, cfg["heads"], max_batch=8)
        for _ in range(3):
            many = encn.encode(x, want=("codes", "packed", "concept_attn"))
            torch.cuda.synchronize()
            for k in ("codes", "packed", "concept_attn"):
                assert torch.equal(one[k], many[k]), (ns, k)
        encn.close()


def test_layernorm_fold_chain_matches_the_unfolded_chain(dev, monkeypatch):
    """Default: LayerNorm folded into the consumer GEMMs (bf16 rounding BEFORE the normalisation, no LayerNorm launches in
    the layer loop).  CH_LN_FOLD=0: LayerNorm as separate launches (rounding after).  Same weights, same images: both
    within the encode tolerance of the oracle, and close to each other."""
    from oracle import encoder_oracle as eo
    cfg = dict(eo.CONFIGS["vit_b16"])
    cfg["L"] = 4
    sd = eo.synthetic_state_dict(cfg, nbit=64, nclass=10)
    x = eo.synthetic_images(5, cfg["image"])
    enc = _encoder(sd, cfg["heads"], max_batch=8)
    fold = enc.encode(x.to(dev))["codes"].cpu()
    hid_fold = enc.hidden_states(x.to(dev), cfg["L"]).cpu()
    monkeypatch.setenv("CH_LN_FOLD", "0")
    enc0 = _encoder(sd, cfg["heads"], max_batch=8)
    plain = enc0.encode(x.to(dev))["codes"].cpu()
    hid_plain = enc0.hidden_states(x.to(dev), cfg["L"]).cpu()
    monkeypatch.delenv("CH_LN_FOLD")
    st = {}
    ref = eo.encode(sd, x, heads=cfg["heads"], with_pooled=False, stages=st)["codes"]
    assert not torch.equal(fold, plain)                       # really a different code path
    assert _rel_err(fold, ref) < 4e-2 and _rel_err(plain, ref) < 4e-2 and _rel_err(fold, plain) < 2e-2
    assert _rel_err(hid_fold, hid_plain) < 4e-2               # max-abs / RMS of the final residual stream
    assert float((hid_fold - hid_plain).pow(2).mean().sqrt() / hid_plain.pow(2).mean().sqrt()) < 5e-3


def test_final_layer_row_pruning_is_bit_identical(dev, monkeypatch):
    """Default: after the final layer's attention only the rows the hashing head reads (CLS + Q concept tokens per image) are
    carried through out_proj / adapters / MLP.  Same kernels, same per-row arithmetic -> every output is bit-identical to the
    unpruned chain (CH_PRUNE_LAST=0); the hidden-state tap always runs unpruned."""
    from oracle import encoder_oracle as eo
    cfg = dict(eo.CONFIGS["vit_b16"])
    cfg["L"] = 3
    sd = eo.synthetic_state_dict(cfg, nbit=64, nclass=10)
    x = eo.synthetic_images(7, cfg["image"]).to(dev)
    want = ("codes", "packed", "logits_cont", "logits_bin", "logits_concept", "hash_features", "image_features", "concept_attn")
    enc = _encoder(sd, cfg["heads"], max_batch=4)        # 7 images at max_batch 4: chunks of 4 and 3
    pruned = enc.encode(x, want=want)
    hid = enc.hidden_states(x[:3], cfg["L"])
    assert enc.flops_per_image > 0
    monkeypatch.setenv("CH_PRUNE_LAST", "0")
    enc0 = _encoder(sd, cfg["heads"], max_batch=4)
    full = enc0.encode(x, want=want)
    hid0 = enc0.hidden_states(x[:3], cfg["L"])
    monkeypatch.delenv("CH_PRUNE_LAST")
    torch.cuda.synchronize()
    for k in want:
        assert torch.equal(pruned[k], full[k]), k
    assert torch.equal(hid, hid0)
    assert enc.flops_per_image < enc0.flops_per_image
    # two streams + pruning: each micro-batch owns its slice of the compact residual
    monkeypatch.setenv("CH_STREAMS", "2")
    enc2 = _encoder(sd, cfg["heads"], max_batch=8)
    two = enc2.encode(x, want=want)
    torch.cuda.synchronize()
    for k in want:
        assert torch.equal(two[k], full[k]), k


@pytest.mark.parametrize("batch", [1, 2, 9])
def test_small_and_odd_batches_through_the_default_chain(dev, batch):
    """Batch sizes whose row counts are far from any tile multiple (201 .. 1809 token rows; 5 .. 45 compact head rows in the
    pruned final layer) through LayerNorm fold + final-layer pruning, against the fp32 oracle and against a larger batch that
    contains the same images (rows are independent: bit-identical)."""
    from oracle import encoder_oracle as eo
    cfg = dict(eo.CONFIGS["vit_s16"])
    cfg["L"] = 3
    sd = eo.synthetic_state_dict(cfg, nbit=32, nclass=10)
    x = eo.synthetic_images(12, cfg["image"])
    enc = _encoder(sd, cfg["heads"], max_batch=16)
    want = ("codes", "packed", "logits_cont", "hash_features", "image_features")
    small = enc.encode(x[:batch].to(dev), want=want)
    big = enc.encode(x.to(dev), want=want)
    torch.cuda.synchronize()
    for k in want:
        assert torch.equal(small[k], big[k][:batch]), k
    ref = eo.encode(sd, x[:batch], heads=cfg["heads"], with_pooled=True)
    assert _rel_err(small["codes"].cpu(), ref["codes"]) < 4e-2
    assert _rel_err(small["image_features"].cpu(), ref["image_features"]) < 4e-2


def test_graph_replay_of_small_batches_is_bit_identical(dev):
    """Option "graph_max_batch" (opt-in: measured neutral, profiles/r04_graph_replay_ab.txt): a ch_encode call with B <= that value stages its
    images, replays the whole two-chain launch sequence as ONE captured hipGraph and copies the requested outputs out of staging buffers.
    First call of a (batch, dtype, output set) runs eagerly and captures; later calls replay.  Every output must equal the eager path's
    bit for bit -- fp32 and bf16 images, several output sets (incl. every layer's attention rows), batch sizes 1 .. 9, repeated replays
    with other inputs in between -- and an option change must drop the cached graphs."""
    from oracle import encoder_oracle as eo
    cfg = dict(eo.CONFIGS["vit_s16"])
    cfg["L"] = 3
    sd = eo.synthetic_state_dict(cfg, nbit=64, nclass=10)
    x = eo.synthetic_images(9, cfg["image"]).to(dev)
    eager = _encoder(sd, cfg["heads"], max_batch=16, options={"graph_max_batch": 0})
    graph = _encoder(sd, cfg["heads"], max_batch=16, options={"graph_max_batch": 64})
    assert graph.get_option("graph_max_batch") == 16 and eager.get_option("graph_max_batch") == 0    # clamped to max_batch
    wants = [("codes", "packed"), ("codes", "packed", "logits_cont", "logits_bin", "logits_concept", "hash_features", "image_features"),
             ("codes", "concept_attn"), ("codes", "concept_attn_layers")]
    for B in (1, 2, 5, 9):
        for xs in (x[:B], x[:B].to(torch.bfloat16)):
            for want in wants:
                ref = eager.encode(xs, want=want)
                for rep in range(3):                      # capture, replay, replay (another input staged in between)
                    if rep == 2:
                        graph.encode(torch.flip(xs, dims=[0]), want=want)
                    got = graph.encode(xs, want=want)
                    for k in ref:
                        assert torch.equal(got[k], ref[k]), (B, str(xs.dtype), want, k, rep)
    caps, reps = graph.get_option("graph_captures"), graph.get_option("graph_replays")
    assert caps == 4 * 2 * len(wants) and reps >= 2 * caps, (caps, reps)
    assert eager.get_option("graph_captures") == 0
    graph.set_option("streams", 1)                        # a captured chain has the old setting baked in: dropped and re-captured
    got = graph.encode(x[:5], want=wants[0])
    assert graph.get_option("graph_captures") == caps + 1
    assert torch.equal(got["codes"], eager.encode(x[:5], want=wants[0])["codes"])
    big = graph.encode(x, want=wants[0])                  # 9 <= 16: graph; a batch above the limit runs eagerly
    graph.set_option("graph_max_batch", 4)
    again = graph.encode(x, want=wants[0])
    assert torch.equal(big["codes"], again["codes"]) and graph.get_option("graph_captures") == caps + 2
    eager.close()
    graph.close()


def test_chain_auto_rule(dev, monkeypatch):
    """Option "chain_auto" (default on): ch_encode launches ONE chain for the batches where that measures faster -- fewer than 5,600 token
    rows (batch <= 27 of ViT-B/16: every launch is one tile's latency anyway) -- and two everywhere else.  Same bits either way."""
    from oracle import encoder_oracle as eo
    monkeypatch.delenv("CH_CHAIN_AUTO", raising=False)
    cfg = dict(eo.CONFIGS["vit_b16"])
    cfg["L"] = 2
    sd = eo.synthetic_state_dict(cfg, nbit=64, nclass=10)
    auto = _encoder(sd, cfg["heads"], max_batch=112)
    split = _encoder(sd, cfg["heads"], max_batch=112, options={"chain_auto": 0})
    assert auto.get_option("chain_auto") == 1 and auto.get_option("streams") == 2
    x = eo.synthetic_images(112, cfg["image"]).to(dev).to(torch.bfloat16)
    for batch, chains in ((8, 1), (24, 1), (32, 2), (96, 2), (112, 2)):
        a = auto.encode(x[:batch], want=("codes", "packed"))
        assert auto.get_option("last_chains") == chains, (batch, auto.get_option("last_chains"))
        b = split.encode(x[:batch], want=("codes", "packed"))
        assert split.get_option("last_chains") == 2
        torch.cuda.synchronize()
        assert torch.equal(a["codes"], b["codes"]) and torch.equal(a["packed"], b["packed"])
    auto.close()
    split.close()


def test_small_grids_run_the_ring_kernel_with_the_same_bits(dev):
    """Up to 8,192 token rows (batch <= 40 of ViT-B/16) the GEMMs of the 128x128 path run the four-stage ring kernel (gemm_r4.hip; option
    "small_kernel" 0 = that rule, 1 = the two-phase kernel always, 2 = the ring always).  An image's codes do not depend on the kernel or on
    the batch it came in: 8 images alone (ring), the same 8 inside a batch of 64 (two-phase / 256x256 kernels), and either kernel forced."""
    from oracle import encoder_oracle as eo
    cfg = dict(eo.CONFIGS["vit_b16"])
    cfg["L"] = 3
    sd = eo.synthetic_state_dict(cfg, nbit=64, nclass=10)
    x = eo.synthetic_images(64, cfg["image"]).to(dev).to(torch.bfloat16)
    want = ("codes", "packed", "logits_cont", "hash_features")
    encs = {k: _encoder(sd, cfg["heads"], max_batch=64, options={"small_kernel": k}) for k in (0, 1, 2)}
    big = encs[0].encode(x, want=want)
    torch.cuda.synchronize()
    big = {k: v.clone() for k, v in big.items()}
    for n in (1, 8, 40):
        outs = {k: {a: b.clone() for a, b in e.encode(x[:n], want=want).items()} for k, e in encs.items()}
        torch.cuda.synchronize()
        for key in want:
            assert torch.equal(outs[0][key], outs[1][key]) and torch.equal(outs[0][key], outs[2][key]), (n, key)
            assert torch.equal(outs[0][key], big[key][:n]), (n, key)
    for e in encs.values():
        e.close()
