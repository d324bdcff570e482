"""GPU parity for the packed Hamming path (csrc/hamming.hip through the C-ABI): bit-exact against the CPU oracle
(oracle/hamming_oracle.c) on seeded inputs, against the committed label fixtures at the real dataset sizes, and through
size-independent properties at BASELINE.json's full synthetic size (1M x 128 bit)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _t(a, dev):
    if a.dtype == np.uint64:
        a = a.view(np.int64)
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("nbit", [16, 64, 70, 128, 256])
def test_pack_sign_bit_exact(dev, nbit):
    from concepthash_amd import retrieval as rt
    from oracle import hamming_oracle as ho
    rng = np.random.default_rng(nbit)
    codes = rng.standard_normal((1000, nbit)).astype(np.float32)
    codes[3] = 0.0
    codes[4, ::2] = -0.0
    for thr in (0.0, 0.3):
        got = rt.pack_sign(_t(codes, dev), thr).cpu().numpy().view(np.uint64)
        assert np.array_equal(got, ho.pack(codes, thr))
    assert rt.pack_sign(torch.zeros(0, nbit, device=dev)).shape == (0, (nbit + 63) // 64)


@pytest.mark.parametrize("nbit,Qn,G,k", [(64, 300, 1000, 10), (64, 257, 4099, 1), (128, 100, 3000, 16), (128, 64, 2000, 17),
                                          (192, 50, 700, 33), (256, 33, 900, 100), (64, 5, 3, 10), (64, 1, 1, 1),
                                          (64, 700, 70000, 10)])
def test_topk_bit_exact(dev, nbit, Qn, G, k):
    from concepthash_amd import retrieval as rt
    from oracle import hamming_oracle as ho
    q, _ = ho.synthetic_codes(Qn, nbit, seed=11)
    g, _ = ho.synthetic_codes(G, nbit, seed=12)
    idx, dst = rt.hamming_topk(_t(q, dev), _t(g, dev), k)
    ridx, rdst = ho.topk(q, g, k)
    assert np.array_equal(idx.cpu().numpy(), ridx.astype(np.int64))
    assert np.array_equal(dst.cpu().numpy(), rdst)
    if Qn * G <= 4_000_000:
        assert np.array_equal(rt.hamming_dist(_t(q, dev), _t(g, dev)).cpu().numpy(), ho.dist(q, g))


def test_topk_adversarial_ties_and_offsets(dev):
    from concepthash_amd import retrieval as rt
    from oracle import hamming_oracle as ho
    # 3 distinct codes repeated: almost everything ties; order must be by gallery index
    q, _ = ho.synthetic_codes(130, 64, seed=1, nclass=3, flip=0.0)
    g, _ = ho.synthetic_codes(5000, 64, seed=2, nclass=3, flip=0.0)
    idx, dst = rt.hamming_topk(_t(q, dev), _t(g, dev), 40)
    ridx, rdst = ho.topk(q, g, 40)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(dst.cpu().numpy(), rdst)
    # all-zero codes
    z = np.zeros((600, 1), dtype=np.uint64)
    idx, dst = rt.hamming_topk(_t(z[:10], dev), _t(z, dev), 12)
    assert (idx.cpu().numpy() == np.arange(12)[None, :]).all() and (dst.cpu().numpy() == 0).all()
    # gallery shards + merge == one gallery (the multi-GPU data path, on one device)
    q, _ = ho.synthetic_codes(77, 128, seed=3)
    g, _ = ho.synthetic_codes(9001, 128, seed=4)
    bounds = [0, 2000, 2001, 6000, 9001]
    parts = [rt.hamming_topk(_t(q, dev), _t(g[a:b], dev), 10, g_index_base=a) for a, b in zip(bounds[:-1], bounds[1:])]
    midx, mdst = rt.topk_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    ridx, rdst = ho.topk(q, g, 10)
    assert np.array_equal(midx.cpu().numpy(), ridx) and np.array_equal(mdst.cpu().numpy(), rdst)
    # empty gallery / no queries
    e_idx, e_dst = rt.hamming_topk(_t(q[:4], dev), _t(g[:0], dev), 3)
    assert (e_idx.cpu().numpy() == -1).all() and (e_dst.cpu().numpy() == -1).all()
    n_idx, _ = rt.hamming_topk(_t(q[:0], dev), _t(g, dev), 3)
    assert n_idx.shape == (0, 3)


@pytest.mark.parametrize("nbit,R,remove_first,seg", [(64, -1, False, None), (64, 50, False, 300), (64, -1, True, 257),
                                                     (128, -1, False, None), (128, 1000, True, 1000), (192, -1, False, 400)])
def test_map_statistics_bit_exact(dev, nbit, R, remove_first, seg):
    from concepthash_amd import retrieval as rt
    from oracle import hamming_oracle as ho
    q, ql = ho.synthetic_codes(333, nbit, seed=21, nclass=20)
    g, gl = ho.synthetic_codes(4567, nbit, seed=22, nclass=20)
    got = rt.evaluate(_t(q, dev), _t(g, dev), _t(ql, dev), _t(gl, dev), R=R, ks=(1, 5, 10), remove_first=remove_first,
                      seg_rows=seg)
    ref = ho.mean_ap(q, g, ql, gl, R=R, ks=(1, 5, 10), remove_first=remove_first)
    assert np.array_equal(got["S"].cpu().numpy().view(np.uint64), ref["S"])
    assert np.array_equal(got["nrel"].cpu().numpy().astype(np.uint32), ref["nrel"])
    assert np.array_equal(got["hits"].cpu().numpy().astype(np.uint32), ref["hits"])
    assert np.array_equal(got["total"].cpu().numpy().astype(np.uint32), ref["total"])
    assert abs(got["mAP"] - ref["mAP"]) < 1e-12
    assert abs(got["mAP"] - ref["mAP_f64"]) < 1e-9            # fixed point vs reference-style float64 mean
    assert np.allclose(got["precisions"], ref["precisions"], atol=1e-12)
    assert np.allclose(got["recalls"], ref["recalls"], atol=1e-12)


def test_map_histogram_and_multilabel(dev):
    from concepthash_amd import retrieval as rt
    from oracle import hamming_oracle as ho
    q, ql = ho.synthetic_codes(100, 64, seed=31, nclass=9)
    g, gl = ho.synthetic_codes(2500, 64, seed=32, nclass=9)
    qlab, glab, LW = rt.prepare_labels(_t(ql, dev), _t(gl, dev))
    hist = rt.hamming_hist(_t(q, dev), _t(g, dev), qlab, glab, LW, 700)
    ref = ho.mean_ap(q, g, ql, gl, want_hist=True)
    assert np.array_equal(hist.sum(0).cpu().numpy().astype(np.uint32), ref["hist"])
    rng = np.random.default_rng(5)
    qm = (rng.random((100, 130)) < 0.03).astype(np.int64)      # 130 classes -> 3 mask words, some rows empty
    gm = (rng.random((2500, 130)) < 0.03).astype(np.int64)
    got = rt.evaluate(_t(q, dev), _t(g, dev), _t(qm, dev), _t(gm, dev), ks=(1, 5))
    refm = ho.mean_ap(q, g, qm, gm, ks=(1, 5))
    assert np.array_equal(got["S"].cpu().numpy().view(np.uint64), refm["S"])
    assert np.array_equal(got["hits"].cpu().numpy().astype(np.uint32), refm["hits"])
    # one-hot indicator matrices take the single-label path and agree with class ids
    oh = rt.evaluate(_t(q, dev), _t(g, dev), _t(np.eye(9, dtype=np.int64)[ql], dev), _t(np.eye(9, dtype=np.int64)[gl], dev))
    assert np.array_equal(oh["S"].cpu().numpy().view(np.uint64), ref["S"])


@pytest.mark.parametrize("name,nbit", [("cub200", 64), ("cars196", 64)])
def test_dataset_sized_map_with_real_label_vectors(dev, name, nbit):
    """Real label vectors (tests/golden/labels_*.npz, parsed from the reference's list files), clustered synthetic codes."""
    from concepthash_amd import retrieval as rt
    from oracle import hamming_oracle as ho
    z = np.load(os.path.join(GOLDEN, f"labels_{name}.npz"))
    gl, ql = z["db"].astype(np.int32), z["test"].astype(np.int32)
    C = int(max(gl.max(), ql.max())) + 1
    centres = np.random.default_rng(7).integers(0, 2, size=(C, nbit), dtype=np.uint8)

    def codes(labels, seed):
        rng = np.random.default_rng(seed)
        bits = centres[labels] ^ (rng.random((len(labels), nbit)) < 0.2).astype(np.uint8)
        return np.ascontiguousarray(np.packbits(bits, axis=1, bitorder="little")).view("<u8")

    q, g = codes(ql, 1), codes(gl, 2)
    got = rt.evaluate(_t(q, dev), _t(g, dev), _t(ql, dev), _t(gl, dev), R=-1, ks=(1, 5, 10))
    ref = ho.mean_ap(q, g, ql, gl, R=-1, ks=(1, 5, 10))
    assert np.array_equal(got["S"].cpu().numpy().view(np.uint64), ref["S"])
    assert np.array_equal(got["nrel"].cpu().numpy().astype(np.uint32), ref["nrel"])
    assert abs(got["mAP"] - ref["mAP"]) < 1e-12 and abs(got["mAP"] - ref["mAP_f64"]) < 1e-3
    print(f"{name}: mAP@all {got['mAP']:.6f} (oracle {ref['mAP']:.6f}), P@1/5/10 {got['precisions']}")


def test_full_size_properties_1m_gallery(dev):
    """BASELINE.json config 5 size: 1M x 128-bit gallery.  The oracle cannot rank this in seconds for many queries, so:
    (1) returned lists are sorted by (dist, idx), unique, in range;  (2) reported distances equal recomputed distances;
    (3) for a sample of queries the list equals a stable sort of the exact distance row;  (4) sharding + merge is
    idempotent (same answer as the unsharded call)."""
    from concepthash_amd import retrieval as rt
    from oracle import hamming_oracle as ho
    G, Qn, k = 1_000_000, 1024, 10
    g, _ = ho.synthetic_codes(G, 128, seed=41)
    q, _ = ho.synthetic_codes(Qn, 128, seed=42)
    gq, gg = _t(q, dev), _t(g, dev)
    idx, dst = rt.hamming_topk(gq, gg, k)
    torch.cuda.synchronize()
    assert int(idx.min()) >= 0 and int(idx.max()) < G
    comp = dst.long() * (1 << 32) + idx
    assert bool((comp[:, 1:] > comp[:, :-1]).all())                      # strictly ascending (dist, idx)
    gw = g.view(np.int64)
    rows = torch.from_numpy(gw).to(dev)[idx.reshape(-1)].reshape(Qn, k, 2)
    x = rows ^ gq[:, None, :]
    pop = torch.zeros(Qn, k, dtype=torch.int32, device=dev)
    for b in range(64):
        pop += ((x >> b) & 1).sum(-1).int()
    assert torch.equal(pop, dst)                                         # reported distance == recomputed
    sample = [0, 1, 511, 1023]
    drow = rt.hamming_dist(gq[sample], gg)                               # exact rows [4, 1M]
    order = torch.sort(drow.long() * (1 << 32) + torch.arange(G, device=dev)[None, :], dim=1).indices[:, :k]
    assert torch.equal(order, idx[sample])
    ridx, rdst = ho.topk(q[:2], g, k)                                     # two queries through the CPU oracle
    assert np.array_equal(idx[:2].cpu().numpy(), ridx) and np.array_equal(dst[:2].cpu().numpy(), rdst)
    half = G // 2
    p0 = rt.hamming_topk(gq, gg[:half], k, 0)
    p1 = rt.hamming_topk(gq, gg[half:], k, half)
    midx, mdst = rt.topk_merge(torch.stack([p0[0], p1[0]]), torch.stack([p0[1], p1[1]]))
    assert torch.equal(midx, idx) and torch.equal(mdst, dst)


@pytest.mark.parametrize("nbit,k", [(64, 10), (128, 33), (64, 1)])
def test_topk_large_gallery_with_heavy_ties(dev, nbit, k):
    """300,000 rows of clustered codes (few centres, low noise): thousands of rows sit at the k-th distance, exact duplicates of
    a query lie at the very end of the gallery; the lists must equal the CPU oracle's stable (distance, index) ranking."""
    from concepthash_amd import retrieval as rt
    from oracle import hamming_oracle as ho
    G, Qn = 300_000, 48
    g, _ = ho.synthetic_codes(G, nbit, seed=51, nclass=20, flip=0.02)
    q, _ = ho.synthetic_codes(Qn, nbit, seed=52, nclass=20, flip=0.02)
    g = g.copy()
    g[G - 5] = q[0]                      # exact duplicates of query 0 at the very end: distance 0, beyond the sample
    g[G - 1] = q[0]
    g[40_000] = q[1]                     # ... and just outside the sampled prefix
    idx, dst = rt.hamming_topk(_t(q, dev), _t(g, dev), k)
    torch.cuda.synchronize()
    ridx, rdst = ho.topk(q, g, k)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(dst.cpu().numpy(), rdst)


# ---- round 2: NABirds size, reference-pinned distance / arg-min, one-pass multi-limit evaluation ----------------------------
def _clustered(labels, centres, nbit, seed, flip=0.2):
    rng = np.random.default_rng(seed)
    bits = centres[labels] ^ (rng.random((len(labels), nbit)) < flip).astype(np.uint8)
    return np.ascontiguousarray(np.packbits(bits, axis=1, bitorder="little")).view("<u8")


def test_nabirds_sized_map_single_call_and_eight_way_shards(dev):
    """BASELINE.json config 4: NABirds, 64-bit, 24,633 queries x 23,929 gallery rows, C = 555 (real label vectors,
    tests/golden/labels_nabirds.npz).  mAP@all + P@k / R@k bit-exact vs the C oracle in a single call, and again with the
    gallery split 8 ways (ragged shards) through per-shard histograms -> one global hist_prefix -> per-shard AP pass ->
    integer sum, and per-shard top-k -> topk_merge -- the single-GPU data path of the 8-GPU layout."""
    from concepthash_amd import retrieval as rt
    from concepthash_amd.distributed import shard_bounds
    from oracle import hamming_oracle as ho
    z = np.load(os.path.join(GOLDEN, "labels_nabirds.npz"))
    gl, ql = z["db"].astype(np.int32), z["test"].astype(np.int32)
    assert (len(ql), len(gl)) == (24633, 23929)
    C = int(max(gl.max(), ql.max())) + 1
    assert C >= 555
    centres = np.random.default_rng(17).integers(0, 2, size=(C, 64), dtype=np.uint8)
    q, g = _clustered(ql, centres, 64, 1), _clustered(gl, centres, 64, 2)
    ks = (1, 5, 10)
    ref = ho.mean_ap(q, g, ql, gl, R=-1, ks=ks)
    gq, gg, gql, ggl = _t(q, dev), _t(g, dev), _t(ql, dev), _t(gl, dev)
    got = rt.evaluate(gq, gg, gql, ggl, R=-1, ks=ks)
    assert np.array_equal(got["S"].cpu().numpy().view(np.uint64), ref["S"])
    assert np.array_equal(got["nrel"].cpu().numpy().astype(np.uint32), ref["nrel"])
    assert np.array_equal(got["hits"].cpu().numpy().astype(np.uint32), ref["hits"])
    assert np.array_equal(got["total"].cpu().numpy().astype(np.uint32), ref["total"])
    assert abs(got["mAP"] - ref["mAP"]) < 1e-12
    print(f"nabirds: mAP@all {got['mAP']:.6f}, P@1/5/10 {got['precisions']}")
    # ---- 8 ragged shards on one GPU
    b = shard_bounds(len(gl), 8)
    qlab, glab, LW = rt.prepare_labels(gql, ggl)
    seg = rt.map_seg_rows(len(ql), max(b[i + 1] - b[i] for i in range(8)), 1)
    nseg = max(1, -(-max(b[i + 1] - b[i] for i in range(8)) // seg))
    hists = []
    for r in range(8):
        h = rt.hamming_hist(gq, gg[b[r]:b[r + 1]], qlab, glab[b[r]:b[r + 1]], LW, seg)
        if h.shape[0] < nseg:
            h = torch.cat([h, torch.zeros((nseg - h.shape[0],) + tuple(h.shape[1:]), dtype=h.dtype, device=dev)])
        hists.append(h)
    base_all, totals = rt.hist_prefix(torch.cat(hists))
    limits, idx_of = rt.normalize_limits([-1] + list(ks))
    S = torch.zeros(len(limits), len(ql), dtype=torch.int64, device=dev)
    nrel = torch.zeros(len(limits), len(ql), dtype=torch.int32, device=dev)
    for r in range(8):
        s_r, n_r = rt.hamming_ap_multi(gq, gg[b[r]:b[r + 1]], qlab, glab[b[r]:b[r + 1]], LW, seg,
                                       base_all[r * nseg:(r + 1) * nseg].contiguous(), limits)
        S += s_r
        nrel += n_r
    assert np.array_equal(S[idx_of[0]].cpu().numpy().view(np.uint64), ref["S"])
    assert np.array_equal(nrel[idx_of[0]].cpu().numpy().astype(np.uint32), ref["nrel"])
    for t in range(len(ks)):
        assert np.array_equal(nrel[idx_of[1 + t]].cpu().numpy().astype(np.uint32), ref["hits"][:, t])
    assert np.array_equal(totals[:, 1].cpu().numpy().astype(np.uint32), ref["total"])
    # the same shards in the one-scan form (what ShardedRetrieval.evaluate runs): per-shard histogram + records, the GLOBAL
    # prefix, per-shard AP terms from the shard's own records
    hists, recs = [], []
    for r in range(8):
        h, rc = rt.hamming_hist_rec(gq, gg[b[r]:b[r + 1]], qlab, glab[b[r]:b[r + 1]], LW, seg)
        if h.shape[0] < nseg:
            h = torch.cat([h, torch.zeros((nseg - h.shape[0],) + tuple(h.shape[1:]), dtype=h.dtype, device=dev)])
        hists.append(h)
        recs.append(rc)
    base_rec, _ = rt.hist_prefix(torch.cat(hists))
    assert torch.equal(base_rec, base_all)
    S2 = torch.zeros_like(S)
    nrel2 = torch.zeros_like(nrel)
    for r in range(8):
        s_r, n_r = rt.hamming_ap_rec(gq, gg[b[r]:b[r + 1]], qlab, glab[b[r]:b[r + 1]], LW, seg,
                                     base_rec[r * nseg:(r + 1) * nseg].contiguous(), recs[r], limits)
        S2 += s_r
        nrel2 += n_r
    assert torch.equal(S2, S) and torch.equal(nrel2, nrel)
    parts = [rt.hamming_topk(gq, gg[b[r]:b[r + 1]], 10, g_index_base=b[r]) for r in range(8)]
    midx, mdst = rt.topk_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    idx, dst = rt.hamming_topk(gq, gg, 10)
    assert torch.equal(midx, idx) and torch.equal(mdst, dst)
    ridx, rdst = ho.topk(q[:64], g, 10)
    assert np.array_equal(idx[:64].cpu().numpy(), ridx.astype(np.int64)) and np.array_equal(dst[:64].cpu().numpy(), rdst)


@pytest.mark.parametrize("nbit", [64, 128])
def test_distance_argmin_top5_equal_the_reference_fixture(dev, nbit):
    """tests/golden/get_hd.npz holds OUTPUTS OF THE REFERENCE: get_hd (trainers/orthohash.py:263-264) pair by pair, and the
    arg-min / 5-smallest that calculate_accuracy_hamm_dist (utils/metrics.py:18-29) ranks by, on +-1 codes with engineered ties
    (oracle/gen_retrieval_golden.py).  ch_hamming_dist / utils.hashing.get_hamm_dist reproduce the matrix exactly; arg-min and
    the 5 smallest come out of ch_hamming_topk."""
    from concepthash_amd import retrieval as rt
    from utils import hashing
    z = np.load(os.path.join(GOLDEN, "get_hd.npz"))
    tag = f"b{nbit}/"
    codes, cb = z[tag + "codes"].astype(np.float32), z[tag + "codebook"].astype(np.float32)
    ref = z[tag + "get_hd"]
    got = hashing.get_hamm_dist(torch.from_numpy(codes), torch.from_numpy(cb), normalize=True).cpu().numpy()
    assert got.dtype == np.float32 and np.array_equal(got, ref)                     # k / nbit is exact in fp32
    a, b = rt.pack_sign(torch.from_numpy(codes).to(dev)), rt.pack_sign(torch.from_numpy(cb).to(dev))
    d = rt.hamming_dist(a, b).cpu().numpy()
    assert np.array_equal(d, np.rint(ref * nbit).astype(np.int32))
    idx, dst = rt.hamming_topk(a, b, 5)
    idx, dst = idx.cpu().numpy(), dst.cpu().numpy()
    assert np.array_equal(idx[:, 0], z[tag + "argmin"])                             # ties -> lowest index, as torch.argmin
    assert ref[0, 0] == ref[0, 1] and idx[0, 0] == 0                                # the engineered tie is really there
    ref5 = z[tag + "top5_smallest"]
    assert np.array_equal(dst, np.rint(np.take_along_axis(ref, ref5, 1) * nbit).astype(np.int32))   # same 5 distances
    for i in range(len(idx)):      # same 5 classes wherever no tie straddles the 5th place (torch.topk leaves tie order open)
        kth = np.sort(d[i])[4]
        if (d[i] == kth).sum() == (dst[i] == kth).sum():
            assert set(idx[i]) == set(ref5[i]), i
    labels = z[tag + "labels"]
    assert np.float32((idx[:, 0] == labels).mean()) == z[tag + "acc_argmin"] == z[tag + "acc_argmin_onehot"]
    assert np.float32((idx == labels[:, None]).any(1).mean()) == z[tag + "acc_top5"]


def test_one_pass_multi_limit_evaluation(dev):
    """mAP@R for a list of R, P@k / R@k for any k (also k > 128, the top-k kernel's list limit) and the P/R curve come out of
    ONE histogram pass + ONE AP pass (16 rank limits per gallery scan).  Every entry equals the oracle run per limit."""
    from concepthash_amd import retrieval as rt
    from oracle import hamming_oracle as ho
    from utils import hashing
    q, ql = ho.synthetic_codes(333, 64, seed=61, nclass=7)
    g, gl = ho.synthetic_codes(4000, 64, seed=62, nclass=7)
    Rs = [1, 10, 100, 1000, -1]
    ks = (1, 5, 10, 200, 1500, 5000)                         # 5000 > G
    for remove_first in (False, True):
        got = rt.evaluate(_t(q, dev), _t(g, dev), _t(ql, dev), _t(gl, dev), R=Rs, ks=ks, remove_first=remove_first)
        for i, R in enumerate(Rs):
            ref = ho.mean_ap(q, g, ql, gl, R=R, ks=ks, remove_first=remove_first)
            assert np.array_equal(got["S"][i].cpu().numpy().view(np.uint64), ref["S"]), (R, remove_first)
            assert np.array_equal(got["nrel"][i].cpu().numpy().astype(np.uint32), ref["nrel"])
            assert abs(got["mAP"][i] - ref["mAP"]) < 1e-12
        assert np.array_equal(got["hits"].cpu().numpy().astype(np.uint32), ref["hits"])
        assert np.allclose(got["precisions"], ref["precisions"], atol=1e-12) and np.allclose(got["recalls"], ref["recalls"], atol=1e-12)
    # 20 limits -> two AP passes of <= 16; the public wrappers
    codes_q = np.where(np.unpackbits(q.view(np.uint8), axis=1, bitorder="little") > 0, 1.0, -1.0).astype(np.float32)
    codes_g = np.where(np.unpackbits(g.view(np.uint8), axis=1, bitorder="little") > 0, 1.0, -1.0).astype(np.float32)
    many = list(range(1, 4000, 211)) + [-1]
    assert len(many) > 16
    m, rec, prec = hashing.calculate_mAP(torch.from_numpy(codes_g), torch.from_numpy(gl), torch.from_numpy(codes_q),
                                         torch.from_numpy(ql), many, PRs=[1, 300])
    for R, v in zip(many, m):
        assert abs(v - ho.mean_ap(q, g, ql, gl, R=R)["mAP"]) < 1e-12
    refk = ho.mean_ap(q, g, ql, gl, R=-1, ks=(1, 300))
    assert np.allclose(prec, refk["precisions"], atol=1e-12) and np.allclose(rec, refk["recalls"], atol=1e-12)
    rc, pc, depths = hashing.calculate_pr_curve(torch.from_numpy(codes_g), torch.from_numpy(gl), torch.from_numpy(codes_q),
                                                torch.from_numpy(ql))
    assert depths[0] == 1 and depths[-1] == 4000
    for R, r_, p_ in zip(depths, rc, pc):
        refR = ho.mean_ap(q, g, ql, gl, R=R)
        nr, tot = refR["nrel"].astype(np.float64), refR["total"].astype(np.float64)
        assert abs(p_ - float((nr / R).mean())) < 1e-12
        assert abs(r_ - float(np.where(tot > 0, nr / np.maximum(tot, 1), 0).mean())) < 1e-12


def test_map_at_config5_size_1m_gallery(dev):
    """BASELINE.json config 5 size for the mAP path: 1M x 128-bit gallery (16 segments of 62,500 rows, 132 KB of LDS counters per
    workgroup).  768 queries: S / nrel / hits / total bit-exact against the C oracle for a sample of 48 of them (the oracle ranks
    1M rows per query), and for ALL of them the 8-way row-sharded evaluation (per-shard histograms -> one prefix -> per-shard AP
    passes -> integer sum) equals the single call -- the size-independent property."""
    from concepthash_amd import retrieval as rt
    from concepthash_amd.distributed import shard_bounds
    from oracle import hamming_oracle as ho
    G, Qn, ncls = 1_000_000, 768, 50
    g, gl = ho.synthetic_codes(G, 128, seed=71, nclass=ncls, flip=0.3)
    q, ql = ho.synthetic_codes(Qn, 128, seed=72, nclass=ncls, flip=0.3)
    gq, gg, gql, ggl = _t(q, dev), _t(g, dev), _t(ql, dev), _t(gl, dev)
    ks = (1, 10, 1000)
    got = rt.evaluate(gq, gg, gql, ggl, R=[1000, -1], ks=ks)
    torch.cuda.synchronize()
    sample = np.arange(0, Qn, 16)
    for i, R in enumerate((1000, -1)):
        ref = ho.mean_ap(q[sample], g, ql[sample], gl, R=R, ks=ks)
        assert np.array_equal(got["S"][i][sample].cpu().numpy().view(np.uint64), ref["S"]), R
        assert np.array_equal(got["nrel"][i][sample].cpu().numpy().astype(np.uint32), ref["nrel"]), R
    assert np.array_equal(got["hits"][sample].cpu().numpy().astype(np.uint32), ref["hits"])
    assert np.array_equal(got["total"][sample].cpu().numpy().astype(np.uint32), ref["total"])
    # 8 row shards on one GPU == single call, for every query
    b = shard_bounds(G, 8)
    qlab, glab, LW = rt.prepare_labels(gql, ggl)
    seg = rt.map_seg_rows(Qn, b[1] - b[0], 2)
    nseg = -(-(b[1] - b[0]) // seg)
    hists = [rt.hamming_hist(gq, gg[b[r]:b[r + 1]], qlab, glab[b[r]:b[r + 1]], LW, seg) for r in range(8)]
    assert all(h.shape[0] == nseg for h in hists)
    base_all, totals = rt.hist_prefix(torch.cat(hists))
    limits, idx_of = rt.normalize_limits([1000, -1] + list(ks))
    S = torch.zeros(len(limits), Qn, dtype=torch.int64, device=dev)
    nrel = torch.zeros(len(limits), Qn, dtype=torch.int32, device=dev)
    for r in range(8):
        s_r, n_r = rt.hamming_ap_multi(gq, gg[b[r]:b[r + 1]], qlab, glab[b[r]:b[r + 1]], LW, seg,
                                       base_all[r * nseg:(r + 1) * nseg].contiguous(), limits)
        S += s_r
        nrel += n_r
    for i in range(2):
        assert torch.equal(S[idx_of[i]], got["S"][i]) and torch.equal(nrel[idx_of[i]], got["nrel"][i])
    for t in range(len(ks)):
        assert torch.equal(nrel[idx_of[2 + t]], got["hits"][:, t])
    assert torch.equal(totals[:, 1], got["total"])


def test_randomised_shapes_against_oracle(dev):
    """40 seeded random problems -- query / gallery sizes off every tile and trip boundary (1 .. 1500 x 1 .. 9000), 64 .. 256 bit,
    k, explicit segment sizes (whole blocks, pairs of blocks and tails of the row loop), single-label and multi-hot relevance,
    remove_first, several rank limits -- through top-k and the one-pass evaluation, bit for bit against the oracle."""
    from concepthash_amd import retrieval as rt
    from oracle import hamming_oracle as ho
    rng = np.random.default_rng(2026)
    for case in range(40):
        nbit = int(rng.choice([64, 128, 192, 256]))
        Qn, G = int(rng.integers(1, 1500)), int(rng.integers(1, 9000))
        ncls = int(rng.integers(2, 40))
        q, ql = ho.synthetic_codes(Qn, nbit, seed=1000 + case, nclass=ncls, flip=0.15)
        g, gl = ho.synthetic_codes(G, nbit, seed=2000 + case, nclass=ncls, flip=0.15)
        k = int(rng.integers(1, 40))
        idx, dst = rt.hamming_topk(_t(q, dev), _t(g, dev), k)
        ridx, rdst = ho.topk(q, g, k)
        assert np.array_equal(idx.cpu().numpy(), ridx.astype(np.int64)) and np.array_equal(dst.cpu().numpy(), rdst), case
        multi = case % 4 == 3
        if multi:
            qlab = (rng.random((Qn, 70)) < 0.05).astype(np.int64)
            glab = (rng.random((G, 70)) < 0.05).astype(np.int64)
        else:
            qlab, glab = ql, gl
        remove_first = bool(case % 2) and G > 1
        R = int(rng.choice([-1, 1, 7, 100, 5000]))
        ks = tuple(sorted({int(v) for v in rng.integers(1, 300, size=3)}))
        seg = int(rng.choice([256, 257, 263, 1000, 4096])) if case % 3 else None
        got = rt.evaluate(_t(q, dev), _t(g, dev), _t(qlab, dev), _t(glab, dev), R=R, ks=ks, remove_first=remove_first, seg_rows=seg)
        ref = ho.mean_ap(q, g, qlab, glab, R=R, ks=ks, remove_first=remove_first)
        assert np.array_equal(got["S"].cpu().numpy().view(np.uint64), ref["S"]), (case, nbit, Qn, G, R, ks, remove_first, seg, multi)
        assert np.array_equal(got["nrel"].cpu().numpy().astype(np.uint32), ref["nrel"]), case
        assert np.array_equal(got["hits"].cpu().numpy().astype(np.uint32), ref["hits"]), case
        assert np.array_equal(got["total"].cpu().numpy().astype(np.uint32), ref["total"]), case


def test_scalar_load_form_of_the_map_passes_is_bit_identical(dev):
    """`ch_debug_set_hamming_scalar_loads(1)` (a test tap, include/concepthash_hip_debug.h) selects the scalar-load form of both mAP
    passes (the default takes the gallery through VMEM in 16-row blocks + DPP row broadcast, csrc/hamming.hip).  S / nrel / hits /
    total of four problems (64 .. 256 bit, explicit and default segment sizes) must be the same bytes under both forms, and one of
    them is checked against the oracle."""
    import hashlib
    from concepthash_amd import _lib, retrieval as rt
    from oracle import hamming_oracle as ho
    lib = _lib.load()

    def run():
        out = {}
        for nbit, Qn, G, ncls, seg in ((64, 700, 9001, 25, None), (128, 513, 7777, 12, 1000), (192, 300, 5000, 9, 257), (256, 129, 4100, 5, None)):
            q, ql = ho.synthetic_codes(Qn, nbit, seed=31, nclass=ncls, flip=0.15)
            g, gl = ho.synthetic_codes(G, nbit, seed=32, nclass=ncls, flip=0.15)
            got = rt.evaluate(_t(q, dev), _t(g, dev), _t(ql, dev), _t(gl, dev), R=-1, ks=(1, 5, 10), remove_first=bool(nbit == 128), seg_rows=seg)
            out[str(nbit)] = [hashlib.sha256(got[k].cpu().numpy().tobytes()).hexdigest() for k in ("S", "nrel", "hits", "total")]
        return out

    lib.ch_debug_set_hamming_scalar_loads(1)
    try:
        scalar = run()
    finally:
        lib.ch_debug_set_hamming_scalar_loads(0)
    default = run()
    assert scalar == default
    q, ql = ho.synthetic_codes(513, 128, seed=31, nclass=12, flip=0.15)
    g, gl = ho.synthetic_codes(7777, 128, seed=32, nclass=12, flip=0.15)
    ref = ho.mean_ap(q, g, ql, gl, R=-1, ks=(1, 5, 10), remove_first=True)
    assert hashlib.sha256(np.ascontiguousarray(ref["S"]).tobytes()).hexdigest() == default["128"][0]


@pytest.mark.parametrize("nbit", [64, 128, 192])
def test_record_form_equals_two_scan_form_and_survives_overflow(dev, nbit):
    """The one-scan evaluation (histogram pass that records the relevant rows + AP terms from the records, csrc/hamming.hip MODE 2)
    against the two-scan form and the oracle: default capacity, a capacity of 1 and 3 records per list (nearly every workgroup
    overflows and is redone by the two-scan kernel), a gallery SORTED by class (whole segments relevant to a query), multi-hot
    labels, remove_first and several rank limits."""
    from concepthash_amd import retrieval as rt
    from oracle import hamming_oracle as ho
    rng = np.random.default_rng(nbit)
    q, ql = ho.synthetic_codes(700, nbit, seed=51, nclass=7, flip=0.2)
    g, gl = ho.synthetic_codes(6000, nbit, seed=52, nclass=7, flip=0.2)
    order = np.argsort(gl, kind="stable")
    cases = [("shuffled", g, gl), ("sorted", g[order], gl[order])]
    for name, gg, ggl in cases:
        refs = [ho.mean_ap(q, gg, ql, ggl, R=R, ks=(1, 5, 10), remove_first=True) for R in (-1, 100)]
        for kw in (dict(records=False), dict(records=True), dict(records=True, rec_cap=1), dict(records=True, rec_cap=3),
                   dict(records=True, seg_rows=257)):
            got = rt.evaluate(_t(q, dev), _t(gg, dev), _t(ql, dev), _t(ggl, dev), R=[-1, 100], ks=(1, 5, 10), remove_first=True, **kw)
            for i in range(2):
                assert np.array_equal(got["S"][i].cpu().numpy().view(np.uint64), refs[i]["S"]), (name, kw, i)
                assert np.array_equal(got["nrel"][i].cpu().numpy().astype(np.uint32), refs[i]["nrel"]), (name, kw, i)
            assert np.array_equal(got["hits"].cpu().numpy().astype(np.uint32), refs[0]["hits"]), (name, kw)
    # multi-hot relevance goes through the single-row form of the scan
    qlab = (rng.random((700, 70)) < 0.05).astype(np.int64)
    glab = (rng.random((6000, 70)) < 0.05).astype(np.int64)
    ref = ho.mean_ap(q, g, qlab, glab, R=-1, ks=(1, 5, 10), remove_first=False)
    for kw in (dict(records=True), dict(records=True, rec_cap=2)):
        got = rt.evaluate(_t(q, dev), _t(g, dev), _t(qlab, dev), _t(glab, dev), R=-1, ks=(1, 5, 10), **kw)
        assert np.array_equal(got["S"].cpu().numpy().view(np.uint64), ref["S"]), kw
        assert np.array_equal(got["nrel"].cpu().numpy().astype(np.uint32), ref["nrel"]), kw


def test_evaluate_picks_the_two_scan_form_where_the_lists_would_overflow(dev, monkeypatch):
    """evaluate() left to its defaults predicts -- exactly, from the labels -- in how many (query tile, segment) workgroups a record
    list would overflow, and runs two scans where most would (a gallery listed class by class under shuffled queries: every list of
    the class's segment overflows; the one-scan form then pays its recording scan AND the redo).  The choice never changes a bit."""
    from concepthash_amd import retrieval as rt
    monkeypatch.setattr(rt, "OVERFLOW_MIN_PAIRS", 1)            # predict at this test's size too
    gen = torch.Generator(device=dev).manual_seed(3)
    Qn, G, ncls = 2048, 120_000, 12
    q = torch.randint(-2 ** 63, 2 ** 63 - 1, (Qn, 2), dtype=torch.int64, device=dev, generator=gen)
    g = torch.randint(-2 ** 63, 2 ** 63 - 1, (G, 2), dtype=torch.int64, device=dev, generator=gen)
    ql = torch.randint(0, ncls, (Qn,), dtype=torch.int32, device=dev, generator=gen)
    gl_sorted = (torch.arange(G, device=dev) * ncls // G).to(torch.int32)
    gl_shuffled = gl_sorted[torch.randperm(G, device=dev, generator=gen)]
    seg = rt.map_seg_rows(Qn, G, 2)
    cap = rt.record_cap(Qn, G, 2, seg)
    f_sorted, f_shuffled = rt.predicted_overflow(ql, gl_sorted, 2, seg, cap), rt.predicted_overflow(ql, gl_shuffled, 2, seg, cap)
    print(f"segments of {seg} rows, list capacity {cap}: predicted overflowing workgroups {f_sorted:.2f} (class-sorted gallery), "
          f"{f_shuffled:.2f} (shuffled)")
    assert f_sorted > 0.9 and f_shuffled == 0.0
    calls = []
    real = rt.hamming_hist_rec
    monkeypatch.setattr(rt, "hamming_hist_rec", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    ev_sorted = rt.evaluate(q, g, ql, gl_sorted)
    assert not calls                                            # two scans
    ev_shuffled = rt.evaluate(q, g, ql, gl_shuffled)
    assert calls == [1]                                         # one scan
    # the prediction is exact: the recording scan really overflows (almost) everywhere on the sorted listing, nowhere on the other
    _, (_, _, _, flags) = real(q, g, ql, gl_sorted, 0, seg)
    assert abs(float(flags.float().mean()) - f_sorted) < 1e-6
    _, (_, _, _, flags) = real(q, g, ql, gl_shuffled, 0, seg)
    assert int(flags.sum()) == 0
    for ev, gl in ((ev_sorted, gl_sorted), (ev_shuffled, gl_shuffled)):
        forced = rt.evaluate(q, g, ql, gl, records=True)
        assert torch.equal(ev["S"], forced["S"]) and torch.equal(ev["nrel"], forced["nrel"])
