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
