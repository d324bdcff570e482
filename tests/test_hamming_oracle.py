"""CPU: the Hamming/mAP oracle (oracle/hamming_oracle.c) against independent numpy restatements and against the one
retrieval formula that IS in the reference snapshot (get_hd, trainers/orthohash.py:263-264)."""
import numpy as np
import pytest

from oracle import hamming_oracle as ho


def _unpack(p, nbit):
    return ((p[:, :, None] >> np.arange(64, dtype=np.uint64)[None, None, :]) & np.uint64(1)).reshape(p.shape[0], -1)[:, :nbit]


@pytest.mark.parametrize("nbit", [16, 64, 70, 128])
def test_pack_bit_order(nbit):
    rng = np.random.default_rng(nbit)
    codes = rng.standard_normal((37, nbit)).astype(np.float32)
    codes[0, :] = 0.0  # sign(0) -> bit 0
    p = ho.pack(codes)
    assert p.shape == (37, (nbit + 63) // 64)
    assert (_unpack(p, nbit) == (codes > 0)).all()
    p2 = ho.pack(codes, threshold=0.25)
    assert (_unpack(p2, nbit) == ((codes - np.float32(0.25)) > 0)).all()


@pytest.mark.parametrize("nbit", [64, 128])
def test_distance_equals_reference_get_hd_formula(nbit):
    """get_hd(a, b) = 0.5 * (nbit - a @ b.T) / nbit on +-1 codes (trainers/orthohash.py:263-264) == popcount(xor)/nbit."""
    rng = np.random.default_rng(1)
    a = rng.standard_normal((40, nbit)).astype(np.float32)
    b = rng.standard_normal((90, nbit)).astype(np.float32)
    sa, sb = np.where(a > 0, 1.0, -1.0), np.where(b > 0, 1.0, -1.0)
    get_hd = 0.5 * (nbit - sa @ sb.T) / nbit
    d = ho.dist(ho.pack(a), ho.pack(b))
    assert np.array_equal(d, np.rint(get_hd * nbit).astype(np.int32))
    assert np.array_equal(d, ho.float_hamming(a, b).astype(np.int32))


def test_topk_is_stable_by_distance_then_index():
    q, _ = ho.synthetic_codes(33, 64, seed=3, nclass=4, flip=0.02)   # few distinct codes -> many ties
    g, _ = ho.synthetic_codes(500, 64, seed=4, nclass=4, flip=0.02)
    d = ho.dist(q, g)
    idx, dst = ho.topk(q, g, 25)
    order = np.lexsort((np.broadcast_to(np.arange(500), d.shape), d), axis=1)[:, :25]
    assert np.array_equal(idx, order)
    assert np.array_equal(dst, np.take_along_axis(d, order, axis=1))
    idx2, dst2 = ho.topk(q[:2], g[:3], 5)      # k > G pads with -1
    assert (idx2[:, 3:] == -1).all() and (dst2[:, 3:] == -1).all()


def _ap_reference_style(d_row, rel_row, R):
    """HashNet/OrthoHash-style AP: argsort (stable), tgnd = gnd[:R], mean(count / tindex)."""
    order = np.argsort(d_row, kind="stable")
    tg = rel_row[order][:R]
    n = int(tg.sum())
    if n == 0:
        return 0.0
    count = np.linspace(1, n, n)
    tindex = np.where(tg == 1)[0] + 1.0
    return float(np.mean(count / tindex))


@pytest.mark.parametrize("R", [-1, 20])
@pytest.mark.parametrize("remove_first", [False, True])
def test_map_against_numpy(R, remove_first):
    q, ql = ho.synthetic_codes(40, 64, seed=5, nclass=7)
    g, gl = ho.synthetic_codes(300, 64, seed=6, nclass=7)
    r = ho.mean_ap(q, g, ql, gl, R=R, ks=(1, 5, 10), remove_first=remove_first, want_hist=True)
    d = ho.dist(q, g)
    rel = (ql[:, None] == gl[None, :]).astype(np.int64)
    aps = []
    for i in range(40):
        order = np.argsort(d[i], kind="stable")
        if remove_first:
            order = order[1:]
        dd, rr = d[i][order], rel[i][order]
        RR = len(order) if R <= 0 else min(R, len(order))
        aps.append(_ap_reference_style(np.arange(len(order)), rr, RR))  # already ranked
        for t, k in enumerate((1, 5, 10)):
            assert r["hits"][i, t] == rr[:k].sum()
        assert r["total"][i] == rr.sum()
    assert np.allclose(r["ap_f64"], aps, atol=1e-12)
    assert np.allclose(r["ap_fixed"], aps, atol=1e-9)   # 2^-32 fixed point vs float64
    # histogram = bucket counts
    for i in (0, 17):
        for b in range(65):
            assert r["hist"][i, b, 0] == (d[i] == b).sum()
            assert r["hist"][i, b, 1] == ((d[i] == b) & (rel[i] == 1)).sum()


@pytest.mark.parametrize("R", [3, 10, -1])
def test_both_ap_conventions_on_queries_without_a_relevant_row_in_the_top_R(R):
    """The one convention that changes numbers at R < G ("Retrieval definition"): a query with NO relevant row in
    its top R.  Default: AP 0, counted in the mean.  `skip_queries_without_relevant=True` (the HashNet /
    OrthoHash-family evaluators: `if tsum == 0: continue`): left out of the mean.  Both against an independent numpy restatement, on a
    set built so that some queries have none (a class that is absent from the gallery, and rare classes ranked late), and the same
    two means out of `retrieval.summarize` (the host code the HIP path shares) from the oracle's integers."""
    import torch
    from concepthash_amd import retrieval as rt
    q, ql = ho.synthetic_codes(60, 64, seed=15, nclass=12, flip=0.3)
    g, gl = ho.synthetic_codes(400, 64, seed=16, nclass=12, flip=0.3)
    ql[:5] = 11                       # queries of a class ...
    gl[gl == 11] = 0                  # ... that the gallery does not hold: no relevant row at any R
    d = ho.dist(q, g)
    rel = (ql[:, None] == gl[None, :]).astype(np.int64)
    G = g.shape[0]
    RR = G if R <= 0 else R
    aps, has = [], []
    for i in range(60):
        order = np.argsort(d[i], kind="stable")
        rr = rel[i][order]
        aps.append(_ap_reference_style(np.arange(G), rr, RR))
        has.append(rr[:RR].sum() > 0)
    aps, has = np.array(aps), np.array(has)
    assert 5 <= (~has).sum() < 60     # the set exercises the convention (more queries without a hit at small R)
    r_all = ho.mean_ap(q, g, ql, gl, R=R)
    r_skip = ho.mean_ap(q, g, ql, gl, R=R, skip_queries_without_relevant=True)
    assert np.array_equal(r_all["S"], r_skip["S"]) and np.array_equal(r_all["nrel"], r_skip["nrel"])   # integers do not depend on it
    assert abs(r_all["mAP_f64"] - aps.mean()) < 1e-12 and abs(r_all["mAP"] - aps.mean()) < 1e-9
    assert abs(r_skip["mAP_f64"] - aps[has].mean()) < 1e-12 and abs(r_skip["mAP"] - aps[has].mean()) < 1e-9
    assert r_skip["mAP"] > r_all["mAP"]
    # the product's host-side summary from the same integers
    S = torch.from_numpy(r_all["S"].view(np.int64))[None]
    nrel = torch.from_numpy(r_all["nrel"].astype(np.int32))[None]
    total = torch.from_numpy(r_all["total"].astype(np.int32))
    for skip, want in ((False, r_all["mAP"]), (True, r_skip["mAP"])):
        sm = rt.summarize(S, nrel, total, [0], [R], [], skip_queries_without_relevant=skip)
        assert abs(sm["mAPs"][0] - want) < 1e-12
    # no query with a hit at all: 0.0 under both, no division by zero
    e = ho.mean_ap(q[:5], g, ql[:5], gl, R=R, skip_queries_without_relevant=True)
    assert e["mAP"] == 0.0
    z = rt.summarize(S[:, :5], nrel[:, :5], total[:5], [0], [R], [], skip_queries_without_relevant=True)
    assert z["mAPs"][0] == 0.0


def test_multihot_equals_single_label():
    q, ql = ho.synthetic_codes(20, 128, seed=7, nclass=70)
    g, gl = ho.synthetic_codes(200, 128, seed=8, nclass=70)
    a = ho.mean_ap(q, g, ql, gl)
    b = ho.mean_ap(q, g, np.eye(70, dtype=np.int64)[ql], np.eye(70, dtype=np.int64)[gl])
    assert np.array_equal(a["S"], b["S"]) and np.array_equal(a["nrel"], b["nrel"])
    # genuinely multi-label: relevant <=> share >= 1 class
    rng = np.random.default_rng(0)
    qm = (rng.random((20, 70)) < 0.05).astype(np.int64)
    gm = (rng.random((200, 70)) < 0.05).astype(np.int64)
    c = ho.mean_ap(q, g, qm, gm, ks=(3,))
    d = ho.dist(q, g)
    rel = (qm @ gm.T) > 0
    for i in range(20):
        order = np.argsort(d[i], kind="stable")
        assert c["hits"][i, 0] == rel[i][order][:3].sum()


def test_edge_cases():
    q, ql = ho.synthetic_codes(3, 64, seed=1, nclass=2)
    z = np.zeros((5, 1), dtype=np.uint64)           # all-zero codes: every distance ties
    r = ho.mean_ap(q, z, ql, np.zeros(5, np.int32))
    assert r["S"].shape == (3,)
    idx, dst = ho.topk(q, z, 4)
    assert (idx == np.arange(4)[None, :]).all()
    e = ho.mean_ap(q, np.zeros((0, 1), np.uint64), ql, np.zeros(0, np.int32))    # empty gallery
    assert e["mAP"] == 0.0 and (e["nrel"] == 0).all()
    e2 = ho.mean_ap(np.zeros((0, 1), np.uint64), z, np.zeros(0, np.int32), np.zeros(5, np.int32))  # no queries
    assert e2["mAP"] == 0.0
