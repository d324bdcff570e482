"""ctypes/numpy front-end of oracle/hamming_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.
PARITY UNPINNED (see the header of hamming_oracle.c and DESIGN.md): the reference's ``utils.hashing`` is an
un-vendored, unpinned third-party module whose source is absent; this restates SURVEY.md section 8c's
normative definition and is anchored on the reference call sites cited there.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "hamming_oracle.c")
_SO = os.path.join(_HERE, "_build", "libhamming_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O2", "-march=x86-64-v2", "-shared", "-fPIC", "-o", _SO, _SRC])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.ho_bench_topk.restype = ctypes.c_uint64
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t)) if a is not None else None


def pack(codes: np.ndarray, threshold: float = 0.0) -> np.ndarray:
    codes = np.ascontiguousarray(codes, dtype=np.float32)
    rows, nbit = codes.shape
    W = (nbit + 63) // 64
    out = np.zeros((rows, W), dtype=np.uint64)
    lib().ho_pack(_p(codes, ctypes.c_float), ctypes.c_int64(rows), ctypes.c_int(nbit), ctypes.c_float(threshold),
                  _p(out, ctypes.c_uint64))
    return out


def dist(q: np.ndarray, g: np.ndarray) -> np.ndarray:
    q = np.ascontiguousarray(q, dtype=np.uint64)
    g = np.ascontiguousarray(g, dtype=np.uint64)
    out = np.zeros((q.shape[0], g.shape[0]), dtype=np.int32)
    lib().ho_dist(_p(q, ctypes.c_uint64), _p(g, ctypes.c_uint64), ctypes.c_int64(q.shape[0]),
                  ctypes.c_int64(g.shape[0]), ctypes.c_int(q.shape[1]), _p(out, ctypes.c_int32))
    return out


def topk(q: np.ndarray, g: np.ndarray, k: int):
    q = np.ascontiguousarray(q, dtype=np.uint64)
    g = np.ascontiguousarray(g, dtype=np.uint64)
    idx = np.zeros((q.shape[0], k), dtype=np.int32)
    dst = np.zeros((q.shape[0], k), dtype=np.int32)
    lib().ho_topk(_p(q, ctypes.c_uint64), _p(g, ctypes.c_uint64), ctypes.c_int64(q.shape[0]),
                  ctypes.c_int64(g.shape[0]), ctypes.c_int(q.shape[1]), ctypes.c_int(k),
                  _p(idx, ctypes.c_int32), _p(dst, ctypes.c_int32))
    return idx, dst


def pack_multihot(labels: np.ndarray) -> np.ndarray:
    """(rows, C) {0,1} -> (rows, ceil(C/64)) uint64 bitmasks."""
    labels = np.asarray(labels)
    rows, C = labels.shape
    LW = (C + 63) // 64
    out = np.zeros((rows, LW), dtype=np.uint64)
    nz_r, nz_c = np.nonzero(labels)
    np.bitwise_or.at(out, (nz_r, nz_c // 64), np.uint64(1) << (nz_c % 64).astype(np.uint64))
    return out


def mean_ap(q: np.ndarray, g: np.ndarray, q_labels: np.ndarray, g_labels: np.ndarray, R: int = -1,
            ks=(1, 5, 10), remove_first: bool = False, want_hist: bool = False, skip_queries_without_relevant: bool = False) -> dict:
    """labels: 1-D int class ids (single label) or 2-D {0,1} multi-hot.
    skip_queries_without_relevant: False = SURVEY.md section 8c (a query with no relevant row in its top R has AP 0 and counts in the
    mean); True = the HashNet / OrthoHash-family convention (`if tsum == 0: continue`: such queries are left out of the mean)."""
    q = np.ascontiguousarray(q, dtype=np.uint64)
    g = np.ascontiguousarray(g, dtype=np.uint64)
    Qn, W = q.shape
    G = g.shape[0]
    q_labels = np.asarray(q_labels)
    g_labels = np.asarray(g_labels)
    if q_labels.ndim == 1:
        ql32 = np.ascontiguousarray(q_labels, dtype=np.int32)
        gl32 = np.ascontiguousarray(g_labels, dtype=np.int32)
        qlm = glm = None
        LW = 0
    else:
        ql32 = gl32 = None
        qlm = pack_multihot(q_labels)
        glm = pack_multihot(g_labels)
        LW = qlm.shape[1]
    ks_a = np.ascontiguousarray(ks, dtype=np.int32)
    nk = len(ks_a)
    nb = 64 * W + 1
    S = np.zeros(Qn, dtype=np.uint64)
    nrel = np.zeros(Qn, dtype=np.uint32)
    ap = np.zeros(Qn, dtype=np.float64)
    hits = np.zeros((Qn, max(nk, 1)), dtype=np.uint32)
    total = np.zeros(Qn, dtype=np.uint32)
    hist = np.zeros((Qn, nb, 2), dtype=np.uint32) if want_hist else None
    lib().ho_map(_p(q, ctypes.c_uint64), _p(g, ctypes.c_uint64), _p(ql32, ctypes.c_int32), _p(gl32, ctypes.c_int32),
                 _p(qlm, ctypes.c_uint64), _p(glm, ctypes.c_uint64), ctypes.c_int(LW), ctypes.c_int64(Qn),
                 ctypes.c_int64(G), ctypes.c_int(W), ctypes.c_int64(R), ctypes.c_int(int(remove_first)),
                 _p(ks_a, ctypes.c_int32), ctypes.c_int(nk), _p(S, ctypes.c_uint64), _p(nrel, ctypes.c_uint32),
                 _p(ap, ctypes.c_double), _p(hits, ctypes.c_uint32), _p(total, ctypes.c_uint32),
                 _p(hist, ctypes.c_uint32))
    ap_fixed = np.where(nrel > 0, S.astype(np.float64) / (np.maximum(nrel, 1).astype(np.float64) * 4294967296.0), 0.0)
    ks_f = np.asarray(ks_a, dtype=np.float64)
    precisions = (hits[:, :nk] / ks_f[None, :]).mean(axis=0) if nk else np.zeros(0)
    recalls = np.where(total[:, None] > 0, hits[:, :nk] / np.maximum(total, 1)[:, None], 0.0).mean(axis=0) \
        if nk else np.zeros(0)
    if skip_queries_without_relevant:
        keep = nrel > 0
        m_fixed = float(ap_fixed[keep].mean()) if keep.any() else 0.0
        m_f64 = float(ap[keep].mean()) if keep.any() else 0.0
    else:
        m_fixed = float(ap_fixed.mean()) if Qn else 0.0
        m_f64 = float(ap.mean()) if Qn else 0.0
    return dict(S=S, nrel=nrel, ap_f64=ap, ap_fixed=ap_fixed, hits=hits[:, :nk], total=total, hist=hist,
                mAP=m_fixed, mAP_f64=m_f64, precisions=precisions, recalls=recalls)


# ---- reference-style float path (sign -> matmul -> argsort), the CPU-baseline "port" leg -----------------
def float_hamming(q_codes: np.ndarray, g_codes: np.ndarray) -> np.ndarray:
    """0.5*(nbit - sign(q).sign(g)^T): in-repo twin trainers/orthohash.py:263-264 without the /nbit."""
    a = np.where(q_codes > 0, 1.0, -1.0).astype(np.float32)
    b = np.where(g_codes > 0, 1.0, -1.0).astype(np.float32)
    return 0.5 * (a.shape[1] - a @ b.T)


def float_topk(q_codes: np.ndarray, g_codes: np.ndarray, k: int):
    d = float_hamming(q_codes, g_codes)
    order = np.argsort(d, axis=1, kind="stable")[:, :k]
    return order.astype(np.int32), np.take_along_axis(d, order, axis=1).astype(np.int32)


def bench_topk(q: np.ndarray, g: np.ndarray, k: int) -> int:
    q = np.ascontiguousarray(q, dtype=np.uint64)
    g = np.ascontiguousarray(g, dtype=np.uint64)
    return int(lib().ho_bench_topk(_p(q, ctypes.c_uint64), _p(g, ctypes.c_uint64), ctypes.c_int64(q.shape[0]),
                                   ctypes.c_int64(g.shape[0]), ctypes.c_int(q.shape[1]), ctypes.c_int(k)))


from concepthash_amd.synthetic import synthetic_codes  # noqa: E402,F401  (generator shared with bench.py)
