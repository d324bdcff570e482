#!/usr/bin/env python3
"""Generate tests/golden/get_hd.npz by running the only retrieval arithmetic the reference snapshot holds:

  * ``get_hd`` (trainers/orthohash.py:263-264): normalised Hamming distance of two +-1 vectors, called by the reference with
    1-D vectors (trainers/orthohash.py:287) -- run here pair by pair, exactly that way;
  * ``calculate_accuracy_hamm_dist`` (utils/metrics.py:18-29): arg-min / 5-smallest accuracy over a (B, nclass) distance matrix.

Run in the build container only:   python -B oracle/gen_retrieval_golden.py
Both functions are imported UNMODIFIED from /root/reference.  ``trainers.orthohash`` imports modules that are absent from the
snapshot (hydra, and the un-vendored ``utils.io`` / ``utils.hashing`` / ``utils.misc`` of kamwoh/sdc): they get import-only
stand-ins; none of them is called by the two functions above.  Ranking / AP stay "parity unpinned" (their source,
``utils.hashing``, is not in the snapshot).  Output is data only: inputs, and the reference functions' outputs.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
# the repo has its own regular package `utils`, which would shadow the reference's namespace package: keep the repo root
# (and the script directory's parent) OFF sys.path while the reference modules are imported
sys.path = [p for p in sys.path if os.path.abspath(p or os.getcwd()) not in (ROOT, HERE)]
sys.path.insert(0, HERE)

import numpy as np
import torch

import _ref_shim as shim

GOLDEN = os.path.join(ROOT, "tests", "golden")


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        def _never(*a, **k):
            raise RuntimeError(f"stand-in {self.__name__}.{name} called: not part of the functions being pinned")
        return _never


def main():
    shim.install()        # omegaconf / torchvision / timm stand-ins + /root/reference first on sys.path
    for name in ("hydra", "hydra.utils", "utils.io", "utils.hashing", "utils.misc"):
        if name not in sys.modules:
            sys.modules[name] = _Stub(name)
    import utils.metrics as ref_metrics          # unmodified reference source
    import trainers.orthohash as ref_ortho       # unmodified reference source
    assert ref_metrics.__file__.startswith("/root/reference/") and ref_ortho.__file__.startswith("/root/reference/")

    payload = {}
    g = torch.Generator().manual_seed(20241108)
    for nbit, B, C in ((64, 48, 20), (128, 40, 12)):
        cb = torch.randn(C, nbit, generator=g).sign()
        cb[cb == 0] = 1.0
        labels = torch.randint(0, C, (B,), generator=g)
        codes = cb[labels].clone()
        flip = torch.rand(B, nbit, generator=g) < 0.3                # noisy copies of the class codewords
        codes[flip] *= -1.0
        # engineered ties: query 0 is equidistant from codewords 0 and 1 (differs from each in disjoint halves of the
        # positions where they differ); query 1 is a codeword itself (distance 0); query 2 is the complement of one
        if int((cb[0] != cb[1]).sum()) % 2:                          # an exact tie needs an even number of differing bits
            cb[1, int((cb[0] == cb[1]).nonzero()[0])] *= -1.0
        d01 = (cb[0] != cb[1]).nonzero().flatten()
        q0 = cb[0].clone()
        q0[d01[::2]] *= -1.0
        codes[0], labels[0] = q0, 1
        codes[1], labels[1] = cb[3].clone(), 3
        codes[2], labels[2] = -cb[4], 4
        hd = torch.empty(B, C)
        for i in range(B):
            for j in range(C):
                hd[i, j] = ref_ortho.get_hd(codes[i], cb[j])          # 1-D call, as at trainers/orthohash.py:287
        onehot = torch.nn.functional.one_hot(labels, C).float()
        acc1 = ref_metrics.calculate_accuracy_hamm_dist(hd, labels)
        acc1_onehot = ref_metrics.calculate_accuracy_hamm_dist(hd, onehot)
        acc5 = ref_metrics.calculate_accuracy_hamm_dist(hd, onehot, multiclass=True)
        tag = f"b{nbit}/"
        payload[tag + "codes"] = codes.numpy().astype(np.int8)
        payload[tag + "codebook"] = cb.numpy().astype(np.int8)
        payload[tag + "labels"] = labels.numpy().astype(np.int32)
        payload[tag + "get_hd"] = hd.numpy()                                  # fp32, = popcount(xor) / nbit
        payload[tag + "argmin"] = hd.argmin(1).numpy().astype(np.int64)       # what calculate_accuracy_hamm_dist ranks by
        payload[tag + "top5_smallest"] = hd.topk(5, 1, False, True)[1].numpy().astype(np.int64)
        payload[tag + "acc_argmin"] = np.float32(acc1)
        payload[tag + "acc_argmin_onehot"] = np.float32(acc1_onehot)
        payload[tag + "acc_top5"] = np.float32(acc5.item())
        print(tag, "hd range", float(hd.min()), float(hd.max()), "acc", float(acc1), float(acc5),
              "tie at query 0:", float(hd[0, 0]), float(hd[0, 1]))
    path = os.path.join(GOLDEN, "get_hd.npz")
    np.savez_compressed(path, **payload)
    print("->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
