import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_fixture(name):
    """tests/golden/<name>.npz -> (state_dict of torch tensors, npz handle)"""
    import numpy as np
    import torch
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    # "sdbf/..." = bf16 bit patterns (uint16) of tensors that are exactly bf16-representable (oracle/gen_golden.py)
    sd.update({k[5:]: bf16_bits_to_f32(z[k]) for k in z.files if k.startswith("sdbf/")})
    return sd, z


def bf16_bits_to_f32(a):
    import numpy as np
    import torch
    return torch.from_numpy((a.astype(np.uint32) << 16).view(np.float32).copy())


def fixture_images(z):
    """input images of a fixture: fp32 ("in/images") or bf16 bit patterns ("inbf/images")"""
    import torch
    if "in/images" in z.files:
        return torch.from_numpy(z["in/images"])
    return bf16_bits_to_f32(z["inbf/images"])


@pytest.fixture(autouse=True)
def _always_split_encode_chains(monkeypatch):
    """ch_encode falls back to ONE launch chain for small batches (option "chain_auto", default on: fewer than 5,600 token rows).  The
    fixtures of this suite are that small: without this every two-chain test would silently run one chain.  `test_chain_auto_rule`
    removes the variable again."""
    monkeypatch.setenv("CH_CHAIN_AUTO", "0")
