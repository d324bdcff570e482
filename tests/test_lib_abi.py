"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/concepthash_hip.h declares.
No compute calls (there is no GPU here); argument validation paths are exercised because they run on the host."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from concepthash_amd import build, _lib
    build.build()
    return _lib.load()


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "concepthash_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ch_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from concepthash_amd import _lib
    declared = _declared_functions()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == declared


def test_abi_version_and_error_string(lib):
    assert lib.ch_abi_version() == 1
    st = lib.ch_pack_sign(None, -1, 0, 0.0, None, None)
    assert st != 0 and b"pack_sign" in lib.ch_last_error()
    st = lib.ch_hamming_topk(None, 4, None, 4, 9, 10, 0, None, None, None, 0, None)
    assert st != 0 and b"W" in lib.ch_last_error()
    st = lib.ch_hamming_topk(None, 4, None, 4, 1, 1000, 0, None, None, None, 0, None)
    assert st != 0 and b"k" in lib.ch_last_error()


def test_model_config_validation_runs_on_host(lib):
    from concepthash_amd import _lib
    bad = _lib.ModelConfig(image_size=224, patch=16, dim=100, layers=12, heads=12, ffn=3072, adapter_dim=384, ncontext=4,
                           nbit=64, nclass=200, proj_dim=512, center_dim=512, upt_heads=8, act=0, max_batch=8,
                           ln_eps=1e-5, bn_eps=1e-5)
    h = ctypes.c_void_p()
    t = (_lib.Tensor * 1)()
    assert lib.ch_model_create(ctypes.byref(bad), t, 1, ctypes.byref(h)) != 0
    assert b"dim" in lib.ch_last_error()
    assert not h.value


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from concepthash_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_product_code_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under concepthash_amd/ (or the drop-in surface modules) may touch it."""
    offenders = []
    for top in ("concepthash_amd", "models", "trainers", "utils", "experiments"):
        for dp, _, fns in os.walk(os.path.join(ROOT, top)):
            for fn in fns:
                if fn.endswith((".py", ".hip", ".h", ".cpp")):
                    src = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "oracle/" in src and fn.endswith(".py") \
                            and "import" in src and re.search(r"import.*hamming_oracle|import.*encoder_oracle", src):
                        offenders.append(os.path.join(dp, fn))
    assert not offenders, offenders
