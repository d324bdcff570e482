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


def test_map_pass_heuristics_stay_inside_the_kernels_limits():
    """retrieval.map_seg_rows / record_cap (host arithmetic only): segments of 256 .. 65,535 rows (16-bit counters) that cover the
    gallery, whole rounds of the LDS-limited workgroup slots where the sizes allow it, and record lists inside their budgets."""
    from concepthash_amd import _lib, retrieval as rt
    lib = _lib.load()
    for Qn, G, W in [(1, 1, 1), (5794, 5994, 1), (24633, 23929, 1), (16384, 1_000_000, 2), (333, 4567, 1), (100, 300, 4),
                     (50000, 50000, 2), (10, 70000, 2), (5, 10_000_000, 2), (25250, 75750, 1), (1500, 9000, 3), (7, 70_000_000, 4)]:
        seg = rt.map_seg_rows(Qn, G, W)
        assert 256 <= seg <= 65535 or seg >= G, (Qn, G, W, seg)
        nseg = -(-G // seg)
        assert nseg <= 65535 and nseg * seg >= G
        blk = int(lib.ch_hamming_rec_block(W))
        assert blk == (256 if W <= 2 else 128)
        wgs = int(lib.ch_hamming_rec_workgroups(Qn, G, W, seg))
        assert wgs == (-(-Qn // blk)) * nseg
        cap = rt.record_cap(Qn, G, W, seg)
        assert 1 <= cap <= seg
        assert cap >= min(seg, seg // 64 + 32) or wgs * blk * 8 * (cap + 1) > rt.REC_BUDGET_BYTES
        assert wgs * blk * 8 * cap <= rt.REC_BUDGET_BYTES
        assert rt.records_fit(Qn, G, W, seg) == (cap >= min(seg, seg // 64 + 32))
    assert rt.records_fit(16384, 1_000_000, 2, rt.map_seg_rows(16384, 1_000_000, 2))
    assert not rt.records_fit(200_000, 20_000_000, 2, rt.map_seg_rows(200_000, 20_000_000, 2))   # 4 GiB cannot hold the minimum lists
    # whole rounds: NABirds size at 64 bit -> one round of the 512 slots; the 1M x 128-bit problem -> exactly four of 256
    assert -(-24633 // 256) * -(-23929 // rt.map_seg_rows(24633, 23929, 1)) <= 512
    assert -(-16384 // 256) * -(-1_000_000 // rt.map_seg_rows(16384, 1_000_000, 2)) == 1024


def test_hand_placed_dpp_instructions_have_no_read_after_valu_write_hazard():
    """csrc/hamming.hip broadcasts gallery rows through DPP operands inside inline asm, where the compiler cannot insert the two
    wait states gfx9 needs between a VALU write of a VGPR and a DPP read of it.  tools/check_dpp_hazards.py disassembles the built
    object and checks every DPP instruction (a guard against a future compiler moving a copy in front of one)."""
    import importlib.util
    obj = os.path.join(ROOT, "concepthash_amd", "csrc", "build", "hamming.o")
    if not os.path.exists(obj):
        from concepthash_amd import build
        build.build(verbose=False)
    spec = importlib.util.spec_from_file_location("check_dpp_hazards", os.path.join(ROOT, "tools", "check_dpp_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    total, bad = mod.check(obj)
    assert total > 1000, "the DPP form of the scans was not built"
    assert not bad, bad[:5]
