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


def _declared_functions(headers=("concepthash_hip.h", "concepthash_hip_debug.h")):
    names = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(ch_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_debug_taps_live_in_their_own_header():
    """The drop-in boundary (concepthash_hip.h) declares no ch_debug_* tap; they are all in concepthash_hip_debug.h."""
    assert not [n for n in _declared_functions(("concepthash_hip.h",)) if n.startswith("ch_debug_")]
    dbg = _declared_functions(("concepthash_hip_debug.h",))
    assert len(dbg) >= 10 and all(n.startswith("ch_debug_") for n in dbg)


def test_every_declared_symbol_is_exported_and_bound(lib):
    from concepthash_amd import _lib
    declared = _declared_functions()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == declared


def test_abi_version_and_error_string(lib):
    from concepthash_amd import _lib
    header = open(os.path.join(ROOT, "include", "concepthash_hip.h")).read()
    assert lib.ch_abi_version() == 3 == _lib.ABI_VERSION == int(re.search(r"#define CH_ABI_VERSION (\d+)", header).group(1))
    st = lib.ch_pack_sign(None, -1, 0, 0.0, None, None)
    assert st != 0 and b"pack_sign" in lib.ch_last_error()
    st = lib.ch_hamming_topk(None, 4, None, 4, 9, 10, 0, None, None, None, 0, None)
    assert st != 0 and b"W" in lib.ch_last_error()
    st = lib.ch_hamming_topk(None, 4, None, 4, 1, 1000, 0, None, None, None, 0, None)
    assert st != 0 and b"k" in lib.ch_last_error()


def test_the_library_reads_no_environment_variable():
    """"no hidden global state except an opaque handle".  Every knob is ch_model_set_option state of one handle;
    the CH_* variables of earlier rounds survive as debug overrides read by the PYTHON wrapper only.  No product kernel source calls
    getenv, every option key the wrapper knows is documented in the header, and every environment override maps onto such a key."""
    from concepthash_amd import _lib, build
    for src in build.SOURCES + build.HEADERS:
        text = open(os.path.join(build.CSRC, src)).read()
        assert "getenv" not in text, f"{src} reads the environment"
    header = open(os.path.join(ROOT, "include", "concepthash_hip.h")).read()
    for key in _lib.OPTION_KEYS:
        assert f'"{key}"' in header, f"option {key} is not documented in the header"
    assert {k for k, _ in _lib._ENV_OVERRIDES.values()} <= set(_lib.OPTION_KEYS)
    model_src = open(os.path.join(build.CSRC, "model.hip")).read()
    assert set(re.findall(r'\{"([a-z_]+)", [012], CH_OPT_FIELD', model_src)) == set(_lib.OPTION_KEYS)


def test_option_and_capacity_arguments_are_validated_on_the_host(lib):
    """ch_model_set_option / get_option / profile_end refuse null handles (no GPU needed to see the argument checks)."""
    v = ctypes.c_int64()
    assert lib.ch_model_set_option(None, b"streams", 1) != 0 and b"null model" in lib.ch_last_error()
    assert lib.ch_model_get_option(None, b"streams", ctypes.byref(v)) != 0
    assert lib.ch_model_profile_end(None, 20, None, None, None) != 0


def test_env_overrides_are_read_in_the_wrapper(monkeypatch):
    from concepthash_amd import _lib
    for k in _lib._ENV_OVERRIDES:
        monkeypatch.delenv(k, raising=False)
    assert _lib.env_option_overrides() == {}
    monkeypatch.setenv("CH_STREAMS", "1")
    monkeypatch.setenv("CH_RESID_NT", "0")
    monkeypatch.setenv("CH_NT_OUT", "1")
    assert _lib.env_option_overrides() == {"streams": 1, "resid_nt": -1, "nt_out": 1}


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


def _hazard_tool():
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_dpp_hazards", os.path.join(ROOT, "tools", "check_dpp_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_hand_placed_instructions_of_the_map_scans_pass_the_static_checks():
    """csrc/hamming.hip broadcasts gallery rows through DPP operands inside inline asm and keeps an inline-asm VMEM ring behind
    counted `s_waitcnt vmcnt(N)` -- neither is visible to the compiler's hazard recogniser / wait insertion.  tools/check_dpp_hazards.py
    disassembles the built object and checks, over the control-flow graph of every kernel: the VALU-write -> DPP-read distance
    (2 wait states) and the EXEC-write -> DPP distance (5), that no instruction touches the destination of an outstanding
    vector-memory load before the vmcnt wait that covers it, and that the map scans have no scratch (spill) traffic."""
    obj = os.path.join(ROOT, "concepthash_amd", "csrc", "build", "hamming.o")
    if not os.path.exists(obj):
        from concepthash_amd import build
        build.build(verbose=False)
    r = _hazard_tool().check_all(obj)
    assert r["dpp_total"] > 1000, "the DPP form of the scans was not built"
    assert not r["dpp_bad"], r["dpp_bad"][:5]
    assert r["vmem_loads"] > 500 and not r["vmem_bad"], r["vmem_bad"][:5]
    assert not r["scratch_bad"], r["scratch_bad"][:5]


def test_no_built_kernel_selects_the_high_half_of_src1_in_a_packed_fp32_instruction():
    """Rule 5 of tools/check_dpp_hazards.py over every object of the product library: `v_pk_{fma,mul,add}_f32` with op_sel set on
    SRC1 -- measured on MI355X to return a wrong low lane whenever a wave of another kernel executes MFMAs on the same SIMD
    (profiles/r03_pk_opsel_hazard.txt, tools/pk_opsel_repro.py; the LN-fold epilogue's (rstd, mean) table layout exists to keep the
    compiler away from it) -- must not appear in whatever the compiler emitted for this build."""
    mod = _hazard_tool()
    objs = mod.product_objects()
    if not all(os.path.exists(o) for o in objs):
        from concepthash_amd import build
        build.build(verbose=False)
    seen = 0
    for o in objs:
        if not mod.has_device_code(o):
            assert not mod.expects_device_code(o), f"{o}: kernels in the source but no extractable device code"
            continue
        total, bad = mod.check_pk_opsel(o)
        seen += total
        assert not bad, (os.path.basename(o), bad[:5])
    assert seen > 5000                                   # the GEMM epilogues are full of packed-fp32 arithmetic: the scan saw them
    # and the rule fires on the measured instruction, not on its safe spellings
    m = mod.PK_F32.match("v_pk_fma_f32 v[62:63], v[62:63], v[106:107], v[110:111] op_sel:[0,1,0]")
    assert m and m.group(3) == "1"
    m = mod.PK_F32.match("v_pk_fma_f32 v[62:63], v[106:107], v[62:63], v[110:111] op_sel:[1,0,0]")
    assert m and m.group(3) == "0"
    assert mod.PK_F32.match("v_pk_fma_f32 v[62:63], v[62:63], v[106:107], v[110:111] op_sel_hi:[1,0,1]") is None


def test_the_static_checker_catches_what_it_claims_to():
    """Synthetic listings: a use of a ring register before its wait (also across a loop back-edge), a VALU write two slots in front
    of a DPP read reached only through a branch, and a v_cmpx in front of a DPP instruction -- each must be reported; the corrected
    listings must be clean."""
    mod = _hazard_tool()

    def kern(lines):
        rows, addr = [], 0x1000
        labels = {}
        for ln in lines:                      # "L1:" defines a label, "... @L1" is a branch target
            if ln.endswith(":"):
                labels[ln[:-1]] = addr
                continue
            addr += 4
        addr = 0x1000
        for ln in lines:
            if ln.endswith(":"):
                continue
            cmt = f" {addr:012X}: 00000000"
            if "@" in ln:
                ln, lab = ln.split("@")
                cmt += f" <k+0x{labels[lab] - 0x1000:x}>"
            rows.append((addr, ln.strip(), cmt))
            addr += 4
        return mod.Kernel("k", rows)

    # ring register v4 consumed under vmcnt(1) although ONE younger load is allowed to be outstanding -> it is not covered
    bad = kern(["global_load_dword v4, v[0:1], off", "global_load_dword v5, v[2:3], off", "s_waitcnt vmcnt(2)", "v_add_u32_e32 v6, v4, v4",
                "s_endpgm"])
    assert len(bad.check_vmem_order()[1]) == 1
    ok = kern(["global_load_dword v4, v[0:1], off", "global_load_dword v5, v[2:3], off", "s_waitcnt vmcnt(1)", "v_add_u32_e32 v6, v4, v4",
               "s_waitcnt vmcnt(0)", "v_add_u32_e32 v6, v5, v5", "s_endpgm"])
    assert not ok.check_vmem_order()[1]
    # loop: the load at the bottom is consumed at the top of the next iteration without a wait on the back-edge path
    loop = kern(["global_load_dword v4, v[0:1], off", "s_waitcnt vmcnt(0)", "L:", "v_add_u32_e32 v6, v4, v4",
                 "global_load_dword v4, v[0:1], off", "s_cbranch_scc1 0@L", "s_waitcnt vmcnt(0)", "s_endpgm"])
    # (the re-issue into the still-pending v4 is reported too: a write under an outstanding load)
    assert [x[2] for x in loop.check_vmem_order()[1]] == ["v_add_u32_e32 v6, v4, v4", "global_load_dword v4, v[0:1], off"]
    # DPP: the writer sits in front of a branch INTO the DPP instruction (listing order shows harmless instructions before it)
    dpp = kern(["v_mov_b32_e32 v7, v1", "s_branch 0@T", "s_nop 4", "s_nop 4", "T:", "v_xor_b32_dpp v2, v7, v3 row_newbcast:1 row_mask:0xf bank_mask:0xf",
                "s_endpgm"])
    assert [b[0] for b in dpp.check_dpp()[1]] == ["valu->dpp"]
    fixed = kern(["v_mov_b32_e32 v7, v1", "s_nop 1", "s_branch 0@T", "T:", "v_xor_b32_dpp v2, v7, v3 row_newbcast:1 row_mask:0xf bank_mask:0xf",
                  "s_endpgm"])
    assert not fixed.check_dpp()[1]
    cx = kern(["v_cmpx_eq_u32_e32 v1, v2", "s_nop 1", "v_xor_b32_dpp v2, v7, v3 row_newbcast:1 row_mask:0xf bank_mask:0xf", "s_endpgm"])
    assert [b[0] for b in cx.check_dpp()[1]] == ["exec->dpp"]
