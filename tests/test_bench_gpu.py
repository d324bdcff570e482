"""The driver's contract with bench.py, on the GPU: one JSON line with the agreed keys, the encode-only form that the rocprofv3 passes
trace, per-shape roofline rows whose traffic (when the tracked table has the shape) is never below the algorithmic bytes, and the
single-chain roofline pass reproducing the timed region's codes bit for bit."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_encode_only_bench_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--encode-only", "--steps", "4", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, PYTHONUNBUFFERED="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # ONE JSON line on stdout
    b = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "roofline_per_kernel", "roofline_pass", "encode_mfma_frac_end_to_end"):
        assert key in b, key
    assert b["n_gpus"] == 1 and b["steps"] == 4 and b["warmup"] == 1 and b["higher_is_better"] is True and b["scaling"] == "weak"
    assert b["unit"] == "images/s" and b["dtype"] == "bf16" and b["data"].startswith("synthetic") and "workload" in b["config"]
    assert b["vs_baseline"] is None                                # BASELINE.md has no published number for this metric
    assert b["value"] > 10000 and abs(b["value"] - 256 / b["ms_per_step"] * 1e3) / b["value"] < 1e-3
    ro = b["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "rocprof_name", "grid"):
        assert key in ro, key
    assert ro["bound"] == "mfma" and ro["peak"] == 2500.0 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3 and 0.2 < ro["frac"] < 1.0
    assert b["roofline_pass"]["codes_identical_to_timed_region"] is True
    names = set()
    for row in b["roofline_per_kernel"]:
        names.add(row["rocprof_name"])
        assert row["launches_per_step"] >= 11 and row["avg_launch_us"] > 10
        if row["traffic"] is not None:                            # a counter figure below the algorithmic bytes is refused, never printed
            assert row["traffic"] >= 0.98 * row["algorithmic_bytes_per_launch"], row
    # one problem shape per row, under the names rocprofv3 prints: the cache-policy instances at this size
    assert {"gemm_pp_kernel<6, 0, 0, 0>", "gemm_pp_kernel<6, 0, 0, 1>", "gemm_pp_kernel<9, 0, 0, 2>", "gemm_pp_kernel<8, 0, 0, 2>",
            "gemm_bf16_kernel<7, true>", "gemm_bf16_kernel<10, false>"} <= names, names
