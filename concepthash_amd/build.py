"""Build libconcepthash_hip.so (gfx950) in-tree with hipcc.  `python -m concepthash_amd.build [--force]`."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libconcepthash_hip.so")
SOURCES = ["model.hip", "gemm_bf16.hip", "gemm_pp.hip", "gemm_r4.hip", "attention.hip", "rowops.hip", "head.hip", "small_f32.hip", "hamming.hip",
           "preprocess.hip", "train_kernels.hip", "attention_bwd.hip", "train.hip", "jpeg.hip", "jpeg_host.cpp", "errors.cpp"]
# plain C++ sources (no HIP): compiled by the same driver as host code; tests/test_jpeg.py also builds them with g++ -fsanitize=address,undefined
HOST_SOURCES = ["jpeg_host.cpp", "errors.cpp"]
# kernels that lost to the dispatched ones (DESIGN.md sections 3.8-3.9): kept in csrc/experiments/ with their parity tests, compiled
# only into an experiments build (CH_BUILD_EXPERIMENTS=1), never into the product library
EXPERIMENT_SOURCES = [os.path.join("experiments", f) for f in ("gemm_pq.hip", "gemm_ppp.hip", "gemm_dp.hip",
                                                                "adapter_fused.hip", "gemm_rows.hip", "gemm_wide.hip")]
HEADERS = ["ch_common.h", "ch_host.h", "kernels.h", "gemm_epilogue.h", "model_internal.h", os.path.join("..", "..", "include", "concepthash_hip.h"),
           os.path.join("..", "..", "include", "concepthash_hip_debug.h")]
# attention post-processes every MFMA result on the VALU: keep accumulators in VGPRs (no v_accvgpr_read round trips)
# preprocess reproduces Pillow's double-precision filter coefficients bit for bit: no fused multiply-adds there
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "preprocess.hip": ["-ffp-contract=off"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    experiments = os.environ.get("CH_BUILD_EXPERIMENTS", "0") == "1"
    stamp = os.path.join(OBJDIR, "experiments.flag")
    if (os.path.exists(stamp) and open(stamp).read().strip() == "1") != experiments:
        force = True                       # switching between the product and the experiments build recompiles everything
    with open(stamp, "w") as f:
        f.write("1" if experiments else "0")
    sources = SOURCES + (EXPERIMENT_SOURCES if experiments else [])
    flags = FLAGS + (["-DCH_EXPERIMENTS"] if experiments else [])
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    jobs = []
    objs = []
    for src in sources:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJDIR, os.path.basename(src).replace(".hip", ".o").replace(".cpp", ".o"))
        objs.append(op)
        if force or _stale(op, [sp] + hdrs):
            if src in HOST_SOURCES:       # plain C++: no offload pass
                jobs.append([hipcc, "-x", "c++", "-O3", "-std=c++17", "-fPIC", "-Wall", "-I", CSRC, "-c", sp, "-o", op])
            else:
                jobs.append([hipcc] + flags + ["-I", CSRC] + EXTRA_FLAGS.get(src, []) + ["-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
