"""Builds libvoxtral_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.environ.get("VOX_LIB") or os.path.join(PKG_DIR, "libvoxtral_hip.so")   # VOX_LIB: measurement builds (ablations)
SOURCES = ["vox_kernels.hip", "vox_engine.hip", "vox_engine_b16.hip", "vox_api.cpp"]
HEADERS = ["vox_kernels.h", "vox_engine_common.h", os.path.join("..", "..", "include", "voxtral_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def variant_lib_path(tag: str) -> str:
    return os.path.join(PKG_DIR, f"libvoxtral_hip_{tag}.so")


def build_all(verbose: bool = False, force: bool = False, tag: str | None = None, extra_flags=()) -> str:
    """Product build (tag None): build/*.o -> libvoxtral_hip.so.  Measurement variants (`tag`, e.g. "timeline" with
    extra_flags=["-DVOX_TIMELINE"]): build/<tag>/*.o -> libvoxtral_hip_<tag>.so, selected at run time with VOX_LIB=<that path>."""
    bdir = os.path.join(PKG_DIR, "build", tag) if tag else os.path.join(PKG_DIR, "build")
    lib_path = variant_lib_path(tag) if tag else LIB_PATH
    os.makedirs(bdir, exist_ok=True)
    deps = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(bdir, src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [sp] + deps):
            jobs.append([HIPCC, *FLAGS, *extra_flags, "-x", "hip", "-c", sp, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=max(1, len(jobs))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(lib_path, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path, *objs])
    return lib_path


VARIANTS = {"timeline": ["-DVOX_TIMELINE"],      # measurement builds (tools/timeline.py)
            "pk_as_compiled": ["-DVOX_PK_AS_COMPILED"],      # rope / dense GEMV / split-K norm / batched engine with hipcc's own packed-FP32 forms (the MI355X hazard's encodings): A/B of what avoiding them costs
            "gemm_oldstage": ["-DVOX_GEMM_OLD_STAGING"],
            # q4_gemm_big_kernel, round-5 A/Bs (tools/gemm_big_ab.py, profiles/r05_gemm_big_slots.txt): the round-4 loop nest (one chain in flight, packed scale FMAs, single A-plane
            # buffer), the single-buffer form of the product loop, the hand-ordered slot form, and its timing-only ablations (wrong results)
            "big_serial": ["-DVOX_GEMM_BIG_SERIAL", "-DVOX_GEMM_BIG_PKFMA", "-DVOX_GEMM_BIG_SBUF"], "big_sbuf": ["-DVOX_GEMM_BIG_SBUF"], "big_slots": ["-DVOX_GEMM_BIG_SLOTS"],
            "big_slots_nomfma": ["-DVOX_GEMM_BIG_SLOTS", "-DVOX_ABL_S_NOMFMA"], "big_slots_nofma": ["-DVOX_GEMM_BIG_SLOTS", "-DVOX_ABL_S_NOFMA"], "big_slots_nods": ["-DVOX_GEMM_BIG_SLOTS", "-DVOX_ABL_S_NODS"],      # q4_gemm_kernel with the pre-round-5 LDS staging map (16-way bank conflicts on the writes): same-box A/B
            # GEMV ablations (tools/gemv_ablate.py; results are wrong by construction, only the timing is read)
            "abl_noscale": ["-DVOX_ABL_NOSCALE"], "abl_nox": ["-DVOX_ABL_NOX"], "abl_wfirst": ["-DVOX_ABL_WFIRST"],      # (abl_wfirst predates the x-first default and is a no-op now)
            "abl_noconsume": ["-DVOX_ABL_NOCONSUME"], "abl_xfirst_resid": ["-DVOX_ABL_XFIRST_RESID_ONLY"], "abl_xfirst_noswiglu": ["-DVOX_ABL_XFIRST_NO_SWIGLU"], "abl_noreduce": ["-DVOX_ABL_NOREDUCE"],
            "abl_all": ["-DVOX_ABL_NOSCALE", "-DVOX_ABL_NOX", "-DVOX_ABL_NOCONSUME", "-DVOX_ABL_NOREDUCE"]}

if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1:
        print(build_all(verbose=True, tag=sys.argv[1], extra_flags=VARIANTS.get(sys.argv[1], sys.argv[2:])))
    else:
        print(build_all(verbose=True))
