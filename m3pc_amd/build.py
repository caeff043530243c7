"""Build libm3pc_hip.so (gfx950) in-tree with hipcc.  No torch involvement: the library is a plain
C-ABI shared object (include/m3pc_hip.h).  ``python -m m3pc_amd.build`` or ``build_library()``."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libm3pc_hip.so")
SOURCES = ["gemm.hip", "gemm_glds.hip", "gemm_big.hip", "gemm_line.hip", "gemm_f32_direct.hip", "block_fused.hip", "attn.hip",
           "attn_bf16.hip", "elementwise.hip", "select.hip", "m3pc_plans.hip", "m3pc_passes.hip", "m3pc.hip"]
# the lab build (libm3pc_hip_lab.so, `python -m m3pc_amd.build --lab`, used by tools/ with M3PC_LIB=...): adds the experimental
# GEMM tilings, the timing variants of block_fused.hip and the M3PC_GEMM_VARIANT environment override (-DM3PC_LAB)
LAB_SOURCES = ["gemm_ring.hip", "gemm_persist.hip", "gemm_rs.hip"]
LAB_LIB = os.path.join(HERE, "libm3pc_hip_lab.so")
ARCH = "gfx950"
# per-file flags.  block_fused.hip: its gelu runs beside MFMAs, where packed fp32 VALU instructions cost more issue time
# than the two scalar ones they replace (MI355X_MICROARCH.md, cycle constants) -- keep the SLP vectorizer off there.
EXTRA_FLAGS = {"block_fused.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False, lab: bool = False) -> str:
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, n) for n in ("kernels.h", "gemm_epilogue.h", "gemm_stage_asm.h", "m3pc_internal.h", "exports.map")] + \
              [os.path.join(os.path.dirname(HERE), "include", n) for n in ("m3pc_hip.h", "m3pc_hip_debug.h")]
    objdir = os.path.join(CSRC, "build_lab" if lab else "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
             "-fno-gpu-rdc", "-ffp-contract=off", "-fvisibility=hidden"] + (["-DM3PC_LAB"] if lab else [])
    jobs = []
    objs = []
    lib = LAB_LIB if lab else LIB
    for s in SOURCES + (LAB_SOURCES if lab else []):
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([hipcc, *flags, *EXTRA_FLAGS.get(s, []), "-c", src, "-o", obj])

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
    if force or jobs or _stale(lib, objs):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", f"-Wl,--version-script={os.path.join(CSRC, 'exports.map')}", "-o", lib, *objs])
    return lib


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True, lab="--lab" in sys.argv))
