"""Build libdeeplip_hip.so (gfx950) in-tree with hipcc.

    python -m deeplip_amd.build            # rebuild if sources are newer than the library
    python -m deeplip_amd.build --lab      # libdeeplip_hip_lab.so: -DDLIP_LAB (experimental tiles, in-kernel stamps);
                                           # used by tools/ through DLIP_LIB_PATH, never by the product

hipcc cross-compiles without a GPU.  The library lands in deeplip_amd/lib/ so that it travels
with the repo snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libdeeplip_hip.so")
SOURCES = ["capi.hip", "plan.hip", "conv_igemm.hip", "conv_igemm_f16x3.hip", "conv_igemm_f16x3_dma.hip", "conv_win_f16x3.hip", "stem3d.hip", "stem3d_f16x3.hip", "pool_ops.hip", "score_ops.hip", "layout_ops.hip", "train_ops.hip", "encoder_train_ops.hip", "video_train_ops.hip", "frontend_ops.hip"]
ARCH = "gfx950"


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 to build libdeeplip_hip.so)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "dlip_common.h"), os.path.join(CSRC, "conv_common.h"), os.path.join(CSRC, "conv_dma_common.h"),
                                                       os.path.join(ROOT, "include", "deeplip_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, lab: bool = False) -> str:
    if lab:
        return _build(os.path.join(LIBDIR, "libdeeplip_hip_lab.so"), ["-DDLIP_LAB"], verbose, "lab_")
    if not force and not needs_build():
        return LIB
    return _build(LIB, [], verbose, "")


def _build(LIB: str, extra, verbose: bool, tag: str) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    hipcc = _hipcc()
    common = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
              "-I" + CSRC, "-Wall", "-Wno-unused-function"] + list(extra)
    procs = []
    for s in SOURCES:
        o = os.path.join(LIBDIR, tag + s.replace(".hip", ".o"))
        objs.append(o)
        cmd = common + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    for o in objs:
        os.remove(o)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, lab="--lab" in sys.argv))
