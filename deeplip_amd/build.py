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
SOURCES = ["capi.hip", "plan.hip", "conv_igemm.hip", "conv_igemm_f16x3.hip", "conv_igemm_f16x3_dma.hip", "conv_win_f16x3.hip", "conv_rows_f16x3.hip", "stem3d.hip", "stem3d_f16x3.hip", "pool_ops.hip", "score_ops.hip", "layout_ops.hip", "train_ops.hip", "encoder_train_ops.hip", "video_train_ops.hip", "frontend_ops.hip"]
ARCH = "gfx950"


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 to build libdeeplip_hip.so)")


def _deps():
    return [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "dlip_common.h"), os.path.join(CSRC, "conv_common.h"), os.path.join(CSRC, "conv_dma_common.h"),
                                                       os.path.join(CSRC, "conv_dma_lab.inc"), os.path.join(CSRC, "conv_dma_hooks.h"), os.path.join(CSRC, "conv_dma_hook_consts.inc"),
                                                       os.path.join(CSRC, "conv_dma_hook_window.inc"), os.path.join(CSRC, "conv_dma_hook_tile256.inc"),
                                                       os.path.join(CSRC, "conv_dma_lab_menu.inc"), os.path.join(ROOT, "include", "deeplip_hip.h")]


def source_sha() -> str:
    """sha256 over every source and header the library is built from (names and bytes, in a fixed order): what
    dlip_source_sha() of a library built from THIS tree returns."""
    import hashlib
    h = hashlib.sha256()
    for d in _deps():
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


_MARK = b"DLIP_SOURCE_SHA="


def library_sha(path: str = None) -> str:
    """The source hash stamped into a built library, read from the file's bytes (no dlopen: a process that already holds an
    older copy of the library would be handed that one again); '' if the file has none."""
    try:
        with open(path or LIB, "rb") as f:
            blob = f.read()
    except OSError:
        return ""
    i = blob.find(_MARK)
    return blob[i + len(_MARK): i + len(_MARK) + 64].decode("ascii", "replace") if i >= 0 else ""


def needs_build() -> bool:
    """True unless the library exists AND was built from exactly these sources (hash stamped into it at link time -- file
    times say nothing once a tree has been copied, checked out or shipped to another box)."""
    if not os.path.exists(LIB):
        return True
    return library_sha() != source_sha()


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
    # the stamp: a generated translation unit that answers "which sources am I?" (include/deeplip_hip.h: dlip_source_sha)
    stamp_c = os.path.join(LIBDIR, tag + "stamp.cpp")
    stamp_o = os.path.join(LIBDIR, tag + "stamp.o")
    with open(stamp_c, "w") as f:
        f.write('extern "C" const char* dlip_source_sha(void) { static const char s[] = "DLIP_SOURCE_SHA=%s"; return s + 16; }\n' % source_sha())
    subprocess.check_call([hipcc, "-O2", "-fPIC", "-c", "-x", "c++", stamp_c, "-o", stamp_o])
    os.remove(stamp_c)
    objs.append(stamp_o)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    for o in objs:
        os.remove(o)
    return LIB


def prove_compile(source: str = "capi.hip", verbose: bool = True) -> float:
    """Compile ONE translation unit of the tree for gfx950 into a scratch object and throw it away: shows that the toolchain
    and the sources as they lie here still compile, whatever prebuilt library the tree carries.  Returns the seconds taken."""
    import tempfile
    import time
    with tempfile.TemporaryDirectory() as td:
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
               "-c", os.path.join(CSRC, source), "-o", os.path.join(td, "probe.o")]
        if verbose:
            print(" ".join(cmd), flush=True)
        t0 = time.time()
        subprocess.check_call(cmd)
        return time.time() - t0


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, lab="--lab" in sys.argv))


# What profiles/traffic_latest.json (PMC bytes per launch of the dominant kernel) is stamped with, and what bench.py compares the
# stamp to: the sources the dominant kernel's translation unit is compiled from.  Entry points added elsewhere (an ABI bump) do not
# change that kernel; an edit of a header it includes does.
DOMINANT_KERNEL_SOURCES = ("conv_igemm_f16x3_dma.hip", "conv_dma_common.h", "conv_dma_hooks.h", "conv_dma_hook_consts.inc", "conv_common.h", "dlip_common.h")


def dominant_kernel_sha() -> str:
    import hashlib
    h = hashlib.sha256()
    for name in DOMINANT_KERNEL_SOURCES:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


def box_id() -> str:
    """What tells one GPU box of the pool from another in a record: the first GPU's unique id (sysfs), else the host name."""
    import glob
    import socket
    for f in sorted(glob.glob("/sys/class/drm/card*/device/unique_id")):
        try:
            v = open(f).read().strip()
            if v:
                return "gpu-" + v
        except OSError:
            pass
    return socket.gethostname()
