"""CPU-side checks of the drop-in boundary: the in-tree library loads, exports every symbol
that include/deeplip_hip.h declares, and the ctypes binding covers exactly that set.
No compute call is made (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "deeplip_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dlip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from deeplip_amd import build
    path = build.build(verbose=False)
    lib = ctypes.CDLL(path)
    syms = header_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in deeplip_hip.h but not exported"


def test_binding_matches_header():
    from deeplip_amd import _lib
    assert sorted(list(_lib.SIGNATURES) + ["dlip_error_string", "dlip_source_sha"]) == header_symbols()
    l = _lib.lib()
    assert l.dlip_abi_version() == _lib.ABI_VERSION
    from deeplip_amd import build
    assert l.dlip_source_sha().decode() == build.source_sha() == build.library_sha()      # the library IS these sources
    assert l.dlip_error_string(0) == b"ok"
    assert b"invalid argument" in l.dlip_error_string(-1)


def test_ops_refuse_cpu_tensors():
    import torch
    from deeplip_amd import ops
    from deeplip_amd._lib import DeepLipHipError
    with pytest.raises(DeepLipHipError):
        ops.meanstd_pool(torch.zeros(1, 4, 8))
    with pytest.raises(DeepLipHipError):
        ops.conv_nhwc(torch.zeros(1, 2, 2, 4), torch.zeros(4, 1, 1, 4))


def test_binding_arity_matches_header():
    """Every ctypes signature has as many arguments as the header's prototype (ctypes would push a short or long list without
    complaint; the kernel would then read garbage pointers)."""
    from deeplip_amd import _lib
    text = open(os.path.join(ROOT, "include", "deeplip_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    seen = 0
    for m in re.finditer(r"\b(dlip_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        if name in _lib.SIGNATURES:
            assert len(_lib.SIGNATURES[name]) == n, f"{name}: header has {n} arguments, binding {len(_lib.SIGNATURES[name])}"
            seen += 1
    assert seen == len(_lib.SIGNATURES)
