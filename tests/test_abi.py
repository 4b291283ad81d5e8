"""CPU: the C-ABI library loads and exports exactly what include/gngf.h declares; bindings mirror the header."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "gngf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|int64_t)\s+(gngf_\w+)\s*\(([^)]*)\)\s*;", src):
        args = [a.strip() for a in m.group(2).split(",") if a.strip() and a.strip() != "void"]
        out[m.group(1)] = args
    return out


def test_library_exports_every_declared_symbol():
    from collision_handling_in_instantngp_amd import _lib
    lib = _lib.load()
    decl = header_functions()
    assert len(decl) >= 8
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/gngf.h but not exported"
    assert lib.gngf_abi_version() == _lib.ABI_VERSION == 13


def test_bindings_mirror_header_arity_and_kinds():
    import ctypes
    from collision_handling_in_instantngp_amd import _lib
    decl = header_functions()
    assert set(decl) == set(_lib.SIGNATURES), set(decl) ^ set(_lib.SIGNATURES)
    for name, args in decl.items():
        sig = _lib.SIGNATURES[name]
        assert len(sig) == len(args), (name, len(sig), len(args))
        for a, ct in zip(args, sig):
            if "*" in a:
                want = ctypes.c_void_p
            elif a.startswith("int64_t"):
                want = ctypes.c_int64
            elif a.startswith("float") or a.startswith("double"):
                want = ctypes.c_float if a.startswith("float") else ctypes.c_double
            else:
                want = ctypes.c_int
            assert ct is want, (name, a, ct)


def test_cpu_tensors_raise_no_fallback():
    import torch
    from collision_handling_in_instantngp_amd import _lib, ops
    with pytest.raises(_lib.GngfLibraryError):
        ops.hash_indices(torch.zeros(4, 2), torch.tensor([8, 16], dtype=torch.int32), 256)
