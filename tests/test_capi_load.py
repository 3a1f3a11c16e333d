"""The C-ABI library: builds for gfx950, loads, exports every declared symbol, and fails loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, gpu_available


@pytest.fixture(scope="module")
def lib(pkg):
    from mpc_code_amd import capi
    capi.build_library()
    return capi.load_library()


def test_every_symbol_of_the_header_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "mpc_amd.h")).read()
    declared = set(re.findall(r"\b(mpc_[a-z_0-9]+)\s*\(", hdr))
    from mpc_code_amd import capi
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_build_info_names_the_target_and_the_compiled_dimensions(lib):
    info = lib.mpc_build_info().decode()
    assert info.startswith("gfx950") and "3/2/3/3/3/0" in info and "4/2/2/2/4/1" in info


def test_code_object_is_gfx950_only(lib):
    from mpc_code_amd import capi
    blob = open(capi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx90a", b"gfx942", b"sm_80", b"nvptx"):
        assert other not in blob


@pytest.mark.skipif(gpu_available(), reason="checks the no-GPU failure mode")
def test_create_fails_loudly_without_a_gpu(cstr):
    from mpc_code_amd import capi
    with pytest.raises(capi.MpcAmdError) as e:
        capi.Solver(cstr)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_missing_library_is_an_error_not_a_fallback(tmp_path):
    from mpc_code_amd import capi
    with pytest.raises(capi.MpcAmdError):
        capi.load_library(str(tmp_path / "libmpc_amd.so"))


def test_product_never_imports_the_oracle():
    pkgdir = os.path.join(ROOT, "mpc-code_amd")
    for dp, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "mpc_oracle" not in src and "riccati_np" not in src and "oracle_c" not in src, os.path.join(dp, f)
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), os.path.join(dp, f)


def test_oracle_never_imports_the_product():
    """The checker reads the examples with its own loader (oracle/exnum.py) and differentiates their functions numerically: no import of the
    product's package - loader, stand-ins, tracer, generated code - anywhere under oracle/."""
    for f in os.listdir(os.path.join(ROOT, "oracle")):
        if f.endswith((".py", ".c")):
            src = open(os.path.join(ROOT, "oracle", f), errors="ignore").read()
            assert not re.search(r"^\s*(from|import)\s+(mpc_code_amd|symtrace|exfile|nlcodegen|econcodegen)", src, re.M), f


def test_product_and_bench_do_not_use_pytorch():
    """north_star: host side is Python + ctypes, no PyTorch; the collective is RCCL inside the library."""
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    for dp, _, fs in os.walk(os.path.join(ROOT, "mpc-code_amd")):
        files += [os.path.join(dp, f) for f in fs if f.endswith((".py", ".hip", ".hpp", ".h"))]
    for f in files:
        src = open(f, errors="ignore").read()
        assert not re.search(r"^\s*(from|import)\s+torch", src, re.M), f
    hip = open(os.path.join(ROOT, "mpc-code_amd", "csrc", "mpc_amd.hip")).read()
    assert "ncclAllGather" in hip and "ncclCommInitRank" in hip
