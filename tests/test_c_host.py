"""The C-ABI from C: the public headers are self-contained C99, and a C host program links against libmpc_amd.so and gets the LQR known answer."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

HOST = os.path.join(ROOT, "mpc-code_amd", "examples", "c_host")
INC = os.path.join(ROOT, "include")
CSRC = os.path.join(ROOT, "mpc-code_amd", "csrc")


def test_headers_are_self_contained_c99(tmp_path):
    """gcc -std=c99 -pedantic -Werror on a file that includes the three headers and takes the address of every entry point they declare."""
    src = open(os.path.join(HOST, "abi_check.c")).read()
    declared = set()
    for h in ("mpc_amd.h", "mpc_nmpc.h", "mpc_enmpc.h"):
        declared |= set(re.findall(r"^(?:int|void|float|const char \*|void \*)\s*\*?\s*((?:mpc|nmpc|enmpc)_[a-z0-9_]+)\s*\(", open(os.path.join(INC, h)).read(), flags=re.M))
    assert len(declared) > 60
    full = "#include \"mpc_amd.h\"\n#include \"mpc_nmpc.h\"\n#include \"mpc_enmpc.h\"\ntypedef void (*fn)(void);\nfn all_entry_points[] = {" + ", ".join(f"(fn){n}" for n in sorted(declared)) + "};\n"
    for name, text in (("abi_check.c", src), ("abi_all.c", full)):
        f = tmp_path / name
        f.write_text(text)
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", INC, "-c", str(f), "-o", str(tmp_path / (name + ".o"))])
        subprocess.check_call(["g++", "-x", "c++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-Wno-old-style-cast", "-I", INC, "-c", str(f), "-o", str(tmp_path / (name + ".oo"))])      # and as C++


def _build_host(tmp_path):
    exe = str(tmp_path / "lqr_host")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", INC, os.path.join(HOST, "lqr_host.c"), "-o", exe, "-L", CSRC, "-lmpc_amd", "-lm"])
    env = dict(os.environ, LD_LIBRARY_PATH=CSRC + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    return exe, env


def test_c_host_links_and_is_refused_loudly_without_a_gpu(pkg, tmp_path):
    exe, env = _build_host(tmp_path)
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
    if r.returncode == 0:      # (a box with a GPU: the run itself is test_gpu_c_host_gets_the_lqr_law)
        return
    assert r.returncode == 2 and "no HIP device" in r.stderr and "no CPU fallback" in r.stderr, (r.returncode, r.stderr)


@pytest.mark.gpu
def test_gpu_c_host_gets_the_lqr_law(pkg, tmp_path):
    exe, env = _build_host(tmp_path)
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok: first moves equal the LQR law" in r.stdout, (r.returncode, r.stdout, r.stderr)
    assert r.stdout.count("status 0") == 5


@pytest.mark.gpu
def test_gpu_c_host_of_the_economic_seam_follows_the_golden_loop(pkg, tmp_path):
    """examples/c_host/enmpc_host.c - plain C99, the per-call seam of include/mpc_enmpc.h (enmpc_mhe_update, enmpc_target_solve, enmpc_ocp_solve, the plant)
    linked against the shipped example's per-model library - walks the golden closed loop of tests/golden/enmpc_reactor.npz."""
    import numpy as np
    from mpc_code_amd import econcodegen
    lib = econcodegen.build_enmpc_library(pkg.load_problem(pkg.example_path("reactor_enmpc.py")))
    exe = str(tmp_path / "enmpc_host")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", INC, os.path.join(HOST, "enmpc_host.c"), lib, "-o", exe, "-lm"])
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.dirname(lib) + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, "5"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    g = np.load(os.path.join(ROOT, "tests", "golden", "enmpc_reactor.npz"))
    rows = [l for l in r.stdout.splitlines() if l.startswith("step") and "instance 0" in l]
    assert len(rows) == 5
    for k, l in enumerate(rows):
        v = l.split()
        assert abs(float(v[5]) - g["ship_U"][k, 0, 0]) < 1e-7 and abs(float(v[7]) - g["ship_XS"][k, 0, 0]) < 1e-7 and abs(float(v[10]) - g["ship_US"][k, 0, 0]) < 1e-7, l
        assert v[12] == "0/0/0" and int(v[14].split("/")[0]) == int(g["ship_ITERS_DYN"][k, 0]), l
