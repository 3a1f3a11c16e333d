"""One entry point for every example: mpc_code_amd.run_example / run_exfile.py (the reference's `python MPC_code.py` for a batch)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
EXAMPLES = ["cstr_lmpc.py", "wood_berry_lmpc.py", "cstr_nlplant_lmpc.py", "cstr_xp_nlplant_lmpc.py", "cstr_nmpc.py", "quadtank_nmpc_dis.py", "reactor_nmpc.py",
            "reactor_enmpc.py", "reactor_enmpc_ekf.py"]


def test_figures_of_the_reference_from_result_arrays(tmp_path):
    """mpc_code_amd.plots.make_plots on the golden economic loop (no GPU): State / Input / Output / Disturbance Estimate, one PDF per component (Utilities.py:422-496)"""
    from mpc_code_amd.plots import make_plots
    g = np.load(os.path.join(ROOT, "tests", "golden", "enmpc_reactor.npz"))
    out = {k: g["ship_" + k] for k in ("U", "X_HAT", "XS", "US", "D_HAT", "Xp")}
    out["Yp"], out["YS"] = out["Xp"], out["XS"] + out["D_HAT"]
    files = make_plots(out, 2.0, str(tmp_path / "fig"))
    assert [os.path.basename(f) for f in files] == ["State 1.pdf", "State 2.pdf", "Input 1.pdf", "Output 1.pdf", "Output 2.pdf", "Disturbance Estimate 1.pdf", "Disturbance Estimate 2.pdf"]
    assert all(open(f, "rb").read(5) == b"%PDF-" for f in files)


@pytest.mark.parametrize("ex", EXAMPLES)
def test_cli_loads_and_classifies_every_example(pkg, ex, capsys):
    import run_exfile
    assert run_exfile.main([pkg.example_path(ex), "--load-only", "-o", "N=12"]) == 0
    line = capsys.readouterr().out.strip().splitlines()[-1]
    assert line.startswith(ex + ": ") and "N=12" in line and ("Problem" in line)


@pytest.mark.gpu
@pytest.mark.parametrize("ex", EXAMPLES)
def test_gpu_every_example_runs_through_the_one_entry_point(pkg, ex, tmp_path, capsys):
    """Batch of five, four steps, the file's own start spread by 0.1 % (two of the examples are open-loop unstable): the reference's result arrays come back under their names with the
    right shapes, finite, inputs inside their bounds; the same numbers from run_example directly."""
    import warnings
    import run_exfile
    out = str(tmp_path / "r.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert run_exfile.main([pkg.example_path(ex), "--batch", "5", "--spread", "0.001", "--nsteps", "4", "--out", out, "--plots", str(tmp_path / "fig"), "--instance", "2"]) == 0
        p = pkg.load_problem(pkg.example_path(ex))
    figs = sorted(os.listdir(tmp_path / "fig"))      # the reference's figures (MPC_code.py:897-935), one PDF per component
    assert figs.count("Input 1.pdf") == 1 and len([f for f in figs if f.startswith("State ")]) == p.nx and len([f for f in figs if f.startswith("Input ")]) == p.nu, figs
    assert len([f for f in figs if f.startswith("Output ")]) == p.ny and all(os.path.getsize(tmp_path / "fig" / f) > 1000 for f in figs)
    r = np.load(out)
    for k, d in (("U", p.nu), ("X_HAT", p.nx), ("XS", p.nx), ("US", p.nu), ("Xp", len(p.x0_p))):
        assert r[k].shape == (4, 5, d) and np.isfinite(r[k]).all(), (ex, k)
    for k in ("Yp", "Y_HAT", "YS", "STATUS_DYN", "STATUS_SS"):
        assert k in r.files and r[k].shape[:2] == (4, 5), (ex, k)
    umin, umax = np.asarray(p.umin, dtype=float), np.asarray(p.umax, dtype=float)
    assert (r["U"] >= umin - 1e-7).all() and (r["U"] <= umax + 1e-7).all(), ex
    assert np.isin(r["STATUS_DYN"], (0, 1, 2)).all()      # (the shipped CSTR start holds its first steps: MPC_code.py:804-805)
    rng = np.random.default_rng(0)
    x0p = np.tile(np.asarray(p.x0_p, dtype=float), (5, 1)) * (1.0 + 0.001 * rng.uniform(-1, 1, size=(5, len(p.x0_p))))
    kw = {} if type(p).__name__ == "EconomicMPCProblem" else {"x0_m": np.tile(np.asarray(p.x0_m, dtype=float), (5, 1)) * (1.0 + 0.001 * rng.uniform(-1, 1, size=(5, len(p.x0_m))))}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        d = pkg.run_example(p, x0_p=x0p, nsteps=4, **kw)
    assert np.array_equal(d["U"], r["U"]) and np.array_equal(d["Xp"], r["Xp"]), ex
