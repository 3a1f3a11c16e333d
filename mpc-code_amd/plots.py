"""The figures the reference's driver ends with (``MPC_code.py:897-935`` through ``makeplot``, ``Utilities.py:422-496``), for ONE instance of a batch: 'State',
'Input', 'Output' against their targets (and set points where the example defines ``defSP``) and 'Disturbance Estimate', one PDF per component named
``<label><k>.pdf`` under a directory, time on the x-axis, inputs as steps.  Host side, matplotlib's Agg backend; nothing here is on the hot path."""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import numpy as np


def _figure(plt, t, series, label, k, path, steps=False, legend=None):
    plt.figure()
    for s in series:
        (plt.step if steps else plt.plot)(t, s)
    plt.xlabel("Time "); plt.ylabel(label + str(k + 1))
    if legend and len(series) > 1:
        plt.legend(legend[:len(series)])
    plt.grid(True)
    f = os.path.join(path, f"{label}{k + 1}.pdf")
    plt.savefig(f, format="pdf", transparent=True, bbox_inches="tight")
    plt.close()
    return f


def make_plots(out: Dict[str, np.ndarray], h: float, path: str, instance: int = 0, ysp: Optional[np.ndarray] = None) -> List[str]:
    """``out``: the result arrays of a closed-loop run ([nsteps, B, dim], names of MPC_code.py:877-895); ``h``: the sampling time; returns the files written."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    os.makedirs(path, exist_ok=True)
    n = np.asarray(out["U"]).shape[0]
    t = np.linspace(0.0, (n - 1) * h, n)                    # MPC_code.py:898
    col = lambda name: None if name not in out else np.asarray(out[name], dtype=float)[:, instance]
    files = []
    for label, act, tgt, steps in (("State ", "X_HAT", "XS", False), ("Input ", "U", "US", True), ("Output ", "Yp", "YS", False)):      # :921-929
        a, g = col(act), col(tgt)
        if a is None:
            continue
        for k in range(a.shape[1]):
            series = [a[:, k]] + ([g[:, k]] if g is not None and g.shape[1] > k else [])
            if label == "Output " and ysp is not None:
                series.append(np.asarray(ysp, dtype=float)[:n, k])
            files.append(_figure(plt, t, series, label, k, path, steps, ("Actual", "Target", "Set-Point")))
    d = col("D_HAT")
    if d is not None:
        for k in range(d.shape[1]):
            files.append(_figure(plt, t, [d[:, k]], "Disturbance Estimate ", k, path))      # :931
    return files
