"""MI355X-native batched linear MPC: the hot path of CPCLAB-UNIPI/MPC-code behind a C-ABI.

Layout
------
``exfile``   run an unmodified Ex-style problem file without CasADi -> namespace
``problem``  namespace -> numeric :class:`LinearMPCProblem` (DARE, bounds, estimator gains)
``symtrace`` / ``nlproblem``  non-linear examples: the Ex-file's model functions traced into expression DAGs
             (:class:`NonlinearMPCProblem`), differentiated and emitted as device code
``econproblem`` economic examples (user cost functions, continuous-time cost quadrature, moving-horizon estimator)
``capi``     ctypes binding of ``include/mpc_amd.h`` (``libmpc_amd.so``, hand-written HIP, gfx950)
``driver``   the closed loop of the reference's ``MPC_code.py:485-875`` over a batch of instances
``shard``    batch partition across ranks, rendezvous and host side of the all-gather of u* (RCCL inside the library)
``csrc/``    the HIP kernels and the C-ABI

Nothing here imports ``oracle/`` - that directory is test infrastructure.
"""
import os as _os

PKG_DIR = _os.path.dirname(_os.path.abspath(__file__))
EXAMPLES_DIR = _os.path.join(PKG_DIR, "examples")

from .exfile import load_exfile, DEFAULTS  # noqa: E402,F401
from .problem import LinearMPCProblem, UnsupportedProblem, problem_from_namespace  # noqa: E402,F401
from .nlproblem import NonlinearMPCProblem, nl_problem_from_namespace  # noqa: E402,F401
from .econproblem import EconomicMPCProblem, econ_problem_from_namespace, is_economic  # noqa: E402,F401


def load_problem(path, overrides=None):
    """Ex-style file -> :class:`LinearMPCProblem`, or :class:`NonlinearMPCProblem` when the model is a user function
    (``User_fxm_Cont`` / ``User_fxm_Dis``: the reference's own tests, MPC_code.py:94-111)."""
    ns = load_exfile(path, overrides)
    if is_economic(ns):      # user cost functions / moving-horizon estimator: the economic path (Ex_ENMPC.py)
        return econ_problem_from_namespace(ns)
    if ns.get("User_fxm_Cont") is not None or ns.get("User_fxm_Dis") is not None:
        return nl_problem_from_namespace(ns)
    return problem_from_namespace(ns)


def example_path(name):
    return _os.path.join(EXAMPLES_DIR, name)


def run_example(problem_or_path, x0_p=None, x0_m=None, nsteps=None, overrides=None, **kw):
    """The reference's ``python MPC_code.py`` for a batch: load an Ex-style file (or take a loaded problem) and run its closed loop on the
    GPU through the path the problem belongs to - linear (``driver.run_closed_loop``), non-linear tracking (``nmpc.run_nmpc_closed_loop``,
    ``max_sqp=`` SQP iterations per OCP) or economic with a moving-horizon estimator (``enmpc.run_enmpc_closed_loop``).  ``x0_p`` / ``x0_m``:
    [B, nxp] / [B, nx] plant and model start states (default: the file's own, one instance); ``nsteps`` default: the file's ``Nsim``.
    Returns the reference's result arrays (``MPC_code.py:877-895``: ``U, X_HAT, Y_HAT, XS, US, YS, Xp, Yp, D_HAT, TIME_*``) shaped
    ``[nsteps, B, dim]`` plus status / iteration words.  Raises without a GPU: there is no CPU path."""
    p = load_problem(problem_or_path, overrides) if isinstance(problem_or_path, (str, bytes, _os.PathLike)) else problem_or_path
    if isinstance(p, EconomicMPCProblem):
        from .enmpc import run_enmpc_closed_loop
        if x0_m is not None:
            raise ValueError("the economic path starts the model state from the file's x0_m / x_bar (MPC_code.py:449-463); pass x0_p only")
        return run_enmpc_closed_loop(p, p.x0_p[None] if x0_p is None else x0_p, nsteps, **kw)
    if isinstance(p, NonlinearMPCProblem):
        from .nmpc import run_nmpc_closed_loop
        return run_nmpc_closed_loop(p, x0_p, x0_m, nsteps, **kw)
    from .driver import run_closed_loop
    return run_closed_loop(p, x0_p, x0_m, nsteps, **kw)
