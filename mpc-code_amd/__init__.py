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
