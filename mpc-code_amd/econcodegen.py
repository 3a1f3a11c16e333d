"""Device code for a traced economic problem (:class:`EconomicMPCProblem`), and the per-model library build.

The interior point method of ``csrc/mpc_enmpc.hpp`` needs, for every shooting interval, the discrete map AND its first and second
derivatives with respect to the interval's initial state and input (the reference gets them from CasADi's AD through the unrolled
integrator, ``Control_Calc.py:102-111,153-158``; IPOPT uses the exact Hessian).  They are propagated through the Runge-Kutta
stages as forward sensitivities: with ``X(p)`` the stage value as a function of the parameters ``p = (x_0, u)``,

    K   = f(X, u)
    dK  = f_z dZ                                   Z = (X, u),  dZ = (dX, [0 I])
    d2K = f_z d2Z + sum_ce f_zz[c][e] dZ_c dZ_e

``emit_rhs`` writes exactly these three lines as straight-line code for one traced right-hand side, with every structurally zero
entry of ``f_z`` / ``f_zz`` and every zero of the unit rows dropped at generation time (the bilinear reactor of ``Ex_ENMPC.py`` has
3 non-zero second derivatives out of 27).  Scalar functions (target cost, terminal cost, estimator cost) are emitted with dense
gradient and Hessian.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
from typing import Dict, List, Sequence

import numpy as np

from . import PKG_DIR
from . import symtrace as st

CSRC = os.path.join(PKG_DIR, "csrc")
# -amdgpu-spill-vgpr-to-agpr=0: a code-generator fault of this toolchain (AMD clang 22.0.0git, roc-7.2.0), found in round 5 in enmpc_mhe_kernel<64>: with that option on (the
# default) a 128-bit register tuple was spilled as three dwords to scratch and the fourth into a spare accumulation register ("Reload Reuse" in the -S dump), and RELOADED into
# an accumulation-register tuple as the three dwords alone - the upper half of the tuple's second double came back as whatever its register held (the estimator's second
# disturbance estimate, 36 of 36 randomised models).  Without the option the same spill is four dwords to scratch and back; register and scratch sizes of
# every kernel are unchanged.  (Not the fault of round 3's broadcasts-in-scalar-registers builds: those were wrong with this option off too, profiles/r03_enmpc_bcast_matrix.txt.)
# -disable-machine-licm: the compiler hoists loop invariants out of the interior point iteration - the polynomial coefficients of exp and log as vector-register copies among them -
# and, having no registers for them across the integration and the sweeps, spills them (204 B of scratch per lane in enmpc_ocp_kernel<64>, reloaded one dependent scratch load
# after the other in every iteration).  Without machine LICM they are rebuilt where they are used: enmpc_ocp_kernel<64> has no scratch frame, the estimator kernels 520 B for 690 B,
# and the two economic workloads run 4.6 % / 3.5 % faster with the same bits in every result (profiles/r05_enmpc_variants.json).
ENMPC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ldl", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0", "-mllvm", "-disable-machine-licm"]


def _pp(j: int, k: int, NP: int) -> int:
    """index of the pair (j <= k) in the packed upper triangle"""
    if j > k:
        j, k = k, j
    return j * NP - j * (j - 1) // 2 + (k - j)


def _ind(code: str, n: int = 8) -> str:
    pad = " " * n
    return "\n".join(pad + l for l in code.split("\n") if l.strip() or True)


def emit_rhs(name: str, exprs: Sequence[st.Sym], svars: Sequence[st.Sym], pvars: Sequence[st.Sym], vm: Dict[str, str]) -> str:
    """``struct name``: ``eval0`` (values) and ``eval2`` (values + first and second forward sensitivities) of the right-hand side
    ``exprs`` (one per state row; rows beyond ``len(svars)`` - a cost quadrature - are not arguments of any expression).  Columns of
    the sensitivities: the ``len(svars)`` initial states, then the ``len(pvars)`` parameters (inputs)."""
    NR, NCX, NPAR = len(exprs), len(svars), len(pvars)
    NP = NCX + NPAR
    NPP = NP * (NP + 1) // 2
    zv = list(svars) + list(pvars)
    J = st.jacobian(exprs, zv)
    H = [st.jacobian(J[i], zv) for i in range(NR)]      # H[i][c][e]
    outs, names = list(exprs), [f"K[{i}]" for i in range(NR)]
    code0 = st.emit_cpp(outs, names, vm)
    # named derivative values
    dnames: Dict[int, str] = {}
    dexpr, dn = [], []
    for i in range(NR):
        for c in range(NP):
            if not J[i][c].is_const(0.0):
                nm = f"j{i}_{c}"; dexpr.append(J[i][c]); dn.append(nm)
            for e in range(c, NP):
                if not H[i][c][e].is_const(0.0):
                    nm = f"h{i}_{c}_{e}"; dexpr.append(H[i][c][e]); dn.append(nm)
    decl = ("double " + ", ".join(dn) + ";\n") if dn else ""
    code2 = decl + st.emit_cpp(outs + dexpr, names + dn, vm)
    lines: List[str] = []
    dZ = lambda c, j: (f"dX[{c}][{j}]" if c < NCX else ("1" if j == c else None))      # unit rows of the parameters
    for i in range(NR):
        for j in range(NP):
            terms = []
            for c in range(NP):
                if J[i][c].is_const(0.0):
                    continue
                z = dZ(c, j)
                if z is None:
                    continue
                terms.append(f"j{i}_{c}" + ("" if z == "1" else f" * {z}"))
            lines.append(f"dK[{i}][{j}] = " + (" + ".join(terms) if terms else "0.0") + ";")
        for j in range(NP):
            for k in range(j, NP):
                terms = []
                for c in range(NCX):
                    if not J[i][c].is_const(0.0):
                        terms.append(f"j{i}_{c} * d2X[{c}][{_pp(j, k, NP)}]")
                for c in range(NP):
                    for e in range(NP):
                        hce = H[i][min(c, e)][max(c, e)]
                        if hce.is_const(0.0):
                            continue
                        a, b = dZ(c, j), dZ(e, k)
                        if a is None or b is None:
                            continue
                        fac = [f"h{i}_{min(c, e)}_{max(c, e)}"] + [t for t in (a, b) if t != "1"]
                        terms.append(" * ".join(fac))
                lines.append(f"d2K[{i}][{_pp(j, k, NP)}] = " + (" + ".join(terms) if terms else "0.0") + ";")
    body2 = code2 + "\n" + "\n".join(lines)
    return f"""    struct {name} {{
        static constexpr int NR = {NR}, NCX = {NCX}, NP = {NP}, NPP = {NPP};
        __device__ static __forceinline__ void eval0(const double *X, const Ctx &c, double t, double *K)
        {{
            (void)X; (void)c; (void)t;
{_ind(code0, 12)}
        }}
        __device__ static __forceinline__ void eval2(const double *X, const Ctx &c, double t, const double (*dX)[NP], const double (*d2X)[NPP],
                                                     double *K, double (*dK)[NP], double (*d2K)[NPP])
        {{
            (void)X; (void)c; (void)t; (void)dX; (void)d2X;
{_ind(body2, 12)}
        }}
    }};
"""


def emit_scalar(name: str, expr: st.Sym, zv: Sequence[st.Sym], vm: Dict[str, str], sig: str) -> str:
    """value, gradient and (dense, symmetric) Hessian of a scalar function of ``zv``"""
    n = len(zv)
    g = [st.diff(expr, v) for v in zv]
    Hs = st.jacobian(g, zv)
    outs, names = [expr], ["*f"]
    for i in range(n):
        outs.append(g[i]); names.append(f"g[{i}]")
        for j in range(n):
            outs.append(Hs[i][j]); names.append(f"H[{i}][{j}]")
    return f"""    __device__ static __forceinline__ void {name}({sig}, double *f, double *g, double (*H)[{n}])
    {{
{_ind(st.emit_cpp(outs, names, vm), 8)}
    }}
"""


def emit_rows(name: str, exprs: Sequence[st.Sym], zv: Sequence[st.Sym], vm: Dict[str, str]) -> str:
    """``struct name``: values, Jacobian and (packed, per row) Hessian of the vector function ``exprs`` of ``zv`` = (x, u): the user rows of the OCP"""
    n, NP = len(exprs), len(zv)
    NPP = NP * (NP + 1) // 2
    if n == 0:
        return ""
    J = st.jacobian(exprs, zv)
    outs, names = [], []
    for r in range(n):
        outs.append(exprs[r]); names.append(f"g[{r}]")
        for c in range(NP):
            outs.append(J[r][c]); names.append(f"J[{r}][{c}]")
        Hr = st.jacobian(J[r], zv)
        for c in range(NP):
            for e in range(c, NP):
                outs.append(Hr[c][e]); names.append(f"H[{r}][{_pp(c, e, NP)}]")
    return f"""    struct {name} {{
        static constexpr int NG = {n}, NP = {NP}, NPP = {NPP};
        __device__ static __forceinline__ void eval(const double *X, const Ctx &c, double t, double *g, double (*J)[NP], double (*H)[NPP])
        {{
            (void)X; (void)c; (void)t;
{_ind(st.emit_cpp(outs, names, vm), 12)}
        }}
    }};
"""


def emit_econ_header(p) -> str:
    nx, nu, ny, nd, nxp, nw = p.nx, p.nu, p.ny, p.nd, p.nxp, p.n_w
    vx, vu = st.symvec("x", nx), st.symvec("u", nu)
    vm = {f"x[{i}]": f"X[{i}]" for i in range(nx)}
    vm.update({f"u[{i}]": f"c.u[{i}]" for i in range(nu)})
    vm.update({f"d[{i}]": f"c.d[{i}]" for i in range(nd)})
    vm.update({f"xs[{i}]": f"c.xs[{i}]" for i in range(nx)})
    vm.update({f"us[{i}]": f"c.us[{i}]" for i in range(nu)})
    vm["t"] = "t"
    vmp = {f"xp[{i}]": f"X[{i}]" for i in range(nxp)}
    vmp.update({f"u[{i}]": f"c.u[{i}]" for i in range(nu)})
    vmp["t"] = "t"
    ocp = emit_rhs("Ocp", list(p.f) + [p.ell], vx, vu, vm)                 # [f + px; l]: ContForm has no Bd d (Control_Calc.py:103)
    mdl = emit_rhs("Mdl", list(p.f), vx, vu, vm)
    mhe = emit_rhs("Mhe", list(p.f_mhe), vx, [], vm)
    plant = emit_rhs("Plant", list(p.fp), st.symvec("xp", nxp), [], vmp)
    nv = nx + nu + ny
    vss = st.symvec("xs", nx) + st.symvec("us", nu) + st.symvec("ys", ny)
    vmss = {f"xs[{i}]": f"w[{i}]" for i in range(nx)}
    vmss.update({f"us[{i}]": f"w[{nx + i}]" for i in range(nu)})
    vmss.update({f"ys[{i}]": f"w[{nx + nu + i}]" for i in range(ny)})
    fss = emit_scalar("fss", p.fss, vss, vmss, "const double *w")
    vmv = {f"x[{i}]": f"x[{i}]" for i in range(nx)}
    vmv.update({f"xs[{i}]": f"xs[{i}]" for i in range(nx)})
    vfin = emit_scalar("vfin", p.vfin, vx, vmv, "const double *x, const double *xs")
    vwv = st.symvec("w", nw) + st.symvec("v", ny)
    vmw = {f"w[{i}]": f"wv[{i}]" for i in range(nw)}
    vmw.update({f"v[{i}]": f"wv[{nw + i}]" for i in range(ny)})
    vmw["t"] = "t"
    cmhe = emit_scalar("cmhe", p.c_mhe, vwv, vmw, "const double *wv, double t")
    gin = emit_rows("Gin", list(getattr(p, "g_ineq", []) or []), list(vx) + list(vu), vm)
    if gin:      # (a model without user rows keeps the header it always had)
        gin = "    // User_g_ineq(x, u, y, d, t, px, py) <= 0, the OCP's user rows at every stage   (Control_Calc.py:94-100,132-147)\n#define MPC_EC_HAS_GIN 1\n" + gin
    def cmat(name, a):      # a constant matrix as a constexpr function of its indices: zeros and ones are known to the compiler where the loops unroll
        a = [[float(v) for v in row] for row in a]
        terms = "".join(f"(i == {i} && j == {j}) ? {v!r} : " for i, row in enumerate(a) for j, v in enumerate(row) if v != 0.0)
        return f"    __host__ __device__ static constexpr double {name}(int i, int j) {{ return {terms}0.0; }}\n"
    consts = cmat("Bd", p.Bd) + cmat("Cd", p.Cd) + cmat("G", p.G_mhe)
    return f"""// GENERATED by mpc-code_amd/econcodegen.py from the traced functions of '{p.name}' - do not edit.
#pragma once
struct EcModel {{
    static constexpr int NX = {nx}, NU = {nu}, NY = {ny}, ND = {nd}, NXP = {nxp}, NW = {nw}, MX = {p.Mx};
    static constexpr bool W_BOUNDS = {"true" if (np.isfinite(p.wmin).any() or np.isfinite(p.wmax).any()) else "false"};      // the estimator's state noise has bounds (wmin / wmax): its solver then carries them
    // disturbance model and noise input of the example (offree = 'lin': Bd, Cd, Utilities.py:174-177,202-204; G_mhe, MPC_code.py:387): part of the problem
    // definition, compiled in so that the estimator's recursions skip their zeros and ones; enmpc_create checks the descriptor against them
{consts}    struct Ctx {{ double u[NU], d[ND], xs[NX], us[NU]; }};      // what a right-hand side reads besides its state
    // shooting interval of the ContForm OCP: [User_fxm_Cont + px; User_fobj_Cont]   (Control_Calc.py:102-111)
{ocp}
    // the model alone: target's fixed-point equation, hold rule   (Utilities.py:157-183)
{mdl}
    // the estimator's model, User_fx_mhe_Cont   (Utilities.py:749-761)
{mhe}
    // the plant, User_fxp_Cont   (Utilities.py:58-82)
{plant}
    // User_fssobj(xs, us, ys) in w = [xs; us; ys]   (Target_Calc.py:109-124)
{fss}
    // User_vfin(x, xs)   (Control_Calc.py:194-210)
{vfin}
    // User_fobj_mhe(w, v, t) in wv = [w; v]   (Utilities.py:928-932)
{cmhe}{gin}}};
"""


def enmpc_library_path(header_text: str) -> str:
    inc = os.path.join(os.path.dirname(PKG_DIR), "include")
    srcs = [os.path.join(CSRC, f) for f in ("mpc_enmpc.hip", "mpc_enmpc.hpp", "mpc_rk4s2.hpp", "mpc_device.hpp", "mpc_sym.hpp", "mpc_tp.hpp", "mpc_comm.hpp")] + [os.path.join(inc, "mpc_enmpc.h")]
    hsh = hashlib.sha256(header_text.encode())
    hsh.update(" ".join(ENMPC_FLAGS).encode())
    for s in srcs:
        hsh.update(open(s, "rb").read())
    return os.path.join(CSRC, "jit", f"libmpc_enmpc_{hsh.hexdigest()[:16]}.so")


def build_enmpc_library(p, verbose: bool = False, extra_flags: Sequence[str] = ()) -> str:
    """Compile the economic-MPC kernels for this model (hipcc --offload-arch=gfx950, in-tree under csrc/jit/); returns the path."""
    text = emit_econ_header(p)
    out = enmpc_library_path(text + " ".join(extra_flags))
    if os.path.exists(out):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    hdr = out[:-3] + "_model.hpp"
    with open(hdr, "w") as fh:
        fh.write(text)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tmp = out + f".{os.getpid()}.tmp"
    cmd = [hipcc, *ENMPC_FLAGS, *extra_flags, f'-DMPC_EC_MODEL_HEADER="{hdr}"', "-o", tmp, os.path.join(CSRC, "mpc_enmpc.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(tmp, out)
    return out
