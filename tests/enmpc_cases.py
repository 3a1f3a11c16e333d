"""Randomised reactor models for the economic path's tests and tools (tools/enmpc_fuzz.py, tools/enmpc_fuzz_cpu.py, tools/enmpc_prebuild.py): rate constants,
price, sampling time, both horizons and the estimator's update drawn from the seed; odd seeds above 4 also draw other boxes and make the disturbance bounds
bounds of the estimator (MPC_code.py:657-664) - the models on which round 3's restated interior point and IPOPT parted (a slack that rounds to zero, solves
of 150 full steps)."""
import numpy as np

FUZZ_SEEDS = list(range(5, 37))      # the 32 models of profiles/r03_enmpc_fuzz.txt / r04_enmpc_fuzz.txt
FUZZ_STEPS, FUZZ_STARTS = 12, 6
# solves that take more than 60 interior-point iterations on these models (C restatement, 6 starts x 12 steps x 3 NLPs x 32 models = 6912 solves): the
# target problem of model 7 - a cold start (MPC_code.py:696-700) whose input bounces between its bounds for 70 - 100 iterations before the iteration
# settles; everything else stays at or below 45
FUZZ_LONG_SOLVES = {7: ("ITERS_SS", 110)}


def draw(seed):
    rng = np.random.default_rng(1000 + seed)
    over = {"K1": float(rng.uniform(0.6, 1.6)), "K2": float(rng.uniform(0.02, 0.2)), "PRICE_B": float(rng.uniform(2.5, 6.0)), "h": float(rng.choice([1.0, 2.0, 3.0])),
            "N": int(rng.integers(8, 41)), "N_mhe": int(rng.integers(3, 15)), "mhe_up": str(rng.choice(["smooth", "filter"]))}
    if seed % 2 and seed > 4:
        over.update({"umax": [float(rng.uniform(0.8, 3.0))], "xmax": np.array([1.0, float(rng.uniform(0.5, 1.0))]), "dmin": np.array([-0.05, -0.02]), "dmax": np.array([0.03, 0.05])})
    x0 = rng.uniform([0.4, 0.0], [1.0, 0.6], size=(FUZZ_STARTS, 2))
    return over, x0
