#!/usr/bin/env python3
"""Build (here, without a GPU) the per-model libraries of the randomised reactor models, so that the GPU box finds them in csrc/jit/.
   tools/enmpc_prebuild.py [first seed] [count] [workers]"""
import os, sys, warnings
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.simplefilter("ignore")


def build(seed):
    import mpc_code_amd as m
    from mpc_code_amd import econcodegen
    from enmpc_cases import draw
    over, _ = draw(seed)
    return seed, econcodegen.build_enmpc_library(m.load_problem(m.example_path("reactor_enmpc.py"), overrides=over))


if __name__ == "__main__":
    s0, n, w = (int(sys.argv[1]) if len(sys.argv) > 1 else 5), (int(sys.argv[2]) if len(sys.argv) > 2 else 32), (int(sys.argv[3]) if len(sys.argv) > 3 else 3)
    with ProcessPoolExecutor(w) as ex:
        for seed, path in ex.map(build, range(s0, s0 + n)):
            print(seed, os.path.basename(path), flush=True)
