"""Shared by tests/test_nmpc.py and tests/golden/make_nmpc_golden.py: variations of the shipped non-linear examples."""
import numpy as np


def quadtank_mild_setpoints(t):
    """A set-point change the quadruple tank can follow without emptying an upper tank.  (The shipped schedule asks for level 6 in
    tank 2 at t > 50; the predicted level of tank 3 then sits at its lower bound 0, where the outflow law sqrt(2 g h) has no
    derivative - a comparison of derivative-based iterations is meaningless there.)"""
    usp = np.array([39.5185, 38.1743]); xsp = np.array([50.0, 50.0, 10.0, 10.0, 2.0, 2.0])
    return [np.array([11.9996, 12.1883]) if t <= 30 else np.array([10.5, 13.0]), usp, xsp]
