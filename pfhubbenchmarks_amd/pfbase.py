"""Host-side mirror of the helpers of the reference's dolfin/pfbase.py that sit on the BM1 / BM6 hot path.

Same names and argument meaning where a counterpart makes sense on a uniform grid:
  InitialConditionsBench1(c0, epsilon)   pfbase.py:177-193   (numpy evaluation; the GPU path uses pf_set_ic_bm1)
  InitialConditionsBench6(c0, c1)        pfbase.py:322-339
  f_chem / dfdc                          bench1.py:63-65
  total_solute / total_free_energy       bench1.py:121-125  (thin wrappers over solver.diagnostics())
"""
import numpy as np


class InitialConditionsBench1:
    def __init__(self, c0=0.5, epsilon=0.05, **kwargs):
        self.c0, self.epsilon = c0, epsilon

    def eval(self, x, y):
        """c(x, y); mu = 0 (pfbase.py:187-190)"""
        return self.c0 + self.epsilon * (np.cos(0.105 * x) * np.cos(0.11 * y)
                                         + (np.cos(0.13 * x) * np.cos(0.087 * y)) ** 2
                                         + np.cos(0.025 * x - 0.15 * y) * np.cos(0.07 * x - 0.02 * y))

    def on_grid(self, nx, ny, h):
        x = np.arange(nx) * h
        y = np.arange(ny) * h
        return self.eval(x[None, :], y[:, None])


class InitialConditionsBench6(InitialConditionsBench1):
    def __init__(self, c0=0.5, c1=0.04, **kwargs):
        self.c0, self.epsilon = c0, c1

    def eval(self, x, y):
        """pfbase.py:332-334: first wavenumber 0.2 instead of 0.105; mu = phi = 0"""
        return self.c0 + self.epsilon * (np.cos(0.2 * x) * np.cos(0.11 * y)
                                         + (np.cos(0.13 * x) * np.cos(0.087 * y)) ** 2
                                         + np.cos(0.025 * x - 0.15 * y) * np.cos(0.07 * x - 0.02 * y))


def f_chem(c, rho_s=5.0, c_alpha=0.3, c_beta=0.7):
    return rho_s * (c - c_alpha) ** 2 * (c_beta - c) ** 2


def dfdc(c, rho_s=5.0, c_alpha=0.3, c_beta=0.7):
    a, b = c - c_alpha, c_beta - c
    return 2.0 * rho_s * a * b * (b - a)


def total_solute(solver):
    return solver.diagnostics()[1]


def total_free_energy(solver):
    return solver.diagnostics()[0]
