"""Dahlquist test problem u' = lambda*u, u(0) = 1: scalar state, four one-step methods.

API mirror of the reference's ``pymgrit.dahlquist.dahlquist`` (reference src/pymgrit/dahlquist/dahlquist.py:12-111).
A scalar has nothing to offload: this Application carries no device description and always runs on the plugin path
(BASELINE config 1 is the CPU plumbing case).
"""
import numpy as np

from pymgrit_amd.core.application import Application
from pymgrit_amd.core.vector import Vector


class VectorDahlquist(Vector):
    """One float per time point."""

    def __init__(self, value):
        super().__init__()
        self.value = value

    # value-semantics algebra
    def __add__(self, other):
        return type(self)(self.value + other.get_values())

    def __sub__(self, other):
        return type(self)(self.value - other.get_values())

    def __mul__(self, factor):
        return type(self)(self.value * factor)

    def norm(self):
        return np.linalg.norm(self.value)

    # construction
    def clone(self):
        return type(self)(self.value)

    def clone_zero(self):
        return type(self)(0)

    def clone_rand(self):
        return type(self)(np.random.rand(1)[0])

    # data access / exchange payload: the float itself
    def get_values(self):
        return self.value

    def set_values(self, value):
        self.value = value

    pack = get_values
    unpack = set_values


def _amplification(method, z):
    """u_new = amp(z) * u for BE / FE / TR, z = dt*lambda (dahlquist.py:99-105)."""
    if method == 'BE':
        return 1 / (1 - z)
    if method == 'FE':
        return 1 + z
    return (1 + z / 2) / (1 - z / 2)


class Dahlquist(Application):
    METHODS = ('BE', 'FE', 'TR', 'MR')

    def __init__(self, constant_lambda=-1, method='BE', *args, **kwargs):
        super().__init__(*args, **kwargs)
        if method not in self.METHODS:
            raise Exception('Unknown method. Choose BE (Backward Euler), FE (Forward Euler), TR (Trapezoidal rule) '
                            'or MR (implicit mid-point rule)')
        self.method = method
        self.lambda_value = constant_lambda
        self.vector_t_start = VectorDahlquist(1)
        self.vector_template = VectorDahlquist(0)

    def step(self, u_start: VectorDahlquist, t_start: float, t_stop: float) -> VectorDahlquist:
        dt = t_stop - t_start
        z = dt * self.lambda_value
        u = u_start.get_values()
        if self.method == 'MR':
            # implicit mid-point rule exactly as the reference writes it: the slope hard-codes lambda = -1
            slope = -1 / (1 - z / 2) * u
            return VectorDahlquist(u + dt * slope)
        return VectorDahlquist(_amplification(self.method, z) * u)
