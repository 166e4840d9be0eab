"""Dahlquist test problem u' = lambda*u, u(0) = 1 (scalar ODE). Drop-in for the reference's
``pymgrit.dahlquist.dahlquist`` (reference src/pymgrit/dahlquist/dahlquist.py:12-111). A scalar state has nothing to
offload: this Application carries no device description and runs on the plugin path (BASELINE config 1: CPU plumbing)."""
import numpy as np

from pymgrit_amd.core.application import Application
from pymgrit_amd.core.vector import Vector


class VectorDahlquist(Vector):
    def __init__(self, value):
        super().__init__()
        self.value = value

    def __add__(self, other):
        return VectorDahlquist(self.get_values() + other.get_values())

    def __sub__(self, other):
        return VectorDahlquist(self.get_values() - other.get_values())

    def __mul__(self, other):
        return VectorDahlquist(self.get_values() * other)

    def norm(self):
        return np.linalg.norm(self.value)

    def clone(self):
        return VectorDahlquist(self.value)

    def clone_zero(self):
        return VectorDahlquist(0)

    def clone_rand(self):
        return VectorDahlquist(np.random.rand(1)[0])

    def set_values(self, value):
        self.value = value

    def get_values(self):
        return self.value

    def pack(self):
        return self.value

    def unpack(self, value):
        self.value = value


class Dahlquist(Application):
    METHODS = ('BE', 'FE', 'TR', 'MR')

    def __init__(self, constant_lambda=-1, method='BE', *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.vector_template = VectorDahlquist(0)
        self.vector_t_start = VectorDahlquist(1)
        self.lambda_value = constant_lambda
        if method not in self.METHODS:
            raise Exception('Unknown method. Choose BE (Backward Euler), FE (Forward Euler), TR (Trapezoidal rule) '
                            'or MR (implicit mid-point rule)')
        self.method = method

    def step(self, u_start: VectorDahlquist, t_start: float, t_stop: float) -> VectorDahlquist:
        """BE / FE / TR closed forms; MR as written in the reference (its k1 hard-codes lambda = -1)."""
        u = u_start.get_values()
        z = (t_stop - t_start) * self.lambda_value
        if self.method == 'BE':
            new = 1 / (1 - z) * u
        elif self.method == 'FE':
            new = (1 + z) * u
        elif self.method == 'TR':
            new = (1 + z / 2) / (1 - z / 2) * u
        else:
            k1 = -1 / (1 - z / 2) * u
            new = u + (t_stop - t_start) * k1
        return VectorDahlquist(new)
