"""State of a *pair* of consecutive time points (t, t + dtau) of the 1-D heat problem.

Drop-in for ``pymgrit.heat.vector_heat_1d_2pts.VectorHeat1D2Pts`` (reference src/pymgrit/heat/vector_heat_1d_2pts.py:9-140):
``get_values()`` returns ``(first, second, dtau)``, ``set_values(first, second, dtau)``, ``pack()`` is the 2 x size array,
the norm is the 2-norm of both halves together. In the HIP engine such a state is one slab row ``[first | second]``.
"""
import numpy as np

from pymgrit_amd.core.vector import Vector


class VectorHeat1D2Pts(Vector):
    def __init__(self, size, dtau):
        super().__init__()
        self.size = size
        self.dtau = dtau
        self.values_first_time_point = np.zeros(size)
        self.values_second_time_point = np.zeros(size)

    def _like(self, first, second):
        out = VectorHeat1D2Pts(self.size, self.dtau)
        out.set_values(first, second, self.dtau)
        return out

    def __add__(self, other):
        a1, a2, _ = self.get_values()
        b1, b2, _ = other.get_values()
        return self._like(a1 + b1, a2 + b2)

    def __sub__(self, other):
        a1, a2, _ = self.get_values()
        b1, b2, _ = other.get_values()
        return self._like(a1 - b1, a2 - b2)

    def __mul__(self, other):
        a1, a2, _ = self.get_values()
        return self._like(a1 * other, a2 * other)

    def norm(self):
        return np.linalg.norm(np.concatenate((self.values_first_time_point, self.values_second_time_point)))

    def clone(self):
        return self._like(self.values_first_time_point, self.values_second_time_point)

    def clone_zero(self):
        return VectorHeat1D2Pts(self.size, self.dtau)

    def clone_rand(self):
        return self._like(np.random.rand(self.size), np.random.rand(self.size))

    def get_values(self):
        return self.values_first_time_point, self.values_second_time_point, self.dtau

    def set_values(self, first_time_point, second_time_point, dtau):
        self.values_first_time_point = first_time_point
        self.values_second_time_point = second_time_point
        self.dtau = dtau

    def pack(self):
        return np.array([self.values_first_time_point, self.values_second_time_point])

    def unpack(self, values):
        self.values_first_time_point, self.values_second_time_point = values[0], values[1]
