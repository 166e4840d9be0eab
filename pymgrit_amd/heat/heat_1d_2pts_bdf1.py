"""1-D heat equation, pairs of time points, backward Euler inside and between pairs.

Drop-in for ``pymgrit.heat.heat_1d_2pts_bdf1.Heat1DBDF1`` (reference src/pymgrit/heat/heat_1d_2pts_bdf1.py:17-117):
Phi takes one BDF1 step from ``t_start + dtau`` to ``t_stop`` and one from ``t_stop`` to ``t_stop + dtau``.
"""
from pymgrit_amd.heat.heat_1d import thomas_toeplitz
from pymgrit_amd.heat.heat_1d_2pts import Heat1DTwoPoint
from pymgrit_amd.heat.vector_heat_1d_2pts import VectorHeat1D2Pts


class Heat1DBDF1(Heat1DTwoPoint):
    bdf_order = 1

    def _be(self, u, t_new, dt):
        """(I + dt L)^{-1} (u + dt b(x, t_new))"""
        return thomas_toeplitz(dt * self.fac, dt * (2 * self.fac) + 1, u + self.rhs(self.x, t_new) * dt)

    def _second_start_value(self, first):
        return self._be(first, self.t[0] + self.dtau, self.dtau)   # heat_1d_2pts_bdf1.py:56-58

    def step(self, u_start: VectorHeat1D2Pts, t_start: float, t_stop: float) -> VectorHeat1D2Pts:
        _, second, dtau = u_start.get_values()
        first_new = self._be(second, t_stop, t_stop - t_start - dtau)
        return self._result(u_start, first_new, self._be(first_new, t_stop + dtau, dtau))
