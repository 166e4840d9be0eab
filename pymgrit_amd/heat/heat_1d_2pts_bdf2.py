"""1-D heat equation, pairs of time points, variable-step BDF2.

Drop-in for ``pymgrit.heat.heat_1d_2pts_bdf2.Heat1DBDF2`` (reference src/pymgrit/heat/heat_1d_2pts_bdf2.py:17-138): with
``tau_i = t_i - t_{i-1}``, ``r_i = tau_i / tau_{i-1}`` the update at t_i solves
``(L + (1+2r)/(tau(1+r)) I) u_i = b_i - r^2/(tau(1+r)) u_{i-2} + (1+r)/tau u_{i-1}``; the first pair's second value comes
from one trapezoidal step.
"""
from pymgrit_amd.heat.heat_1d import thomas_toeplitz
from pymgrit_amd.heat.heat_1d_2pts import Heat1DTwoPoint
from pymgrit_amd.heat.vector_heat_1d_2pts import VectorHeat1D2Pts


class Heat1DBDF2(Heat1DTwoPoint):
    bdf_order = 2

    def _second_start_value(self, first):
        h = self.dtau / 2                                                              # heat_1d_2pts_bdf2.py:58-61
        load = (first - h * self._laplace(first)) + h * (self.rhs(self.x, self.t[0]) + self.rhs(self.x, self.t[0] + self.dtau))
        return thomas_toeplitz(h * self.fac, h * (2 * self.fac) + 1, load)

    def _bdf2(self, two_back, one_back, t_new, tau, tau_prev):
        r = tau / tau_prev
        c_two_back = (r ** 2) / (tau * (1 + r))
        c_one_back = (1 + r) / tau
        c_now = (1 + 2 * r) / (tau * (1 + r))
        load = self.rhs(self.x, t_new) - c_two_back * two_back + c_one_back * one_back
        return thomas_toeplitz(self.fac, 2 * self.fac + c_now, load)

    def step(self, u_start: VectorHeat1D2Pts, t_start: float, t_stop: float) -> VectorHeat1D2Pts:
        first, second, dtau = u_start.get_values()
        gap = t_stop - t_start - dtau
        first_new = self._bdf2(first, second, t_stop, gap, dtau)
        return self._result(u_start, first_new, self._bdf2(second, first_new, t_stop + dtau, dtau, gap))
