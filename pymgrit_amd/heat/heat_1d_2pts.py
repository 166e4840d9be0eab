"""Shared part of the two-point (pair-of-time-points) 1-D heat applications: grid, forcing forms, the declarative
device description. One Phi advances a pair ``(u(t), u(t + dtau))`` to ``(u(t'), u(t' + dtau))`` with two tridiagonal
Toeplitz solves (reference src/pymgrit/heat/heat_1d_2pts_bdf1.py:84-117, heat_1d_2pts_bdf2.py:82-138); on MI355X both
solves run inside one workgroup (``mgrit_hip_level_heat1d_2pts``, include/mgrit_hip.h).
"""
import numpy as np

from pymgrit_amd.core.application import Application
from pymgrit_amd.heat.heat_1d import _zero_init, _zero_rhs, detect_separable, separable_rhs
from pymgrit_amd.heat.vector_heat_1d_2pts import VectorHeat1D2Pts


class Heat1DTwoPoint(Application):
    """``u_t - a u_xx = b(x,t)`` on ``nx-2`` interior points, homogeneous Dirichlet BCs; states are pairs spaced dtau."""
    bdf_order = None  # set by the subclasses: 1 or 2

    def __init__(self, x_start, x_end, nx, dtau, a, init_cond=_zero_init, rhs=_zero_rhs, rhs_separable=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.x_start, self.x_end = x_start, x_end
        self.x = np.linspace(self.x_start, self.x_end, nx)[1:-1]
        self.nx = nx - 2
        self.dx = self.x[1] - self.x[0]
        self.a = a
        self.dtau = dtau
        self.fac = self.a / self.dx ** 2
        self._separable = list(rhs_separable) if rhs_separable is not None else None
        self.rhs = separable_rhs(self._separable) if self._separable is not None else rhs
        self.init_cond = init_cond
        self.vector_template = VectorHeat1D2Pts(self.nx, dtau)
        self.vector_t_start = VectorHeat1D2Pts(self.nx, dtau)
        first = self.init_cond(self.x)
        self.vector_t_start.set_values(first_time_point=first, second_time_point=self._second_start_value(first), dtau=dtau)
        self._device_desc = None

    def _second_start_value(self, first):
        raise NotImplementedError

    def _laplace(self, v):
        """L v with L = (a/dx^2) tridiag(-1, 2, -1)"""
        out = 2.0 * v
        out[1:] -= v[:-1]
        out[:-1] -= v[1:]
        return self.fac * out

    def _result(self, like, first, second):
        ret = VectorHeat1D2Pts(like.size, like.dtau)
        ret.set_values(first_time_point=first, second_time_point=second, dtau=self.dtau)
        return ret

    def device_stepper(self):
        """Declarative Phi for libmgrit_hip: forcing as sum_k s_k(x) tau_k(t), evaluated by the engine at t_i and t_i + dtau."""
        if self._device_desc is None:
            if self._separable is not None:
                space = [np.asarray(s_fn(self.x), dtype=np.float64) * np.ones(self.nx) for s_fn, _ in self._separable]
                time_fns = [tau_fn for _, tau_fn in self._separable]
            else:
                space, time_fns = detect_separable(self.rhs, self.x, self.t, type(self).__name__)
            self._device_desc = {"kind": "heat1d_2pts", "order": self.bdf_order, "n": self.nx, "fac": self.fac,
                                 "dtau": float(self.dtau),
                                 "forcing_space": np.array(space, dtype=np.float64).reshape(len(space), self.nx),
                                 "forcing_time": time_fns}
        return self._device_desc
