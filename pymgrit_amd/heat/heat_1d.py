"""1-D heat equation  u_t - a u_xx = b(x,t)  with homogeneous Dirichlet BCs: Vector + Application.

Drop-in for the reference's ``pymgrit.heat.heat_1d`` (reference src/pymgrit/heat/heat_1d.py:14-217): same constructor,
same grid (``nx-2`` interior points, ``dx = x[1]-x[0]``), same backward-Euler step
``u_i = (I + dt L)^{-1} (u_{i-1} + dt b(x, t_i))``, ``L = (a/dx^2) tridiag(-1, 2, -1)``.

On MI355X the step runs as a HIP kernel described by ``device_stepper()``. The kernel needs the forcing in
declarative form ``b(x,t) = sum_k s_k(x) tau_k(t)``: pass ``rhs_separable=[(s_fn, tau_fn), ...]`` (exact), or let the
constructor detect a rank-one ``rhs`` numerically. Any other forcing -- the reference accepts every callable ``rhs(x, t)`` --
runs on the device from precomputed rows ``rhs(x, t_i)*dt_i`` (one per time point, streamed from HBM).
"""
import numpy as np

from pymgrit_amd.core.application import Application
from pymgrit_amd.core.vector import Vector


class VectorHeat1D(Vector):
    """State of one time point: ``size`` interior values (reference heat_1d.py:14-128)."""

    def __init__(self, size):
        super().__init__()
        self.size = size
        self.values = np.zeros(size)

    def _new(self, values):
        out = VectorHeat1D(self.size)
        out.set_values(values)
        return out

    def __add__(self, other):
        return self._new(self.get_values() + other.get_values())

    def __sub__(self, other):
        return self._new(self.get_values() - other.get_values())

    def __mul__(self, other):
        return self._new(self.get_values() * other)

    def norm(self):
        return np.linalg.norm(self.values)

    def clone(self):
        return self._new(self.get_values())

    def clone_zero(self):
        return VectorHeat1D(self.size)

    def clone_rand(self):
        return self._new(np.random.rand(self.size))

    def set_values(self, values):
        self.values = values

    def get_values(self):
        return self.values

    def pack(self):
        return self.values

    def unpack(self, values):
        self.values = values


def _zero_rhs(x, t):
    return x * 0


def _zero_init(x):
    return x * 0


def thomas_toeplitz(beta, diag, d):
    """Solve tridiag(-beta, diag, -beta) x = d (the matrices of heat_1d.py:177-217 are of this form)."""
    n = d.shape[0]
    cp, dp = np.empty(n), np.empty(n)
    piv = diag
    cp[0], dp[0] = -beta / piv, d[0] / piv
    for j in range(1, n):
        piv = diag + beta * cp[j - 1]
        cp[j] = -beta / piv
        dp[j] = (d[j] + beta * dp[j - 1]) / piv
    out = np.empty(n)
    out[-1] = dp[-1]
    for j in range(n - 2, -1, -1):
        out[j] = dp[j] - cp[j] * out[j + 1]
    return out


class NotSeparable(Exception):
    """rhs(x, t) is not of the rank-one form s(x)*tau(t)"""


def detect_separable(rhs, x, t, who='Heat1D'):
    """Find (s, tau) with rhs(x,t) = s(x)*tau(t) to rounding, or K=0 for a zero forcing; raises NotSeparable otherwise.
    The full spatial shape of the forcing is compared at 16 times spread over the grid, and at EVERY time point on a sub-grid
    of up to 9 probe points (a forcing whose shape deviates only between the 16 probe times must not pass: the device would
    run with a wrong forcing, silently); tau itself is evaluated from rhs at a pivot point for every time point, so its time
    dependence is exact whatever it is."""
    nx = x.shape[0]
    probes = np.unique(t[np.unique(np.linspace(0, len(t) - 1, 16).astype(int))])
    samples = [np.asarray(rhs(x, float(tp)), dtype=np.float64) * np.ones(nx) for tp in probes]
    norms = [np.max(np.abs(s)) for s in samples]
    if max(norms) == 0.0:
        return [], []
    s_vec = samples[int(np.argmax(norms))]
    piv = int(np.argmax(np.abs(s_vec)))
    xp = x[piv:piv + 1]

    def tau_fn(tt, _xp=xp, _den=s_vec[piv]):
        return float(np.asarray(rhs(_xp, tt), dtype=np.float64).ravel()[0]) / _den
    refusal = NotSeparable(f'{who}: rhs(x,t) is not of the separable form s(x)*tau(t); pass rhs_separable=[(s_fn, '
                           'tau_fn), ...] to run on the MI355X engine')
    for tp, smp in zip(probes, samples):
        if np.max(np.abs(smp - s_vec * tau_fn(float(tp)))) > 1e-12 * max(norms):
            raise refusal
    sub = np.unique(np.concatenate((np.linspace(0, nx - 1, 8).astype(int), [piv])))
    xs, ss = x[sub], s_vec[sub]
    k_piv = int(np.nonzero(sub == piv)[0][0])
    scale = 0.0
    rows = []
    for tp in t:      # one rhs call per time point on the sub-grid: the pivot's value is tau(t) * s[piv]
        row = np.asarray(rhs(xs, float(tp)), dtype=np.float64) * np.ones(xs.shape[0])
        rows.append(row)
        scale = max(scale, float(np.max(np.abs(row))))
    for row in rows:
        if np.max(np.abs(row - ss * (row[k_piv] / s_vec[piv]))) > 1e-12 * max(scale, max(norms)):
            raise refusal
    return [s_vec], [tau_fn]


def separable_rhs(terms):
    """rhs(x, t) = sum_k s_k(x) * tau_k(t) from a list of (s_fn, tau_fn) pairs."""
    def rhs(x, t, _terms=tuple(terms)):
        total = _terms[0][0](x) * _terms[0][1](t)
        for s_fn, tau_fn in _terms[1:]:
            total = total + s_fn(x) * tau_fn(t)
        return total
    return rhs


class Heat1D(Application):
    def __init__(self, x_start, x_end, nx, a, init_cond=_zero_init, rhs=_zero_rhs, rhs_separable=None, *args, **kwargs):
        """
        :param x_start, x_end: spatial interval; ``nx`` grid points including the two boundary points
        :param a: thermal conductivity
        :param init_cond: callable u(x, 0)
        :param rhs: callable b(x, t) (reference signature)
        :param rhs_separable: optional list of ``(s_fn(x), tau_fn(t))`` pairs with b = sum_k s_k(x)*tau_k(t); when
               given it defines ``rhs`` and is used verbatim by the device stepper
        """
        super().__init__(*args, **kwargs)
        self.x_start, self.x_end = x_start, x_end
        self.x = np.linspace(self.x_start, self.x_end, nx)[1:-1]
        self.nx = nx - 2
        self.dx = self.x[1] - self.x[0]
        self.a = a
        self.fac = self.a / self.dx ** 2
        self._separable = list(rhs_separable) if rhs_separable is not None else None
        if self._separable is not None:
            rhs = separable_rhs(self._separable)
        self.rhs = rhs
        self._rhs_declared = rhs      # a caller that replaces .rhs afterwards gets the replaced forcing on the device too
        self.init_cond = init_cond
        self.vector_template = VectorHeat1D(self.nx)
        self.vector_t_start = VectorHeat1D(self.nx)
        self.vector_t_start.set_values(self.init_cond(self.x))
        self._device_desc = None

    # ---- host stepper (plugin path / user inspection): the same linear solve by the Thomas algorithm ----------
    def step(self, u_start: VectorHeat1D, t_start: float, t_stop: float) -> VectorHeat1D:
        dt = t_stop - t_start
        d = u_start.get_values() + self.rhs(self.x, t_stop) * dt
        ret = VectorHeat1D(self.nx)
        ret.set_values(thomas_toeplitz(dt * self.fac, dt * (2 * self.fac) + 1, d))
        return ret

    # ---- device description ------------------------------------------------------------------------------------
    def _detect_separable(self):
        return detect_separable(self.rhs, self.x, self.t, type(self).__name__)

    def device_stepper(self):
        """Declarative Phi for libmgrit_hip (include/mgrit_hip.h: mgrit_hip_level_heat1d)."""
        if self._device_desc is None or self._device_desc.get("_rhs") is not self.rhs:
            rows = None
            if self._separable is not None and self.rhs is self._rhs_declared:
                space = [np.asarray(s_fn(self.x), dtype=np.float64) * np.ones(self.nx) for s_fn, _ in self._separable]
                time_fns = [tau_fn for _, tau_fn in self._separable]
            else:
                try:
                    space, time_fns = self._detect_separable()
                except NotSeparable:
                    # any rhs(x, t) of the reference (heat/heat_1d.py:138,213): the engine streams one precomputed row
                    # rhs(x, t_i)*dt_i per time point from HBM instead of evaluating s(x)*tau(t) in registers
                    space, time_fns, rows = [], [], self.forcing_row
            self._device_desc = {"kind": "heat1d", "n": self.nx, "fac": self.fac,
                                 "forcing_space": np.array(space, dtype=np.float64).reshape(len(space), self.nx),
                                 "forcing_time": time_fns, "forcing_rows": rows, "_rhs": self.rhs}
        return self._device_desc

    def forcing_row(self, t_start, t_stop):
        """rhs(x, t_stop) * (t_stop - t_start): the term ``step`` adds to u_start (reference heat_1d.py:213), as one row"""
        return np.asarray(self.rhs(self.x, t_stop) * (t_stop - t_start), dtype=np.float64) * np.ones(self.nx)
