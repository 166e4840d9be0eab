"""Behavioural contract of the plugin types (SURVEY section 8 row a9 and the per-application Vector classes), following
the cases of the reference's unit tests: tests/core/test_application.py, test_grid_transfer_copy.py,
test_simple_setup_problem.py, tests/heat/test_heat_1d.py, tests/advection/test_advection_1d.py,
tests/dahlquist/test_dahlquist.py (written against the same behaviours, not copied)."""
import warnings

import numpy as np
import pytest

from pymgrit_amd import (Advection1D, Application, Dahlquist, GridTransferCopy, Heat1D, Vector, simple_setup_problem)
from pymgrit_amd.advection.advection_1d import VectorAdvection1D
from pymgrit_amd.dahlquist.dahlquist import VectorDahlquist
from pymgrit_amd.heat.heat_1d import VectorHeat1D


class VecS(Vector):
    def __init__(self, v=0.0):
        super().__init__()
        self.v = v

    def __add__(self, o): return VecS(self.v + o.v)
    def __sub__(self, o): return VecS(self.v - o.v)
    def __mul__(self, f): return VecS(self.v * f)
    def norm(self): return abs(self.v)
    def clone(self): return VecS(self.v)
    def clone_zero(self): return VecS(0.0)
    def clone_rand(self): return VecS(0.5)
    def set_values(self, v): self.v = v
    def get_values(self): return self.v
    def pack(self): return self.v
    def unpack(self, v): self.v = v


class AppS(Application):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.vector_template = VecS()
        self.vector_t_start = VecS(1.0)

    def step(self, u_start, t_start, t_stop):
        return u_start * (1.0 - (t_stop - t_start))


def test_application_time_grid_forms():
    a = AppS(t_start=0, t_stop=1, nt=11)
    assert a.nt == 11 and a.t_start == 0 and a.t_end == 1 and np.array_equal(a.t, np.linspace(0, 1, 11))
    grid = np.array([0.0, 0.1, 0.4, 1.0])
    b = AppS(t_interval=grid)
    assert b.nt == 4 and b.t_start == 0.0 and b.t_end == 1.0 and b.t is grid
    for bad in (dict(t_start=0, t_stop=1), dict(t_start=0, nt=3), dict(t_stop=1, nt=3), dict()):
        with pytest.raises(Exception):
            AppS(**bad)
    with pytest.raises(Exception):
        AppS(t_interval=[0, 1, 2])


def test_application_missing_attributes():
    class NoTemplate(Application):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.vector_t_start = VecS()

        def step(self, u_start, t_start, t_stop):
            return u_start

    with pytest.raises(ValueError):
        NoTemplate(t_start=0, t_stop=1, nt=3)
    with pytest.raises(TypeError):
        Application(t_start=0, t_stop=1, nt=3)   # abstract


def test_vector_derived_operators():
    a, b = VecS(3.0), VecS(1.5)
    assert (2 * a).v == 6.0            # __rmul__
    a += b
    assert a.v == 4.5                  # __iadd__ rebinding through __add__
    a -= b
    assert a.v == 3.0
    a *= 2
    assert a.v == 6.0
    with pytest.raises(TypeError):
        Vector()


def test_grid_transfer_copy_clones():
    g = GridTransferCopy()
    v = VecS(2.0)
    r, i = g.restriction(v), g.interpolation(v)
    assert r.v == 2.0 and i.v == 2.0 and r is not v and i is not v
    assert g.device_transfer() == 0


def test_simple_setup_problem():
    base = AppS(t_start=0, t_stop=1, nt=17)
    levels = simple_setup_problem(problem=base, level=3, coarsening=2)
    assert levels[0] is base and [p.nt for p in levels] == [17, 9, 5]
    assert np.array_equal(levels[1].t, base.t[::2]) and np.array_equal(levels[2].t, base.t[::4])
    assert levels[2].t_start == 0 and levels[2].t_end == 1 and levels[1] is not levels[2]
    levels[1].vector_t_start.v = 7.0
    assert base.vector_t_start.v == 1.0                       # deep copies
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        simple_setup_problem(problem=AppS(t_start=0, t_stop=1, nt=5), level=2, coarsening=4)
        assert any("only one time point" in str(x.message) for x in w)


@pytest.mark.parametrize("cls,size", [(VectorHeat1D, 5), (VectorAdvection1D, 5)])
def test_array_vectors(cls, size):
    a, b = cls(size), cls(size)
    assert np.all(a.values == 0)
    a.set_values(np.ones(size))
    b.set_values(2 * np.ones(size))
    assert np.array_equal((a + b).get_values(), 3 * np.ones(size)) and np.array_equal((a - b).get_values(), -np.ones(size))
    assert np.array_equal((a * 3).get_values(), 3 * np.ones(size)) and np.array_equal((3 * a).get_values(), 3 * np.ones(size))
    v = cls(size)
    v.values = np.array([1, 2, 3, 4, 5])
    assert v.norm() == np.linalg.norm(np.array([1, 2, 3, 4, 5]))
    z, r, c = a.clone_zero(), a.clone_rand(), a.clone()
    assert isinstance(z, cls) and np.all(z.get_values() == 0) and len(r.get_values()) == size
    assert np.array_equal(c.get_values(), a.get_values()) and c is not a
    payload = b.pack()
    a.unpack(payload)
    assert np.array_equal(a.get_values(), b.get_values())


def test_vector_dahlquist():
    a, b = VectorDahlquist(5), VectorDahlquist(3)
    assert (a + b).get_values() == 8 and (a - b).get_values() == 2 and (a * 7).get_values() == 35
    assert VectorDahlquist(-4).norm() == 4 and a.clone_zero().get_values() == 0 and 0 <= a.clone_rand().get_values() <= 1
    a.set_values(9)
    assert a.get_values() == 9 and a.pack() == 9
    a.unpack(2.5)
    assert a.value == 2.5


def test_dahlquist_constructor_and_steps():
    for m in ("BE", "FE", "TR", "MR"):
        d = Dahlquist(method=m, t_start=0, t_stop=1, nt=11)
        assert d.method == m and d.vector_t_start.get_values() == 1 and d.vector_template.get_values() == 0
    with pytest.raises(Exception):
        Dahlquist(method="unknown", t_start=0, t_stop=1, nt=11)
    # reference tests/dahlquist/test_dahlquist.py:55-88 (literal expectations)
    for m, exp in (("BE", 0.9090909090909091), ("FE", 0.9), ("TR", 0.9047619047619047), ("MR", 0.9047619047619047)):
        d = Dahlquist(method=m, t_start=0, t_stop=1, nt=11)
        np.testing.assert_almost_equal(d.step(u_start=VectorDahlquist(1), t_start=0, t_stop=0.1).get_values(), exp)


def test_heat_1d_constructor_and_host_step():
    h = Heat1D(a=1, x_start=0, x_end=1, nx=11, t_start=0, t_stop=1, nt=11)
    assert h.x_start == 0 and h.x_end == 1 and h.nx == 9
    np.testing.assert_almost_equal(h.dx, 0.1)
    assert np.array_equal(h.x, np.linspace(0, 1, 11)[1:-1])
    assert isinstance(h.vector_template, VectorHeat1D) and np.array_equal(h.vector_t_start.get_values(), np.zeros(9))
    # reference tests/heat/test_heat_1d.py:31-42 (literal expectation), here through the host (Thomas) stepper
    h6 = Heat1D(a=1, init_cond=lambda x: 2 * x, x_start=0, x_end=1, nx=6, t_start=0, t_stop=1, nt=11)
    res = h6.step(u_start=h6.vector_t_start, t_start=0, t_stop=0.1)
    np.testing.assert_almost_equal(res.get_values(), np.array([0.28164, 0.51593599, 0.63660638, 0.53191933]))
    d = h6.device_stepper()
    assert d["kind"] == "heat1d" and d["n"] == 4 and d["forcing_space"].shape == (0, 4)


def test_heat_1d_forcing_forms():
    s, tau = (lambda x: -np.sin(np.pi * x)), (lambda t: np.sin(t) - np.pi ** 2 * np.cos(t))
    rhs = lambda x, t: s(x) * tau(t)
    a = Heat1D(a=1, x_start=0, x_end=1, nx=17, rhs=rhs, t_start=0, t_stop=2, nt=9)
    b = Heat1D(a=1, x_start=0, x_end=1, nx=17, rhs_separable=[(s, tau)], t_start=0, t_stop=2, nt=9)
    da, db = a.device_stepper(), b.device_stepper()
    for t in a.t:
        fa = da["forcing_space"][0] * da["forcing_time"][0](t)
        fb = db["forcing_space"][0] * db["forcing_time"][0](t)
        assert np.abs(fa - rhs(a.x, t)).max() <= 1e-13 * 10 and np.array_equal(fb, rhs(b.x, t))
    # a forcing that is not of the form s(x)*tau(t) is streamed as rows rhs(x, t_i)*dt_i (general forcing)
    g = Heat1D(a=1, x_start=0, x_end=1, nx=17, rhs=lambda x, t: np.sin(x * t), t_start=0, t_stop=2, nt=9).device_stepper()
    assert len(g["forcing_space"]) == 0 and g.get("forcing_rows") is not None


def test_advection_1d_constructor_and_host_step():
    adv = Advection1D(c=1, x_start=0, x_end=1, nx=11, t_start=0, t_stop=1, nt=11)
    assert adv.nx == 10 and adv.c == 1
    np.testing.assert_almost_equal(adv.dx, 0.1)
    assert np.array_equal(adv.x, np.linspace(0, 1, 11)[0:-1])
    assert np.array_equal(adv.vector_t_start.get_values(), np.exp(-np.linspace(0, 1, 11)[0:-1] ** 2))
    # reference tests/advection/test_advection_1d.py:32-44 (literal expectation)
    a6 = Advection1D(c=1, x_start=0, x_end=1, nx=6, t_start=0, t_stop=1, nt=11)
    res = a6.step(u_start=a6.vector_t_start, t_start=0, t_stop=0.1)
    np.testing.assert_almost_equal(res.get_values(), np.array([0.868043, 0.92987396, 0.87805385, 0.75780217, 0.604129]))


def test_mixed_hierarchy_runs_through_the_applications_own_steps(caplog):
    """the reference takes ANY Application per level (core/mgrit.py:79-99): a library Heat1D on the fine level over a user's own
    Application on the coarse one is not refused -- the whole hierarchy runs through the applications' step() methods (plugin
    path), and gives what the same hierarchy gives when both levels are the user's class"""
    import logging
    from pymgrit_amd import Mgrit

    class MyHeat(Application):
        """a user's restatement of backward Euler for the 1-D heat equation (dense solve), vectors of the library's type"""
        def __init__(self, nx, **kw):
            super().__init__(**kw)
            self.x = np.linspace(0, 1, nx)[1:-1]
            self.n, dx = self.x.size, 1.0 / (nx - 1)
            self.L = (np.diag(2 * np.ones(self.n)) - np.diag(np.ones(self.n - 1), 1) - np.diag(np.ones(self.n - 1), -1)) / dx ** 2
            self.vector_template = VectorHeat1D(self.n)
            self.vector_t_start = VectorHeat1D(self.n)
            self.vector_t_start.set_values(np.sin(np.pi * self.x))

        def step(self, u_start, t_start, t_stop):
            out = VectorHeat1D(self.n)
            out.set_values(np.linalg.solve(np.eye(self.n) + (t_stop - t_start) * self.L, u_start.get_values()))
            return out

    t0 = np.linspace(0, 1, 33)
    lib = Heat1D(x_start=0, x_end=1, nx=17, a=1, init_cond=lambda x: np.sin(np.pi * x), t_interval=t0)
    with caplog.at_level(logging.WARNING):
        mixed = Mgrit([lib, MyHeat(17, t_interval=t0[::4])], tol=1e-9, max_iter=8, logging_lvl=30)
    assert type(mixed.backend).__name__ == "PluginBackend"
    assert any("no device description" in r.getMessage() for r in caplog.records)
    conv = mixed.solve()["conv"]
    own = Mgrit([MyHeat(17, t_interval=t0), MyHeat(17, t_interval=t0[::4])], tol=1e-9, max_iter=8, logging_lvl=30).solve()["conv"]
    assert len(conv) == len(own) and np.max(np.abs(conv - own) / own) <= 1e-6
    assert np.all(np.diff(conv) < 0)      # (later cycles give an exact zero: a two-level hierarchy is exact after N_c / 2 iterations)
