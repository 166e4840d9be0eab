"""Local convergence criteria conv_crit 2 (residual) / 3 (jump) on one rank (reference mgrit.py:434-455,627-635) against
fixtures generated from the reference (tests/golden/local_conv.json, make_golden.py --only-local-conv): the host path here,
the HIP path under -m gpu. On several ranks the criteria raise (their drain protocol is out of scope)."""
import numpy as np
import pytest

import cases
from pymgrit_amd import Dahlquist, Heat1D, Mgrit, simple_setup_problem

GOLD = cases.load_json("local_conv.json")


def heat_levels(host_only):
    prob = [Heat1D(x_start=0, x_end=1, nx=33, a=1, init_cond=cases.init_cond, rhs_separable=[(cases.rhs_space, cases.rhs_time)],
                   t_start=0, t_stop=2, nt=nt) for nt in (65, 17, 5)]
    if host_only:
        for p in prob:
            p.device_stepper = lambda: None
    return prob


def check(mg, g):
    conv = mg.solve()["conv"]
    assert len(conv) == len(g["conv"])
    assert np.all(np.abs(conv - np.array(g["conv"])) <= 1e-8 * np.array(g["conv"]) + 2e-11)
    for i, vals in g["samples"].items():
        assert np.allclose(np.asarray(mg.u[0][int(i)].get_values()).ravel(), vals, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("crit", [2, 3])
def test_dahlquist_local_criterion(crit):
    d = simple_setup_problem(Dahlquist(t_start=0, t_stop=5, nt=101), level=2, coarsening=2)
    mg = Mgrit(d, tol=1e-10, conv_crit=crit, logging_lvl=30)
    assert mg.global_conv_crit is False
    check(mg, GOLD[f"dahlquist_crit{crit}"])


@pytest.mark.parametrize("crit", [2, 3])
def test_heat_local_criterion_host_path(crit):
    check(Mgrit(heat_levels(True), tol=1e-7, max_iter=12, conv_crit=crit, logging_lvl=30), GOLD[f"heat_nx33_crit{crit}"])
    check(Mgrit(heat_levels(True), tol=1e-14, max_iter=3, conv_crit=crit, logging_lvl=30), GOLD[f"heat_nx33_crit{crit}_maxiter"])


def test_local_criterion_stops_later_than_the_global_one():
    """every point below tol on its own is stricter than the norm over all points being below tol"""
    n_global = len(Mgrit(heat_levels(True), tol=1e-7, max_iter=12, conv_crit=0, logging_lvl=30).solve()["conv"])
    n_local = len(Mgrit(heat_levels(True), tol=1e-7, max_iter=12, conv_crit=2, logging_lvl=30).solve()["conv"])
    assert n_local >= n_global


@pytest.mark.gpu
@pytest.mark.parametrize("crit", [2, 3])
def test_heat_local_criterion_hip_path(crit):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    mg = Mgrit(heat_levels(False), tol=1e-7, max_iter=12, conv_crit=crit, logging_lvl=30)
    assert mg.backend.name == "hip"
    check(mg, GOLD[f"heat_nx33_crit{crit}"])


def test_local_criterion_on_several_ranks_is_refused():
    class TwoRanks:   # a communicator stand-in: the constructor must refuse before any exchange happens
        def Get_rank(self): return 0
        def Get_size(self): return 2
        def barrier(self): return None
        def allgather_object(self, obj): raise AssertionError("no communication expected")
        def exchange(self, send=None, recv=None): raise AssertionError("no communication expected")
    d = simple_setup_problem(Dahlquist(t_start=0, t_stop=5, nt=101), level=2, coarsening=2)
    with pytest.raises(Exception, match="one rank only"):
        Mgrit(d, conv_crit=2, comm_time=TwoRanks(), logging_lvl=30)
