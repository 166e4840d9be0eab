"""Local convergence criteria conv_crit 2 (residual) / 3 (jump) on one rank (reference mgrit.py:434-455,627-635) against
fixtures generated from the reference (tests/golden/local_conv.json, make_golden.py --only-local-conv): the host path here,
the HIP path under -m gpu; on several ranks (threads with rendezvous sends, tests/mock_comm.py) against the reference's own
multi-rank runs (tests/golden/local_conv_ranks.json)."""
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
    assert np.all(np.abs(conv - np.array(g["conv"])) <= 1e-9 * np.array(g["conv"]) + 2e-11)
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


# ---- several ranks: the reference's drain protocol (fixtures from its true multi-rank path, make_golden.py) ---------------
RANKS = cases.load_json("local_conv_ranks.json")


def _ranks_problem(name):
    if name.startswith("dahlquist"):
        return (lambda: simple_setup_problem(Dahlquist(t_start=0, t_stop=5, nt=101), level=3, coarsening=2)), dict(tol=1e-10, conv_crit=2)
    table = {
        "heat_crit2_V": ([65, 17, 5], dict(tol=1e-7, max_iter=12, conv_crit=2)),
        "heat_crit3_V": ([65, 17, 5], dict(tol=1e-7, max_iter=12, conv_crit=3)),
        "heat_crit2_F_nonested": ([65, 17, 5], dict(tol=1e-7, max_iter=12, conv_crit=2, cycle_type='F', nested_iteration=False)),
        "heat_crit2_maxiter": ([65, 17, 5], dict(tol=1e-14, max_iter=3, conv_crit=2)),
        "heat_crit2_2lvl_cf2": ([65, 9], dict(tol=1e-7, max_iter=12, conv_crit=2, cf_iter=2)),
    }
    nts, opts = table[name]

    def make():
        prob = [Heat1D(x_start=0, x_end=1, nx=33, a=1, init_cond=cases.init_cond, rhs_separable=[(cases.rhs_space, cases.rhs_time)],
                       t_start=0, t_stop=2, nt=nt) for nt in nts]
        for p in prob:
            p.device_stepper = lambda: None
        return prob
    return make, opts


@pytest.mark.parametrize("key", sorted(RANKS))
def test_local_criterion_on_several_ranks_matches_the_reference(key):
    """ranks leave the solve loop in different iterations (mgrit.py:434-455,648-691): per rank the residual history (its
    length = the iteration in which the rank left) and sampled solution values of the reference's own multi-rank run"""
    from mock_comm import run_ranks
    name, size = key.rsplit("_P", 1)
    make, opts = _ranks_problem(name)

    def target(comm):
        mg = Mgrit(make(), comm_time=comm, logging_lvl=30, **opts)
        conv = mg.solve()["conv"]
        own = [int(i) for i in mg.index_local[0]]
        return conv, own, [np.asarray(mg.u[0][i].get_values(), dtype=np.float64).ravel() for i in own]
    res = run_ranks(int(size), target, timeout=30)
    for (conv, own, u), g in zip(res, RANKS[key]):
        assert len(conv) == len(g["conv"]), (key, [len(r[0]) for r in res], [len(x["conv"]) for x in RANKS[key]])
        assert np.all(np.abs(conv - np.array(g["conv"])) <= 1e-9 * np.abs(np.array(g["conv"])) + 2e-11)
        assert len(own) == g["n_owned"]
        for j, vals in g["u"].items():
            assert np.allclose(u[int(j)], vals, rtol=1e-9, atol=1e-11)


def _check_against_reference(res, key):
    for (conv, u), g in zip(res, RANKS[key]):
        assert len(conv) == len(g["conv"])
        assert np.all(np.abs(conv - np.array(g["conv"])) <= 1e-9 * np.abs(np.array(g["conv"])) + 2e-11)
        assert u.shape[0] == g["n_owned"]
        for j, vals in g["u"].items():
            assert np.allclose(u[int(j)], vals, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("key", ["heat_crit2_V_P3", "heat_crit3_V_P2"])
def test_local_criterion_over_torch_distributed(key):
    """the same protocol over real process groups (gloo): per-link communicators, pickled verdicts, staged sends"""
    from test_distributed import launch
    name, size = key.rsplit("_P", 1)
    _check_against_reference(launch(int(size), "lc:" + name, per_rank=True, timeout=120), key)


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["heat_crit2_V_P3", "heat_crit2_F_nonested_P2", "heat_crit3_V_P2"])
def test_local_criterion_on_several_ranks_hip_path(key):
    """device rows as farewell messages (ranks share the one GPU of the test box, gloo transport)"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from test_distributed import launch
    name, size = key.rsplit("_P", 1)
    _check_against_reference(launch(int(size), "lc:" + name, mode="hip", backend="gloo", per_rank=True, timeout=300), key)
