"""GPU parity tests for the two-point BDF steppers (reference heat/heat_1d_2pts_bdf1.py,
heat_1d_2pts_bdf2.py, vector_heat_1d_2pts.py): the HIP kernels through the C ABI against the oracle's spec variant --
states and per-point norms bit-exact -- and the solver against the fixtures generated from the reference
(tests/golden/bdf.json: residual history within 1e-8 rel, the reference being SuperLU)."""
import numpy as np
import pytest

import cases
from test_hip_parity import _need_gpu, assert_state_equal, randomize

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

GOLD = cases.load_json("bdf.json")


def make_pair(oracle, nx, n_pairs, orders, coarsening, forcing, **opts):
    from pymgrit_amd import Mgrit
    opts.setdefault("nested_iteration", False)
    prob = cases.bdf_levels(nx, n_pairs, orders, coarsening, forcing)
    mg = Mgrit(prob, logging_lvl=30, **opts)
    assert mg.backend.name == "hip"
    op = oracle.OracleProblem([cases.bdf_level_spec(p) for p in prob], variant=1, **opts)
    return mg, op


SHAPES = [  # nx (incl. boundary points), pairs, BDF order per level, coarsening, forcing
    (11, 33, [2, 1, 1], 2, "one"), (35, 33, [1, 1, 1], 2, "one"), (35, 33, [2, 2, 2], 2, "two"), (35, 17, [2, 1], 4, "zero"),
    (1026, 17, [2, 1, 1], 2, "one"), (1027, 17, [2, 2, 1], 2, "two"), (3000, 9, [1, 1], 2, "one"),
    (4098, 9, [2, 1, 1], 2, "one"),   # 4096 values per half of the pair: the largest state one workgroup holds
    # wider pairs (round 4): every half-solve as three launches over rows in HBM (csrc/mgrit_hip_wide.inc, wide2_*)
    (4099, 9, [2, 1], 2, "one"), (5002, 9, [2, 1, 1], 2, "one"), (5003, 9, [1, 1], 2, "two"), (20002, 5, [2, 2], 2, "one"),
]


@pytest.mark.parametrize("nx,n_pairs,orders,coarsening,forcing", SHAPES, ids=[f"nx{s[0]}-{''.join(map(str, s[2]))}-{s[4]}" for s in SHAPES])
def test_two_point_sweeps_bit_exact(oracle, nx, n_pairs, orders, coarsening, forcing):
    _need_gpu()
    mg, op = make_pair(oracle, nx, n_pairs, orders, coarsening, forcing)
    randomize(mg, op, seed=nx)
    for lvl in range(mg.lvl_max - 1):
        mg.f_relax(lvl); op.f_relax(lvl)
        assert_state_equal(mg, op)
        mg.c_relax(lvl); op.c_relax(lvl)
        assert_state_equal(mg, op)
        mg.fas_residual(lvl); op.fas_residual(lvl)
        assert_state_equal(mg, op)
    mg.forward_solve(mg.lvl_max - 1); op.forward_solve(mg.lvl_max - 1)
    assert_state_equal(mg, op)
    for lvl in range(mg.lvl_max - 2, -1, -1):
        mg.error_correction(lvl); op.error_correction(lvl)
        assert_state_equal(mg, op)
    got, ref = np.array(mg.compute_residual()), op.residual_norms()
    assert np.array_equal(got, ref), np.abs(got - ref).max()


def test_two_point_weighted_c_relax_and_non_uniform_pairs(oracle):
    """weight_c != 1 and a time grid with several distinct pair distances (several coefficient sets per level)"""
    _need_gpu()
    from pymgrit_amd import Mgrit
    from pymgrit_amd.heat.heat_1d_2pts_bdf1 import Heat1DBDF1
    from pymgrit_amd.heat.heat_1d_2pts_bdf2 import Heat1DBDF2
    t = np.cumsum(np.concatenate(([0.0], np.tile([0.125, 0.25, 0.1875], 11))))[:33]
    kw = dict(x_start=0, x_end=1, nx=67, a=1, dtau=0.0625, init_cond=cases.init_cond, rhs_separable=cases.BDF_FORCING["one"])
    prob = [Heat1DBDF2(t_interval=t, **kw), Heat1DBDF1(t_interval=t[::2], **kw), Heat1DBDF1(t_interval=t[::4], **kw)]
    mg = Mgrit(prob, logging_lvl=30, nested_iteration=False, weight_c=1.3)
    op = oracle.OracleProblem([cases.bdf_level_spec(p) for p in prob], variant=1, nested_iteration=False, weight_c=1.3)
    randomize(mg, op, seed=4)
    for lvl in (0, 1):
        mg.f_relax(lvl); op.f_relax(lvl)
        mg.c_relax(lvl); op.c_relax(lvl)
        assert_state_equal(mg, op)
        mg.fas_residual(lvl); op.fas_residual(lvl)
        assert_state_equal(mg, op)
    mg.forward_solve(2); op.forward_solve(2)
    assert_state_equal(mg, op)


@pytest.mark.parametrize("name", sorted(cases.BDF_CASES))
def test_two_point_solve_matches_oracle_and_reference(oracle, name):
    _need_gpu()
    from pymgrit_amd import Mgrit
    c, g = cases.BDF_CASES[name], GOLD["solve"][name]
    prob = cases.bdf_levels(c["nx"], c["n_pairs"], c["orders"], c["coarsening"], c["forcing"])
    mg = Mgrit(prob, logging_lvl=30, **c["kw"])
    assert mg.backend.name == "hip"
    conv = mg.solve()["conv"]
    op = oracle.OracleProblem([cases.bdf_level_spec(p) for p in prob], variant=1, **c["kw"])
    ref = op.solve()
    assert len(conv) == len(ref) == len(g["conv"])
    assert np.all(np.abs(conv - ref) <= 1e-10 * np.abs(ref)), (conv, ref)      # north_star tolerance; in practice identical
    assert np.array_equal(mg.backend.natural("u", 0), op.state("u", 0))
    assert np.all(np.abs(conv - np.array(g["conv"])) <= 1e-8 * np.array(g["conv"]) + 1e-13)
    for i, (a, b) in g["samples"].items():
        f, s, dtau = mg.u[0][int(i)].get_values()
        assert dtau == prob[0].dtau
        assert np.allclose(f, a, rtol=1e-9, atol=1e-11) and np.allclose(s, b, rtol=1e-9, atol=1e-11)


def test_two_point_limits_fail_loudly():
    _need_gpu()
    from pymgrit_amd import Mgrit
    from pymgrit_amd.core.hip_lib import MgritHipError
    prob = cases.bdf_levels(65540, 3, [1, 1], 2, "zero")      # 65538 values per half: beyond the wide path's 64 groups
    with pytest.raises(MgritHipError, match="two-point"):
        Mgrit(prob, logging_lvl=30)


@pytest.mark.parametrize("kw", [dict(), dict(conv_crit=1), dict(weight_c=1.3, cycle_type='F'), dict(nested_iteration=True)],
                         ids=["residual", "jump", "weighted_F", "nested"])
def test_two_point_wide_solve_matches_oracle(oracle, kw):
    """pairs of 5000 values per half (five groups: beyond one workgroup's registers, csrc/mgrit_hip_wide.inc wide2_*): whole solves --
    residual and jump criterion (both halves' groups in the spec's order), weighted C-relaxation, F-cycle, nested iteration --
    against the oracle: the same history, the same states"""
    _need_gpu()
    from pymgrit_amd import Mgrit
    prob = cases.bdf_levels(5002, 17, [2, 1, 1], 2, "one")
    opts = dict(nested_iteration=False, max_iter=4, tol=1e-14)
    opts.update(kw)
    mg = Mgrit(prob, logging_lvl=30, **opts)
    assert mg.backend.name == "hip"
    conv = mg.solve()["conv"]
    op = oracle.OracleProblem([cases.bdf_level_spec(p) for p in prob], variant=1, **opts)
    ref = op.solve()
    assert len(conv) == len(ref) and np.all(np.abs(conv - ref) <= 1e-10 * np.abs(ref)), (conv, ref)
    for lvl in range(mg.lvl_max):
        assert np.array_equal(mg.backend.natural("u", lvl), op.state("u", lvl)), lvl
