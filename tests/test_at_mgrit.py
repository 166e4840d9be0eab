"""AT-MGRIT (reference src/pymgrit/core/at_mgrit.py): the product's AtMgrit on the host path and on
the HIP path, the oracle's truncated coarsest solve, against the reference's own known answers
(tests/core/test_at_mgrit.py) and fixtures generated from it (tests/golden/at_mgrit.json, make_golden.py --only-at-mgrit)."""
import numpy as np
import pytest

import cases
from mock_comm import run_ranks
from pymgrit_amd import AtMgrit, Dahlquist, Heat1D, simple_setup_problem

GOLD = cases.load_json("at_mgrit.json")

CASES = {   # name -> (nx, nts, k, options)
    "heat_nx33_k1": (33, [65, 17, 5], 1, dict(tol=1e-9, max_iter=8)),
    "heat_nx33_k2": (33, [65, 17, 5], 2, dict(tol=1e-9, max_iter=8)),
    "heat_nx33_k3": (33, [65, 17, 5], 3, dict(tol=1e-9, max_iter=8)),
    "heat_nx33_k5": (33, [65, 17, 5], 5, dict(tol=1e-9, max_iter=8)),
    "heat_nx33_k3_nonested_F": (33, [129, 33, 9], 3, dict(tol=1e-9, max_iter=8, nested_iteration=False, cycle_type='F')),
    "heat_nx33_2lvl_k4_w13": (33, [65, 17], 4, dict(tol=1e-9, max_iter=8, weight_c=1.3)),
    "heat_nx33_k2_jump": (33, [65, 17, 5], 2, dict(tol=1e-9, max_iter=8, conv_crit=1)),
}


def heat(nx, nts, host_only, x_end=1.0):
    prob = [Heat1D(x_start=0, x_end=x_end, nx=nx, a=1, init_cond=cases.init_cond,
                   rhs_separable=[(cases.rhs_space, cases.rhs_time)], t_start=0, t_stop=2, nt=nt) for nt in nts]
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


def test_reference_known_answer():
    """tests/core/test_at_mgrit.py:34-46"""
    mg = AtMgrit(problem=heat(5, [65, 17, 5], True, x_end=2.0), cf_iter=1, nested_iteration=False, max_iter=2,
                 random_init_guess=False, k=2, logging_lvl=30)
    np.testing.assert_almost_equal(np.array([0.1767778, 0.01223507]), mg.solve()['conv'])


def test_local_criteria_are_refused():
    """tests/core/test_at_mgrit.py:195-206"""
    for crit in (2, 3):
        with pytest.raises(Exception):
            AtMgrit(problem=heat(5, [65, 17, 5], True), cf_iter=1, nested_iteration=False, max_iter=2, conv_crit=crit, k=2)


def test_coarsest_level_layout_tables_p7():
    """rank-overwrite trick of tests/core/test_at_mgrit.py:49-192: who holds which coarsest point, local coarse grids"""
    mg = AtMgrit(problem=heat(5, [65, 17, 5], True, x_end=2.0), cf_iter=1, nested_iteration=True, max_iter=2, k=2, logging_lvl=30)
    expect_grid = [np.array([0.]), np.array([0., 0.5]), None, np.array([0.5, 1.]), None, np.array([1., 1.5]), np.array([1.5, 2.])]
    expect_cpts2 = [[0], [1], [], [2], [], [3], [4]]
    for rank in range(7):
        mg.comm_time_size, mg.comm_time_rank = 7, rank
        mg.int_start = mg.int_stop = 0
        for name in ("cpts", "comm_front", "comm_back", "index_local_c", "index_local_f", "index_local", "first_is_f_point",
                     "first_is_c_point", "last_is_f_point", "last_is_c_point", "send_to", "get_from", "t", "global_t", "_ghost",
                     "_is_c_local"):
            setattr(mg, name, [])
        mg.local_coarse_grid = None
        for lvl in range(mg.lvl_max):
            mg.t.append(np.copy(mg.problem[lvl].t))
            mg.setup_points_and_comm_info(lvl=lvl)
        assert list(mg.cpts[2]) == expect_cpts2[rank]
        if expect_grid[rank] is None:
            assert mg.local_coarse_grid is None
        else:
            assert np.array_equal(mg.local_coarse_grid, expect_grid[rank])
    assert list(mg.comm_coarsest_level) == [0, 1, 3, 5, 6] and list(mg.c_points_per_proc) == [1, 1, 1, 1, 1]


@pytest.mark.parametrize("name", sorted(CASES))
def test_host_path_matches_reference(name):
    nx, nts, k, opts = CASES[name]
    check(AtMgrit(problem=heat(nx, nts, True), k=k, logging_lvl=30, **opts), GOLD[name])


def test_dahlquist_matches_reference():
    d = simple_setup_problem(Dahlquist(t_start=0, t_stop=5, nt=101), level=3, coarsening=2)
    check(AtMgrit(problem=d, k=4, tol=1e-10, logging_lvl=30), GOLD["dahlquist_k4"])


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_reference(oracle, name):
    nx, nts, k, opts = CASES[name]
    for variant in (0, 1):
        op = oracle.OracleProblem([cases.heat_level_spec(nx, cases.lin(2, nt)) for nt in nts], variant=variant,
                                  norm_spec=bool(variant), **opts)
        op.set_at(k)
        conv = op.solve()
        g = GOLD[name]
        assert len(conv) == len(g["conv"])
        assert np.all(np.abs(conv - np.array(g["conv"])) <= 1e-9 * np.array(g["conv"]) + 2e-11)


@pytest.mark.parametrize("size", [2, 3, 5])
@pytest.mark.parametrize("name", ["heat_nx33_k3", "heat_nx33_k3_nonested_F"])
def test_several_ranks_equal_one_rank_on_the_host_path(name, size):
    """the rows of the coarsest level are gathered, then every rank recomputes its own points: any distribution works
    (the reference allows one coarsest point per rank only) and gives the one-rank result"""
    nx, nts, k, opts = CASES[name]
    ref = AtMgrit(problem=heat(nx, nts, True), k=k, logging_lvl=30, **opts)
    conv1 = ref.solve()["conv"]
    u1 = np.array([np.asarray(ref.u[0][i].pack()).ravel() for i in range(len(ref.t[0]))])

    def target(comm):
        mg = AtMgrit(problem=heat(nx, nts, True), k=k, comm_time=comm, logging_lvl=30, **opts)
        conv = mg.solve()["conv"]
        return conv, [np.asarray(mg.u[0][int(i)].pack()).ravel() for i in mg.index_local[0]]
    res = run_ranks(size, target)
    for conv, _ in res:
        assert np.array_equal(conv, conv1)
    assert np.array_equal(np.array([v for _, rows in res for v in rows]), u1)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_path_matches_oracle_and_reference(oracle, name):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    nx, nts, k, opts = CASES[name]
    mg = AtMgrit(problem=heat(nx, nts, False), k=k, logging_lvl=30, **opts)
    assert mg.backend.name == "hip"
    conv = mg.solve()["conv"]
    op = oracle.OracleProblem([cases.heat_level_spec(nx, cases.lin(2, nt)) for nt in nts], variant=1, **opts)
    op.set_at(k)
    ref = op.solve()
    assert len(conv) == len(ref) and np.all(np.abs(conv - ref) <= 1e-10 * np.abs(ref))
    assert np.array_equal(mg.backend.natural("u", 0), op.state("u", 0))
    g = GOLD[name]
    assert np.all(np.abs(conv - np.array(g["conv"])) <= 1e-9 * np.array(g["conv"]) + 2e-11)


@pytest.mark.gpu
def test_hip_truncated_solve_is_bit_exact_on_wide_states(oracle):
    """the coarsest-level launch alone, several groups per state, random u and g"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from test_hip_parity import assert_state_equal, randomize
    for nx, k in ((2050, 3), (5000, 7)):
        grids = [cases.lin(2, 33), cases.lin(2, 17)]
        mg = AtMgrit(problem=heat(nx, [33, 17], False), k=k, logging_lvl=30, nested_iteration=False)
        op = oracle.OracleProblem([cases.heat_level_spec(nx, t) for t in grids], variant=1, nested_iteration=False)
        randomize(mg, op, seed=nx)
        mg.forward_solve(1)
        op.at_forward_solve(1, k)
        assert_state_equal(mg, op)


@pytest.mark.gpu
@pytest.mark.parametrize("name,world", [("heat_nx33_k3_nonested_F", 2), ("heat_nx33_k3_nonested_F", 3), ("heat_nx33_k2", 2),
                                        ("heat_nx33_2lvl_k4_w13", 3)])
def test_hip_several_ranks_equal_one_rank(name, world):
    """AT-MGRIT sharded over ranks on the HIP path (ranks share the one GPU of the test box; gloo transport): halo rows of
    the previous rank through the private work level -- bit-identical to the one-rank run"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from test_distributed import launch
    conv1, u1 = launch(1, "at:" + name, mode="hip")
    conv, u = launch(world, "at:" + name, mode="hip", backend="gloo")
    assert np.array_equal(conv, conv1), (conv, conv1)
    assert np.array_equal(u, u1)


@pytest.mark.parametrize("name,world", [("heat_nx33_k3", 3), ("heat_nx33_k2_jump", 2)])
def test_several_processes_equal_one_on_the_host_path(name, world):
    from test_distributed import launch
    conv1, u1 = launch(1, "at:" + name)
    conv, u = launch(world, "at:" + name)
    assert np.array_equal(conv, conv1) and np.array_equal(u, u1)


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["heat2d_BE", "heat2d_CN", "bdf_narrow", "bdf_wide", "heat_wide", "advection_wide"])
def test_hip_truncated_solve_on_every_device_stepper(oracle, what):
    """round 4: mgrit_hip_at_solve beyond the register-resident 1-D steppers -- Heat2D and the wide 1-D states as batches of the
    level's own Phi launches (one batch per step distance, every point at once), two-point pairs in one workgroup by a kernel of
    their own -- against the oracle's AtMgrit.forward_solve (core/at_mgrit.py:79-87) on random u and g: bit for bit"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from pymgrit_amd import Advection1D
    from test_hip_parity import assert_state_equal, randomize
    for k in (1, 3, 6):
        if what.startswith("heat2d"):
            prob = [cases.h2d_app(19, 23, t, what[-2:], True, cases.H2D_A) for t in cases.h2d_grids([33, 9])]
            specs = [cases.h2d_level_spec(a) for a in prob]
            op = oracle.OracleProblem(specs, nested_iteration=False)
        elif what.startswith("bdf"):
            prob = cases.bdf_levels(35 if what == "bdf_narrow" else 5002, 17, [2, 1], 2, "one")
            op = oracle.OracleProblem([cases.bdf_level_spec(p) for p in prob], variant=1, nested_iteration=False)
        elif what == "heat_wide":
            grids = [cases.lin(2, 17), cases.lin(2, 9)]
            prob = heat(20002, [17, 9], False)
            op = oracle.OracleProblem([cases.heat_level_spec(20002, t) for t in grids], variant=1, nested_iteration=False)
        else:
            grids = [cases.lin(2, 17), cases.lin(2, 9)]
            prob = [Advection1D(c=1, x_start=-1, x_end=1, nx=20001, t_interval=t) for t in grids]
            op = oracle.OracleProblem([cases.advection_level_spec(20001, t) for t in grids], variant=1, nested_iteration=False)
        mg = AtMgrit(problem=prob, k=k, logging_lvl=30, nested_iteration=False)
        assert mg.backend.name == "hip"
        randomize(mg, op, seed=7 + k)
        mg.forward_solve(1)
        op.at_forward_solve(1, k)
        assert_state_equal(mg, op)
