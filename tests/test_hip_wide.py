"""Heat1D and Advection1D states wider than one workgroup's registers (16384 < n <= 65536 values per time point; the reference has no
limit, heat/heat_1d.py:154-157, advection/advection_1d.py:84-89): the same Phi as three launches over rows in HBM (csrc/mgrit_hip_wide.inc), every sweep against the
oracle's spec variant bit for bit, solves against the oracle (residual history 1e-10, solution bit for bit), on one rank and on
loopback ranks."""
import numpy as np
import pytest

import cases
from test_hip_parity import assert_state_equal, make_pair, randomize

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

G3 = [cases.lin(2, 17), cases.lin(2, 5), cases.lin(2, 3)]
SHAPES = [("n16385", 16387, G3, True), ("n20000", 20002, G3, True), ("n32768", 32770, G3, False), ("n33000_dts", 33002,
          [cases.lin(5, 21), cases.lin(5, 11), cases.lin(5, 6)], True), ("n65536", 65538, [cases.lin(2, 9), cases.lin(2, 5), cases.lin(2, 3)], True)]


@pytest.mark.parametrize("name,nx,grids,forcing", SHAPES, ids=[s[0] for s in SHAPES])
def test_wide_sweeps_bit_exact(oracle, name, nx, grids, forcing):
    assert torch.cuda.is_available()
    for w in (1.0, 1.3):
        mg, op = make_pair(oracle, "heat", nx, grids, weight_c=w, forcing=forcing)
        assert mg.backend.n[0] > 16384 and mg._level_intervals(0) is None       # sweep by sweep: no fused pass holds such a state
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


ADV_SHAPES = [("adv_n16400", 16401, G3), ("adv_n32768", 32769, G3), ("adv_n40001_dts", 40002, [cases.lin(5, 21), cases.lin(5, 11), cases.lin(5, 6)]),
              ("adv_n65536", 65537, [cases.lin(2, 9), cases.lin(2, 5), cases.lin(2, 3)])]


@pytest.mark.parametrize("name,nx,grids", ADV_SHAPES, ids=[s[0] for s in ADV_SHAPES])
def test_wide_advection_sweeps_bit_exact(oracle, name, nx, grids):
    """Advection1D states of 16385 .. 65536 periodic values (the reference has no limit, advection/advection_1d.py:84-89): every
    sweep through the three-launch Phi against the oracle, bit for bit"""
    assert torch.cuda.is_available()
    mg, op = make_pair(oracle, "advection", nx, grids)
    assert mg.backend.n[0] > 16384
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


def test_wide_advection_solve_with_spatial_coarsening(oracle):
    """a wide periodic fine level over register-resident coarse ones (periodic full weighting), F-cycle: residual history and states"""
    assert torch.cuda.is_available()
    t0 = cases.lin(2, 33)
    mg, op = make_pair(oracle, "advection", [32769, 16385, 8193], [t0, t0[::2], t0[::4]], transfer=[2, 2], cycle_type='F', max_iter=3, tol=0.0)
    conv, oconv = mg.solve()["conv"], op.solve()
    assert len(conv) == len(oconv) and np.max(np.abs(conv - oconv) / oconv) <= 1e-10, (conv, oconv)
    assert_state_equal(mg, op, what=("u",))


def test_wide_spatial_coarsening_and_jump(oracle):
    """a wide fine level over a register-resident coarse one (full weighting / linear interpolation), and the jump criterion"""
    assert torch.cuda.is_available()
    t0 = cases.lin(2, 33)
    mg, op = make_pair(oracle, "heat", [32769, 16385, 8193], [t0, t0[::2], t0[::4]], transfer=[1, 1], x_end=2.0)
    randomize(mg, op, seed=5)
    for lvl in range(2):
        mg.fas_residual(lvl); op.fas_residual(lvl)
        assert_state_equal(mg, op)
    for lvl in (1, 0):
        mg.error_correction(lvl); op.error_correction(lvl)
        assert_state_equal(mg, op)
    mg2, op2 = make_pair(oracle, "heat", 20002, G3, conv_crit=1, max_iter=3, tol=0.0)
    conv, oconv = mg2.solve()["conv"], op2.solve()
    assert len(conv) == len(oconv) and np.max(np.abs(conv - oconv) / oconv) <= 1e-10, (conv, oconv)


@pytest.mark.parametrize("cycle,nested", [("V", True), ("F", False)])
def test_wide_solve_matches_oracle(oracle, cycle, nested):
    """Heat1D(nx = 32770), three levels: the solve of the verdict's item 7"""
    assert torch.cuda.is_available()
    grids = [cases.lin(2, 65), cases.lin(2, 17), cases.lin(2, 5)]
    mg, op = make_pair(oracle, "heat", 32770, grids, cycle_type=cycle, nested_iteration=nested, max_iter=4, tol=0.0)
    conv, oconv = mg.solve()["conv"], op.solve()
    assert len(conv) == len(oconv) and np.max(np.abs(conv - oconv) / oconv) <= 1e-10, (conv, oconv)
    assert_state_equal(mg, op, what=("u",))


def test_wide_states_on_ranks(oracle):
    """two and three loopback ranks (device exchange of 256 KB rows): bit-identical to one rank"""
    assert torch.cuda.is_available()
    from pymgrit_amd import Mgrit
    from pymgrit_amd.core.comm import run_loopback_ranks
    from test_hip_parity import heat_problem
    grids = [cases.lin(2, 33), cases.lin(2, 9), cases.lin(2, 3)]

    def target(comm):
        mg = Mgrit(heat_problem(20002, grids), logging_lvl=30, comm_time=comm, max_iter=3, tol=0.0)
        conv = mg.solve()["conv"]
        return conv, np.array([np.asarray(mg.u[0][int(i)].pack()).ravel() for i in mg.index_local[0]])
    _, (one,) = run_loopback_ranks(1, target)
    for world in (2, 3):
        w, res = run_loopback_ranks(world, target)
        w.close()
        assert all(np.array_equal(r[0], one[0]) for r in res)
        assert np.array_equal(np.concatenate([r[1] for r in res if r[1].size], axis=0), one[1])
