"""GPU: the whole-level passes of level 0 (mgrit_hip_cf_fas = C-relaxation + F-relaxation + FAS residual, mgrit_hip_ec_relax_res =
error correction + F-relaxation + residual sums) against the sweep-by-sweep form (PYMGRIT_AMD_NO_LEVEL_FUSION=1) and the oracle:
same Phi applications on the same values, so everything an iteration leaves behind is bit-identical."""
import os

import numpy as np
import pytest

import cases
import dist_worker

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

CASES = ["heat_nx33_V_nested", "heat_nx33_V_nonested", "heat_nx33_F_nested", "heat_nx33_V_cf2", "heat_nx33_V_cflist", "heat_nx33_2lvl_m8",
         "heat_nx257_nt257", "heat_nx2050_wide", "heat_nx3100_wide_2lvl", "heat_nx33_noforcing", "heat_config2"]


def solve(case, fused, blocks=None):
    from pymgrit_amd import Mgrit
    prob, tr, opts = dist_worker.build_problem(case, "hip")
    old = os.environ.pop("PYMGRIT_AMD_NO_LEVEL_FUSION", None)
    if not fused:
        os.environ["PYMGRIT_AMD_NO_LEVEL_FUSION"] = "1"
    try:
        mg = Mgrit(prob, transfer=tr, logging_lvl=30, plan_blocks=blocks, **opts)
        assert (mg._level_intervals(0) is not None) == fused
        conv = mg.solve()["conv"]
    finally:
        os.environ.pop("PYMGRIT_AMD_NO_LEVEL_FUSION", None)
        if old is not None:
            os.environ["PYMGRIT_AMD_NO_LEVEL_FUSION"] = old
    return conv, [mg.backend.natural("u", lvl) for lvl in range(mg.lvl_max)], mg


@pytest.mark.parametrize("case", CASES)
def test_level_passes_bit_identical(case):
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    conv0, u0, _ = solve(case, False, blocks=1)
    for blocks in (1, None, 4):
        conv, u, mg = solve(case, True, blocks=blocks)
        assert np.array_equal(conv, conv0), (case, blocks, conv, conv0)
        for a, b in zip(u, u0):
            assert np.array_equal(a, b), (case, blocks)


def test_level_passes_against_oracle(oracle):
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from pymgrit_amd import Heat1D, Mgrit
    nx, nts = 3100, (257, 65, 17)
    grids = [cases.lin(2, nt) for nt in nts]
    prob = [Heat1D(x_start=0, x_end=1, nx=nx, a=1, init_cond=cases.init_cond, rhs_separable=[(cases.rhs_space, cases.rhs_time)],
                   t_interval=g) for g in grids]
    mg = Mgrit(prob, logging_lvl=30, max_iter=3, tol=0.0)
    assert mg._level_intervals(0) is not None
    conv = mg.solve()["conv"]
    op = oracle.OracleProblem([cases.heat_level_spec(nx, g) for g in grids], variant=1, max_iter=3, tol=0.0)
    ref = op.solve()
    assert np.max(np.abs(conv - ref) / ref) <= 1e-10, (conv, ref)
    for lvl in range(3):
        assert np.array_equal(mg.backend.natural("u", lvl), op.state("u", lvl)), lvl
    # the per-point residual sums the fused pass delivers are those of the residual kernel
    got = np.asarray(mg.compute_residual())
    mg.backend._residual_cache = None
    assert np.array_equal(got, np.asarray(mg.compute_residual()))


def test_c_point_storage_rebuilds_every_f_point(monkeypatch):
    """the way up stores only the last F-point of every level-0 interval (mgrit_hip_ec_relax_res, store_all_f = 0); whoever
    looks at the solution -- mgrit.u[0][i], natural(), the U slabs, the end of solve() -- sees what an every-point store leaves"""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from pymgrit_amd import Mgrit

    def run(store_all, blocks):
        if store_all:
            monkeypatch.setenv("PYMGRIT_AMD_STORE_ALL_F", "1")
        else:
            monkeypatch.delenv("PYMGRIT_AMD_STORE_ALL_F", raising=False)
        prob, tr, opts = dist_worker.build_problem("heat_nx257_nt257", "hip")
        mg = Mgrit(prob, transfer=tr, logging_lvl=30, plan_blocks=blocks, **opts)
        assert mg._level_intervals(0) is not None
        for it in range(4):       # iterations by hand (no solve()): the fourth replays the captured graph when planned
            mg.iteration(lvl=0, cycle_type='V', iteration=it, first_f=True)
            mg.convergence_criterion(iteration=it + 1)
        return mg
    eager = run(True, 1)
    assert not eager.backend._f_stale
    want = eager.backend.natural("u", 0)
    for blocks in (1, 2, None):
        mg = run(False, blocks)
        assert mg.backend._f_stale                       # F-points pending ...
        row = mg.u[0][2].get_values()                    # ... until somebody looks (an F-point that was not stored)
        assert not mg.backend._f_stale and np.array_equal(row, want[2])
        assert np.array_equal(mg.backend.natural("u", 0), want) and np.array_equal(mg.conv[1:5], eager.conv[1:5])
        mg.iteration(lvl=0, cycle_type='V', iteration=4, first_f=True)
        assert mg.backend._f_stale == 2          # cf_iter = 1: the last F-point's row holds Phi of it (the next C-relaxation's value)
        mg.convergence_criterion(iteration=5)
        mg.c_relax(0)                            # a plain sweep on level 0 settles the rows first
        assert not mg.backend._f_stale
        mg.f_relax(0)
        mg.iteration(lvl=0, cycle_type='V', iteration=5, first_f=True)
        assert mg.backend._f_stale and mg.backend.U[0] is mg.backend._U[0] and not mg.backend._f_stale
    # the same solve with the pre-relaxed rows switched off: same history
    monkeypatch.setenv("PYMGRIT_AMD_NO_PRE_RELAX", "1")
    mg = run(False, None)
    assert mg.backend._f_stale == 1 and np.array_equal(mg.conv[1:5], eager.conv[1:5])


@pytest.mark.parametrize("case", ["heat_nx33_V_nested", "heat_nx257_nt257", "heat_nx3100_wide_2lvl", "heat_nx33_F_nested"])
def test_coarse_level_up_pass_bit_identical(case, monkeypatch):
    """PYMGRIT_AMD_FUSE_UP_COARSE=1: error correction + F-relaxation of the coarser levels through the interval pass that
    corrects the C-point an interval ends on (mgrit_hip_ec_relax_res on lvl > 0) -- same values as the default kernels"""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    conv0, u0, _ = solve(case, True, blocks=1)
    monkeypatch.setenv("PYMGRIT_AMD_FUSE_UP_COARSE", "1")
    for blocks in (1, 3):
        conv, u, mg = solve(case, True, blocks=blocks)
        if mg.lvl_max > 2:
            assert mg._level_intervals(1, up=True) is not None
        assert np.array_equal(conv, conv0), (case, blocks)
        for a, b in zip(u, u0):
            assert np.array_equal(a, b), (case, blocks)


@pytest.mark.parametrize("blocks", [1, None])
def test_first_time_point_is_injected_again_after_a_write(blocks):
    """one rank: the injection of the first time point into the coarser levels is done once and left out of the following
    cycles (nobody relaxes or corrects that point); a write into the level-0 slab from outside brings it back, so the cycles
    after it are those of a solver that started from the written state"""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from pymgrit_amd import Mgrit
    prob, tr, opts = dist_worker.build_problem("heat_nx257_nt257", "hip")
    rng = np.random.default_rng(7)

    def cycles(mg, first, n):
        for it in range(first, first + n):
            mg.iteration(lvl=0, cycle_type='V', iteration=it, first_f=True)
            mg.convergence_criterion(iteration=it + 1)

    a = Mgrit(prob, transfer=tr, logging_lvl=30, plan_blocks=blocks, **opts)
    cycles(a, 0, 3)
    assert a._head_done                                   # the later cycles of `a` ran without the injections
    state = rng.random(a.backend.natural("u", 0).shape)  # every point new, the first one included
    a.backend.set_natural("u", 0, state)
    cycles(a, 3, 3)
    b = Mgrit(prob, transfer=tr, logging_lvl=30, plan_blocks=blocks, **opts)
    b.backend.set_natural("u", 0, state)
    cycles(b, 3, 3)
    for lvl in range(len(prob)):
        assert np.array_equal(a.backend.natural("u", lvl), b.backend.natural("u", lvl)), lvl
    assert np.array_equal(a.conv[4:7], b.conv[4:7])


@pytest.mark.gpu
@pytest.mark.parametrize("nx,switch", [(1024, "MGRIT_HIP_SMALL_WG"), (4099, "MGRIT_HIP_MID_WG"), (8100, "MGRIT_HIP_MID_WG")], ids=["one_wave", "five_waves", "eight_waves"])
@pytest.mark.parametrize("forcing", [True, False], ids=["forcing", "noforcing"])
def test_single_wave_instances_equal_the_1024_thread_instances(oracle, monkeypatch, forcing, nx, switch):
    """levels of one group of values (n <= 1024) launch the cycle's sweeps compiled for ONE wave per workgroup (TB = 64 instances of
    relax<FC>, ecf, cfas, ecfr, fas_fused1: no spill code, the single-group form of Phi), levels of up to 8192 values the instances
    compiled for 512 threads (no VGPR spills); the same source, so the values of the 1024-thread instances (MGRIT_HIP_SMALL_WG=0 /
    MGRIT_HIP_MID_WG=0) over whole V- and F-cycles, and both equal to the oracle"""
    from test_hip_parity import _need_gpu, assert_state_equal, make_pair, randomize
    _need_gpu()
    nt = 257 if nx <= 1024 else 65
    grids = [cases.lin(2, nt), cases.lin(2, nt)[::4], cases.lin(2, nt)[::16]]
    for cyc in ("V", "F"):
        mg_s, op = make_pair(oracle, "heat", nx, grids, forcing=forcing)
        mg_b, _ = make_pair(oracle, "heat", nx, grids, forcing=forcing)
        randomize(mg_s, op, seed=5)
        randomize(mg_b, op, seed=5)
        for it in range(3):
            mg_s.iteration(lvl=0, cycle_type=cyc, iteration=it, first_f=True)
            op.iteration(0, cyc, it, True)
            monkeypatch.setenv(switch, "0")
            mg_b.iteration(lvl=0, cycle_type=cyc, iteration=it, first_f=True)
            mg_b.backend.sync()
            monkeypatch.delenv(switch)
        mg_s.backend.materialise(); mg_b.backend.materialise()
        for l in range(mg_s.lvl_max):
            assert np.array_equal(mg_s.backend.natural("u", l), mg_b.backend.natural("u", l)), (cyc, l)
        assert_state_equal(mg_s, op, what=("u",))

