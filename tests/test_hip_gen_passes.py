"""GPU: the general whole-level passes (mgrit_hip_gen_down = C-relaxation + F-relaxation + FAS residual with the restriction taken
from registers, mgrit_hip_gen_up = interpolated error correction + F-relaxation [+ residual sums]) for every stepper pair and
transfer they cover -- Heat1D with full weighting (examples/example_spatial_coarsening.py:33-82), Advection1D with its
periodic analogue, the identity transfer -- against the sweep-by-sweep form (PYMGRIT_AMD_NO_GEN_PASSES=1) and the oracle driven
through the reference's sweeps (mgrit.py:277-284, 335-370, 292-333, 488-549, 715-726, 387-413): bit-identical states."""
import os

import numpy as np
import pytest

import cases
import dist_worker

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

CASES = ["heat_spatial_coarsening", "heat_spatial_coarsening_F", "advection_example", "advection_3lvl_F", "advection_nx2049_wide",
         "advection_nx4000_wide_F", "advsc:adv_sc_F", "advsc:adv_sc_V", "heat_nx33_V_cf2"]


def solve(case, gen, blocks=None):
    from pymgrit_amd import Mgrit
    prob, tr, opts = dist_worker.build_problem(case, "hip")
    old = os.environ.pop("PYMGRIT_AMD_NO_GEN_PASSES", None)
    if not gen:
        os.environ["PYMGRIT_AMD_NO_GEN_PASSES"] = "1"
    try:
        mg = Mgrit(prob, transfer=tr, logging_lvl=30, plan_blocks=blocks, **opts)
        used = [lvl for lvl in range(mg.lvl_max - 1) if mg._level_intervals(lvl) is None and mg._gen_intervals(lvl) is not None]
        conv = mg.solve()["conv"]
    finally:
        os.environ.pop("PYMGRIT_AMD_NO_GEN_PASSES", None)
        if old is not None:
            os.environ["PYMGRIT_AMD_NO_GEN_PASSES"] = old
    return conv, [mg.backend.natural("u", lvl) for lvl in range(mg.lvl_max)], used


@pytest.mark.parametrize("case", CASES)
def test_general_passes_bit_identical_to_the_sweeps(case):
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    conv0, u0, used0 = solve(case, False, blocks=1)
    assert used0 == []
    for blocks in (1, None, 3):
        conv, u, used = solve(case, True, blocks=blocks)
        assert used, (case, "no level takes the general passes")
        assert np.array_equal(conv, conv0), (case, blocks, conv, conv0)
        for a, b in zip(u, u0):
            assert np.array_equal(a, b), (case, blocks)


@pytest.mark.parametrize("case", ["heat_spatial_coarsening_F", "advsc:adv_sc_F", "advection_nx2049_wide"])
def test_512_thread_instances_equal_the_1024_thread_instances(case, monkeypatch):
    """states of up to 8192 values launch the passes in instances compiled for 512 threads (no VGPR spills; config 5 5.4 -> 4.9 ms):
    the same source, so the same bits as the 1024-thread instances (MGRIT_HIP_GEN_512=0)"""
    if not torch.cuda.is_available():
        pytest.fail("needs the GPU")
    conv_a, states_a, used_a = solve(case, gen=True)
    monkeypatch.setenv("MGRIT_HIP_GEN_512", "0")
    conv_b, states_b, used_b = solve(case, gen=True)
    assert used_a == used_b and used_a
    assert np.array_equal(conv_a, conv_b)
    for a, b in zip(states_a, states_b):
        assert np.array_equal(a, b)


def _by_hand(mg, op, lvl_pairs, top):
    """a cycle by hand: the general passes on the device, the reference's sweeps on the oracle"""
    from test_hip_parity import assert_state_equal
    be = mg.backend
    for lvl in lvl_pairs:
        iv = mg._gen_intervals(lvl)
        assert iv is not None, lvl
        mg.f_relax(lvl); op.f_relax(lvl)
        head = mg._pairs(lvl, skip_first=False)[:1]
        be.restrict_u(lvl, head)
        be.copy_pairs_u_to_v(lvl, head)
        be.gen_down(lvl, iv)
        op.c_relax(lvl); op.f_relax(lvl); op.fas_residual(lvl)
        # the F-points of the way down are not stored by the pass: compare the C-points of this level and everything below it
        c_idx = [0] + [p[0] for p in mg._pairs(lvl, skip_first=True)]
        assert np.array_equal(be.natural("u", lvl)[c_idx], op.state("u", lvl)[c_idx]), lvl
        for name in ("u", "v", "g"):
            assert np.array_equal(be.natural(name, lvl + 1), op.state(name, lvl + 1)), (name, lvl + 1)
    mg.forward_solve(top); op.forward_solve(top)
    for lvl in reversed(lvl_pairs):
        be.gen_up(lvl, mg._gen_intervals(lvl), residual=(lvl == 0))
        op.error_correction(lvl); op.f_relax(lvl)
        for l in range(lvl, mg.lvl_max):     # (the finer levels still lack the F-points their way down did not store)
            assert np.array_equal(be.natural("u", l), op.state("u", l)), (lvl, l)
    assert_state_equal(mg, op, what=("u",))
    got = np.sqrt(np.array(_fetch(mg, len(mg._c_points(0)))))
    assert np.array_equal(got, op.residual_norms())


def _fetch(mg, n):
    import ctypes as C
    from pymgrit_amd.core import hip_lib
    host = np.empty(n, dtype=np.float64)
    hip_lib.check(mg.backend.lib.mgrit_hip_residual_fetch(mg.backend.h, n, host.ctypes.data_as(C.c_void_p)))
    return host


def test_config5_scale_passes_bit_exact(oracle):
    """config 5's spatial sizes (8192 -> 4096 -> 2048 periodic points, 8 / 4 / 2 groups of lanes), short time grid"""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from test_hip_parity import randomize
    from pymgrit_amd import Advection1D, GridTransferAdvection, Mgrit
    t0 = np.linspace(0, 2.0 * 64 / 32768, 65)
    ts, nxs = [t0, t0[::2], t0[::4]], [8193, 4097, 2049]
    prob = [Advection1D(c=1, x_start=-1, x_end=1, nx=nx, t_interval=t) for nx, t in zip(nxs, ts)]
    mg = Mgrit(prob, transfer=[GridTransferAdvection(), GridTransferAdvection()], nested_iteration=False, logging_lvl=30)
    op = oracle.OracleProblem([cases.advection_level_spec(nx, t) for nx, t in zip(nxs, ts)], transfer=[2, 2], variant=1,
                              nested_iteration=False)
    randomize(mg, op, seed=11)
    _by_hand(mg, op, [0, 1], 2)


@pytest.mark.parametrize("nxs", [(101, 51), (21, 11), (4001, 2001, 1001, 501), (2065, 1033), (41, 21, 11)])
def test_periodic_passes_with_ragged_coarse_sizes(oracle, nxs):
    """periodic coarse grids whose size is NOT a multiple of 8 (a lane holds 8 coarse values): the last coarse value sits in the
    middle of a lane and its right neighbour is value 0 (round-3 advisor finding: the interpolation of the way up took a masked
    zero there). 100 -> 50, 20 -> 10, 4000 -> 2000 -> 1000 -> 500, 2064 -> 1032 (two groups), 40 -> 20 -> 10 unknowns"""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from test_hip_parity import randomize
    from pymgrit_amd import Advection1D, GridTransferAdvection, Mgrit
    t0 = np.linspace(0, 0.05, 2 ** (len(nxs) + 2) + 1)
    ts = [t0[::2 ** k] for k in range(len(nxs))]
    prob = [Advection1D(c=1, x_start=-1, x_end=1, nx=nx, t_interval=t) for nx, t in zip(nxs, ts)]
    mg = Mgrit(prob, transfer=[GridTransferAdvection() for _ in nxs[1:]], nested_iteration=False, logging_lvl=30)
    op = oracle.OracleProblem([cases.advection_level_spec(nx, t) for nx, t in zip(nxs, ts)], transfer=[2] * (len(nxs) - 1), variant=1,
                              nested_iteration=False)
    randomize(mg, op, seed=nxs[0])
    _by_hand(mg, op, list(range(len(nxs) - 1)), len(nxs) - 1)


@pytest.mark.parametrize("nxs", [(4097, 2049, 1025), (131, 66), (37, 19, 10)])
def test_heat_full_weighting_passes_bit_exact(oracle, nxs):
    """Heat1D with full weighting / linear interpolation (n_f = 2 n_c + 1): states of one group and of four, two, one (4095 -> 2047 -> 1023
    unknowns: the neighbour values of the restriction cross lanes and waves), separable forcing"""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from test_hip_parity import randomize
    from pymgrit_amd import GridTransferHeat, Heat1D, Mgrit
    t0 = cases.lin(0.05, 33)
    ts = [t0[::2 ** k] for k in range(len(nxs))]
    prob = [Heat1D(x_start=0, x_end=2, nx=nx, a=1, init_cond=cases.init_cond, rhs_separable=[(cases.rhs_space, cases.rhs_time)],
                   t_interval=t) for nx, t in zip(nxs, ts)]
    mg = Mgrit(prob, transfer=[GridTransferHeat() for _ in nxs[1:]], nested_iteration=False, logging_lvl=30)
    op = oracle.OracleProblem([cases.heat_level_spec(nx, t, x_end=2.0) for nx, t in zip(nxs, ts)], transfer=[1] * (len(nxs) - 1),
                              variant=1, nested_iteration=False)
    randomize(mg, op, seed=3)
    _by_hand(mg, op, list(range(len(nxs) - 1)), len(nxs) - 1)


def test_general_forcing_rows_take_the_passes(oracle):
    """Heat1D with a general (non-separable) forcing -- rows of b_i on the device (FORCE 3) -- and the identity transfer"""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from pymgrit_amd import Heat1D, Mgrit
    nx, nts = 70, (33, 9, 3)
    grids = [cases.lin(1, nt) for nt in nts]
    rhs = lambda x, t: np.sin(3 * x + t) * np.exp(-t * x)      # not separable
    prob = [Heat1D(x_start=0, x_end=1, nx=nx, a=1, init_cond=cases.init_cond, rhs=rhs, t_interval=g) for g in grids]
    mg = Mgrit(prob, nested_iteration=False, max_iter=3, tol=0.0, logging_lvl=30)
    assert mg._level_intervals(0) is None and mg._gen_intervals(0) is not None
    conv = mg.solve()["conv"]
    os.environ["PYMGRIT_AMD_NO_GEN_PASSES"] = "1"
    try:
        mg2 = Mgrit([Heat1D(x_start=0, x_end=1, nx=nx, a=1, init_cond=cases.init_cond, rhs=rhs, t_interval=g) for g in grids],
                    nested_iteration=False, max_iter=3, tol=0.0, logging_lvl=30)
        assert mg2._gen_intervals(0) is None
        conv2 = mg2.solve()["conv"]
    finally:
        os.environ.pop("PYMGRIT_AMD_NO_GEN_PASSES", None)
    assert np.array_equal(conv, conv2)
    for lvl in range(3):
        assert np.array_equal(mg.backend.natural("u", lvl), mg2.backend.natural("u", lvl)), lvl


def test_abi_refuses_what_the_passes_do_not_cover():
    """the way up without the way down of the cycle; the residual on a coarser level"""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    import ctypes as C
    from pymgrit_amd import Advection1D, GridTransferCopy, Mgrit
    t0 = np.linspace(0, 1, 17)
    prob = [Advection1D(c=1, x_start=-1, x_end=1, nx=65, t_interval=t) for t in (t0, t0[::2], t0[::4])]
    mg = Mgrit(prob, transfer=[GridTransferCopy(), GridTransferCopy()], nested_iteration=False, logging_lvl=30)
    be, lib = mg.backend, mg.backend.lib
    iv0, iv1 = mg._gen_intervals(0), mg._gen_intervals(1)
    assert iv0 is not None and iv1 is not None
    id0, id1 = be._intervals_id(0, iv0), be._intervals_id(1, iv1)
    assert lib.mgrit_hip_gen_up(be.h, 0, id0, 1, None) == -1          # nothing has filled the side slab yet
    assert b"gen_down" in lib.mgrit_hip_last_error()
    assert lib.mgrit_hip_gen_down(be.h, 1, id1) == 0
    assert lib.mgrit_hip_gen_up(be.h, 1, id1, 1, None) == -1          # residual: level 0 only
    assert lib.mgrit_hip_gen_up(be.h, 1, id1, 0, None) == 0
    assert lib.mgrit_hip_gen_down(be.h, 0, 99) == -1                  # no such list
    assert lib.mgrit_hip_gen_down(be.h, 2, 0) == -1                   # the coarsest level has no coarser one
    be.sync()
