"""Time-parallel forward solve of the coarsest level (csrc/mgrit_hip_blk.inc; reference Mgrit.forward_solve,
src/pymgrit/core/mgrit.py:459-486): the HIP path against the oracle's statement of the same arithmetic
(oracle/mgrit_oracle.c heat1d_block_solve_spec) -- states bit for bit --, the rule that selects it (CPU: library against oracle),
and both against the step-by-step form. Whole solves on such levels against the reference's fixtures: tests/test_hip_parity.py::
test_solve_matches_oracle_and_reference[heat_blk_*]."""
import ctypes as C

import numpy as np
import pytest

import cases

torch = pytest.importorskip("torch")


def _grids(nt0, strides):
    ts = [cases.lin(2, nt0)]
    for s in strides:
        ts.append(ts[-1][::s])
    return ts


NONUNIFORM = [cases.BLK_T0, cases.BLK_T0[::2], cases.BLK_T0[::4]]

SHAPES = [
    ("nx33_4blocks", "heat", 33, _grids(1025, (4, 4)), True),             # one group, exactly 4 blocks
    ("nx1024_2lvl", "heat", 1024, _grids(513, (4,)), True),               # 8 blocks
    ("nx1027_rem", "heat", 1027, _grids(309, (4,)), False),               # 77 steps: 4 blocks, the last one of 29; two groups; no forcing
    ("nx2050_nonuniform", "heat", 2050, NONUNIFORM, True),                # every step its own size, 100 steps = 5 x 16 + 20
    ("nx4099", "heat", 4099, _grids(641, (2, 2)), True),                  # 160 steps, five groups
    ("nx16384", "heat", 16384, _grids(129, (2,)), True),                  # the widest register-resident state, 64 steps
    ("nx1024_nonuniform", "heat", 1024, NONUNIFORM, True),                # one group (the one-launch form, round 5): every step its own size, a last block of 20
    ("nx513_128blocks", "heat", 513, _grids(8193, (4,)), True),           # ... its largest grid: 2048 steps = 128 workgroups around two device-wide barriers
    # short time intervals: the blocks damp slowly, more than 64 sine modes take part (up to MGRIT_HIP_BLOCK_RMAX = 256 since round 5)
    ("nx1025_r127", "heat", 1025, [cases.lin(0.02, 1025), cases.lin(0.02, 1025)[::4]], True),
    ("nx4099_r200", "heat", 4099, [cases.lin(0.004, 257), cases.lin(0.004, 257)[::2]], True),
    # Advection1D: all n Fourier modes (n = nx - 1 a power of two)
    ("adv_n64", "advection", 65, _grids(1025, (4, 4)), None),             # the smallest transform, 64 steps
    ("adv_n1024_rem", "advection", 1025, _grids(309, (4,)), None),        # one group, last block of 29 steps
    ("adv_n2048_nonuniform", "advection", 2049, NONUNIFORM, None),        # two groups (config 5's coarsest level), every step its own size
    ("adv_n8192", "advection", 8193, _grids(129, (2,)), None),            # the largest transform: 128 KB of LDS
    # n not a power of two: the transforms as ordered sums on the matrix cores (round 5)
    ("adv_n200", "advection", 201, _grids(1025, (4, 4)), None),           # 64 steps
    ("adv_n1001_rem", "advection", 1002, _grids(309, (4,)), None),        # an odd n, last block of 29 steps
    ("adv_n2000_nonuniform", "advection", 2001, NONUNIFORM, None),        # config-5-like coarsest level at nx = 8001, every step its own size
    ("adv_n6000", "advection", 6001, _grids(129, (2,)), None),            # six groups
]


def test_fourier_rule():
    """Advection1D: all n modes when 64 <= n <= 8192 (radix-2 transforms for a power of two, ordered sums for any other n since
    round 5) and the level has at least 64 steps, else step by step"""
    from pymgrit_amd.core import hip_lib
    lib = hip_lib.load()
    t = np.linspace(0, 1, 200)
    for n, nt, want in [(64, 200, 64), (2048, 65, 2048), (8192, 200, 8192), (16384, 200, 0), (32, 200, 0), (96, 200, 96), (8000, 200, 8000), (8193, 200, 0), (1024, 64, 0)]:
        r = C.c_int(-7)
        hip_lib.check(lib.mgrit_hip_block_solve_rank(hip_lib.STEPPER_ADVECTION1D, n, 3.0, nt, np.ascontiguousarray(t[:nt]).ctypes.data_as(C.c_void_p), C.byref(r)))
        assert r.value == want, (n, nt)


def test_rank_rule_library_equals_oracle(oracle):
    """the rule (how many sine modes, or step by step) is host arithmetic on both sides: same answers on uniform, non-uniform and
    short grids, small and stiff problems"""
    from pymgrit_amd.core import hip_lib
    lib = hip_lib.load()
    rng = np.random.default_rng(5)
    seen = set()
    for trial in range(200):
        nx = int(rng.choice([5, 17, 65, 257, 1024, 4099, 16384]))
        nt = int(rng.integers(2, 400))
        t = np.sort(rng.uniform(0, 2, nt)) if trial % 3 == 0 else np.linspace(0, float(rng.uniform(0.01, 50)), nt)
        spec = cases.heat_level_spec(nx, t)
        r = C.c_int(-7)
        hip_lib.check(lib.mgrit_hip_block_solve_rank(hip_lib.STEPPER_HEAT1D, spec["n"], spec["fac"], nt, np.ascontiguousarray(t).ctypes.data_as(C.c_void_p), C.byref(r)))
        assert r.value == oracle.block_solve_rank(spec["n"], spec["fac"], t), (nx, nt)
        seen.add(min(r.value, 1))
    assert seen == {0, 1}


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind,nx,grids,forcing", SHAPES, ids=[s[0] for s in SHAPES])
def test_forward_solve_bit_exact(oracle, name, kind, nx, grids, forcing):
    from test_hip_parity import _need_gpu, assert_state_equal, make_pair, randomize
    _need_gpu()
    mg, op = make_pair(oracle, kind, nx, grids, **({"forcing": forcing} if kind == "heat" else {}))
    lvl = mg.lvl_max - 1
    if kind == "heat":
        assert mg.backend.block_r[lvl] == oracle.block_solve_rank(op.n[lvl], cases.heat_level_spec(nx, grids[-1])["fac"], grids[-1]) > 0
    else:
        assert mg.backend.block_r[lvl] == op.n[lvl] == nx - 1
    randomize(mg, op, seed=nx)
    for rep in range(2):     # (twice: the second solve starts from rows the first one has written)
        mg.forward_solve(lvl); op.forward_solve(lvl)
        assert_state_equal(mg, op, what=("u",))
    # ... and the cycle around it
    mg.iteration(lvl=0, cycle_type='V', iteration=0, first_f=True); op.iteration(0, 'V', 0, True)
    mg.backend.materialise()
    assert_state_equal(mg, op, what=("u",))
    got, ref = np.array(mg.compute_residual()), op.residual_norms()
    assert np.array_equal(got, ref), np.abs(got - ref).max()


@pytest.mark.gpu
@pytest.mark.parametrize("nx,grids,forcing", [(1024, _grids(4097, (4, 4)), True), (33, _grids(1025, (4, 4)), False), (700, NONUNIFORM, True)],
                         ids=["config2", "nx33", "nx700_nonuniform"])
def test_one_launch_form_equals_the_phase_launches(oracle, monkeypatch, nx, grids, forcing):
    """small Heat1D levels (one group of values, <= 128 blocks, <= 64 modes, one rank) run the three phases of the solve as ONE
    launch with device-wide barriers (blk_one_kernel): the same bits as the per-phase launches (MGRIT_HIP_BLK_ONE=0), solve after
    solve on the same engine (the barrier counters are reused) and through whole cycles; wider levels keep the phases"""
    from test_hip_parity import _need_gpu, make_pair, randomize
    _need_gpu()
    mg1, op = make_pair(oracle, "heat", nx, grids, forcing=forcing)
    monkeypatch.setenv("MGRIT_HIP_BLK_ONE", "0")
    mg6, _ = make_pair(oracle, "heat", nx, grids, forcing=forcing)
    monkeypatch.delenv("MGRIT_HIP_BLK_ONE")
    lvl = mg1.lvl_max - 1
    assert mg1.backend.block_solve_form(lvl) == 2 and mg6.backend.block_solve_form(lvl) == 1
    assert mg1.backend.block_r[lvl] == mg6.backend.block_r[lvl] > 0
    randomize(mg1, op, seed=nx)
    randomize(mg6, op, seed=nx)
    for rep in range(3):
        mg1.forward_solve(lvl); mg6.forward_solve(lvl)
        assert np.array_equal(mg1.backend.natural("u", lvl), mg6.backend.natural("u", lvl))
    for it in range(3):
        mg1.iteration(lvl=0, cycle_type='V', iteration=it, first_f=True); mg6.iteration(lvl=0, cycle_type='V', iteration=it, first_f=True)
    for l in range(mg1.lvl_max):
        assert np.array_equal(mg1.backend.natural("u", l), mg6.backend.natural("u", l))
    wide, _ = make_pair(oracle, "heat", 1027, _grids(513, (4,)))
    assert wide.backend.block_solve_form(1) == 1


@pytest.mark.gpu
def test_sequential_form_on_request(oracle):
    """options.coarse_solve = 'sequential' keeps the step-by-step chain: bit for bit the oracle's step-by-step result, and within
    rounding of the time-parallel one"""
    from pymgrit_amd.core.options import options
    from test_hip_parity import _need_gpu, assert_state_equal, make_pair, randomize
    _need_gpu()
    grids = _grids(513, (4,))
    try:
        options.coarse_solve = "sequential"
        mg, _ = make_pair(oracle, "heat", 1027, grids)
    finally:
        options.reset("coarse_solve")
    assert mg.backend.block_r[1] == 0
    specs = [cases.heat_level_spec(1027, t) for t in grids]
    op = oracle.OracleProblem(specs, variant=1, nested_iteration=False, block_solve=False)
    randomize(mg, op, seed=11)
    mg.forward_solve(1); op.forward_solve(1)
    assert_state_equal(mg, op, what=("u",))
    mg2, op2 = make_pair(oracle, "heat", 1027, grids)
    assert mg2.backend.block_r[1] > 0
    randomize(mg2, op2, seed=11)
    mg2.forward_solve(1)
    a, b = mg.backend.natural("u", 1), mg2.backend.natural("u", 1)
    assert np.abs(a - b).max() <= 1e-11 * np.abs(a).max()
