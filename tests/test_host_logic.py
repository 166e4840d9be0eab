"""CPU tests of the product's host side: layout (vs the reference fixtures), the Mgrit driver on the plugin path
(Dahlquist = BASELINE config 1; heat/advection through oracle-backed plugin Applications), constructor validation,
and the C-ABI export list. No GPU, no compute calls into libmgrit_hip.so."""
import ctypes
import hashlib
import os
import re

import numpy as np
import pytest

import cases
from oracle_apps import OracleApp
from pymgrit_amd import Dahlquist, GridTransferCopy, GridTransferHeat, Mgrit, simple_setup_problem
from pymgrit_amd.core.layout import compute_layout, split_into, split_points

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAYOUT = cases.load_json("layout.json")
SOLVE = cases.load_json("solve.json")
CASES = cases.solve_cases()


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(np.asarray(a, dtype=np.int64)).tobytes()).hexdigest()


@pytest.mark.parametrize("name", sorted(LAYOUT.keys()))
def test_product_layout_matches_reference(name):
    """pymgrit_amd.core.layout (O(n)) vs reference setup_points_and_comm_info fixtures: bit-exact index sets/flags"""
    case = LAYOUT[name]
    ts = cases.layout_case_grids(name, case["spec"])
    for size_s, ranks in case["sizes"].items():
        size = int(size_s)
        for rank, levels in enumerate(ranks):
            for lvl, rec in enumerate(levels):
                lay = compute_layout(ts, lvl, rank, size)
                for flag in ("comm_front", "comm_back", "first_is_c_point", "first_is_f_point", "last_is_c_point",
                             "last_is_f_point"):
                    assert bool(getattr(lay, flag)) == rec[flag], (name, size, rank, lvl, flag)
                assert lay.send_to == rec["send_to"] and lay.get_from == rec["get_from"], (name, size, rank, lvl)
                assert len(lay.t_local) == rec["n_local"]
                if rec["n_local"]:
                    assert lay.t_local[0] == rec["t_first"] and lay.t_local[-1] == rec["t_last"]
                arrs = {"cpts": lay.cpts, "index_local": lay.index_local, "index_local_c": lay.index_local_c,
                        "index_local_f_sorted": np.sort(lay.index_local_f)}
                for k, v in arrs.items():
                    v = np.asarray(v, dtype=np.int64)
                    if k in rec:
                        assert v.tolist() == rec[k], (name, size, rank, lvl, k)
                    else:
                        assert v.size == rec[k + "_len"] and _sha(v) == rec[k + "_sha256"], (name, size, rank, lvl, k)


def test_split_helpers():
    # reference tests/core/test_mgrit.py:39,50-52
    assert split_into(10, 3).tolist() == [4, 3, 3]
    assert tuple(int(x) for x in split_points(10, 3, 0)) == (4, 0)
    assert tuple(int(x) for x in split_points(10, 3, 1)) == (3, 4)
    assert tuple(int(x) for x in split_points(10, 3, 2)) == (3, 7)


def test_rank_overwrite_trick_p7_tables():
    """reference tests/core/test_mgrit.py:86-218: overwrite comm_time_rank/size on ONE Mgrit and re-run
    setup_points_and_comm_info; the literal tables below are that test's expectations (nt=65/17/5, P=7)."""
    problem = [Dahlquist(t_start=0, t_stop=2, nt=n) for n in (65, 17, 5)]
    mgrit = Mgrit(problem=problem, cf_iter=1, nested_iteration=True, max_iter=2, logging_lvl=30)
    got = []
    for rank in range(7):
        mgrit.comm_time_size, mgrit.comm_time_rank = 7, rank
        for name in ("cpts", "comm_front", "comm_back", "index_local_c", "index_local_f", "index_local",
                     "first_is_f_point", "first_is_c_point", "last_is_f_point", "last_is_c_point", "send_to", "get_from"):
            setattr(mgrit, name, [])
        for lvl in range(mgrit.lvl_max):
            mgrit.t.append(np.copy(mgrit.problem[lvl].t))
            mgrit.setup_points_and_comm_info(lvl=lvl)
        got.append({k: list(getattr(mgrit, k)) for k in TABLES})
    for rank in range(7):
        for key, table in TABLES.items():
            for lvl in range(3):
                mine, ref = got[rank][key][lvl], table[rank][lvl]
                if key == "index_local_f":  # CPython set-order artefact: compare as a set
                    assert sorted(np.asarray(mine).tolist()) == sorted(ref), (key, rank, lvl)
                elif isinstance(ref, list):
                    assert np.asarray(mine).tolist() == ref, (key, rank, lvl, mine, ref)
                else:
                    assert mine == ref, (key, rank, lvl, mine, ref)


# literal expectation tables of reference tests/core/test_mgrit.py:143-203 (data), indexed [rank][level]
T, F = True, False
TABLES = {
    "cpts": [[[0, 4, 8], [0], [0]], [[12, 16], [4], [1]], [[20, 24, 28], [], []], [[32, 36], [8], [2]], [[40, 44], [], []],
             [[48, 52], [12], [3]], [[56, 60, 64], [16], [4]]],
    "comm_front": [[F, F, F], [T, T, F], [F, F, F], [F, F, F], [T, T, F], [T, F, F], [F, T, F]],
    "comm_back": [[T, T, F], [F, F, F], [F, F, F], [T, T, F], [T, F, F], [F, T, F], [F, F, F]],
    "index_local": [[[0, 1, 2, 3, 4, 5, 6, 7, 8, 9], [0, 1, 2], [0]], [[1, 2, 3, 4, 5, 6, 7, 8, 9, 10], [1, 2], [1]],
                    [[1, 2, 3, 4, 5, 6, 7, 8, 9], [1, 2, 3], []], [[1, 2, 3, 4, 5, 6, 7, 8, 9], [1, 2], [1]],
                    [[1, 2, 3, 4, 5, 6, 7, 8, 9], [1, 2], []], [[1, 2, 3, 4, 5, 6, 7, 8, 9], [1, 2], [1]],
                    [[1, 2, 3, 4, 5, 6, 7, 8, 9], [1, 2, 3], [1]]],
    "index_local_f": [[[9, 5, 6, 7, 1, 2, 3], [1, 2], []], [[8, 9, 10, 4, 5, 6, 1, 2], [1], []], [[6, 7, 8, 2, 3, 4], [1, 2, 3], []],
                      [[1, 2, 3, 9, 5, 6, 7], [2], []], [[8, 9, 4, 5, 6, 1, 2], [1, 2], []], [[7, 8, 9, 3, 4, 5, 1], [2], []],
                      [[6, 7, 8, 2, 3, 4], [1, 2], []]],
    "index_local_c": [[[0, 4, 8], [0], [0]], [[3, 7], [2], [1]], [[1, 5, 9], [], []], [[4, 8], [1], [1]], [[3, 7], [], []],
                      [[2, 6], [1], [1]], [[1, 5, 9], [3], [1]]],
    "first_is_c_point": [[F, F, F], [F, F, F], [T, F, F], [F, T, F], [F, F, F], [F, T, F], [T, F, F]],
    "first_is_f_point": [[F, F, F], [F, F, F], [F, T, F], [T, F, F], [F, F, F], [F, F, F], [F, F, F]],
    "last_is_f_point": [[F, F, F], [T, F, F], [F, T, F], [F, F, F], [F, T, F], [T, F, F], [F, F, F]],
    "last_is_c_point": [[F, F, F], [F, T, F], [T, F, F], [F, F, F], [F, F, F], [F, F, F], [F, F, F]],
    "send_to": [[1, 1, 1], [2, 2, 3], [3, 3, -99], [4, 4, 5], [5, 5, -99], [6, 6, 6], [-99, -99, -99]],
    "get_from": [[-99, -99, -99], [0, 0, 0], [1, 1, -99], [2, 2, 1], [3, 3, -99], [4, 4, 3], [5, 5, 5]],
}


@pytest.mark.parametrize("name", [n for n in CASES if n.startswith("dahlquist")])
def test_plugin_solve_dahlquist_matches_reference(name):
    """BASELINE config 1 family on the plugin path; scalar arithmetic -> within 4 ulp of the reference history"""
    c = CASES[name]
    prob = [Dahlquist(constant_lambda=s["lambda"], method=s["method"], t_interval=np.asarray(s["t"])) for s in c["levels"]]
    conv = Mgrit(prob, logging_lvl=30, **c["opts"]).solve()["conv"]
    ref = np.array(SOLVE[name]["conv"])
    assert len(conv) == len(ref)
    assert np.all(np.abs(conv - ref) <= 1e-12 * ref + 1e-24), (conv, ref)


def test_config1_example_dahlquist():
    """examples/example_dahlquist.py:14-20 + tests/mpi/results/dahlquist"""
    dahl = simple_setup_problem(problem=Dahlquist(t_start=0, t_stop=5, nt=101), level=2, coarsening=2)
    info = Mgrit(problem=dahl, tol=1e-10, logging_lvl=30).solve()
    ref = np.array(cases.load_json("ref_results.json")["tests_mpi_results"]["dahlquist"])
    np.testing.assert_allclose(info["conv"], ref, rtol=1e-12)
    assert set(info) == {"conv", "time_setup", "time_solve"}


HOST_CASES = ["heat_nx5_test_mgrit", "heat_nx33_V_nested", "heat_nx33_F_nonested", "heat_nx33_V_weight13", "heat_nx33_V_cflist",
              "heat_nx33_V_cf0", "heat_nx33_V_tnorm1", "heat_nx33_V_tnorm3", "heat_nx33_V_jump", "heat_nx33_V_random",
              "heat_nx33_1lvl", "heat_spatial_coarsening", "heat_spatial_coarsening_F", "advection_example",
              "advection_3lvl_F"]


def _oracle_apps(oracle, name):
    c = CASES[name]
    prob = [OracleApp(oracle, s) for s in c["levels"]]
    tr = None
    if c.get("transfer") is not None:
        tr = [GridTransferHeat() if k == 1 else GridTransferCopy() for k in c["transfer"]]
    return c, prob, tr


@pytest.mark.parametrize("name", HOST_CASES)
def test_plugin_driver_equals_oracle_driver(oracle, name):
    """Same Phi (oracle), two independent drivers: pymgrit_amd.Mgrit (plugin path) vs the oracle's C driver.
    Histories must agree to 1e-10 rel (vector norms: np.linalg.norm vs the spec tree) and to the fixture tolerance
    vs the reference."""
    c, prob, tr = _oracle_apps(oracle, name)
    opts = dict(c["opts"])
    if c.get("seed") is not None:
        np.random.seed(c["seed"])
    conv = Mgrit(prob, transfer=tr, logging_lvl=30, **opts).solve()["conv"]
    oopts = {k: v for k, v in opts.items() if k != "random_init_guess"}
    # (the plugin path steps through the coarsest level point by point, as the reference does: the oracle without its
    # time-parallel form of that solve)
    op = oracle.OracleProblem(c["levels"], transfer=c.get("transfer"), variant=1, block_solve=False, **oopts)
    if opts.get("random_init_guess"):
        np.random.seed(c["seed"])
        u = op.state("u", 0)
        for i in range(u.shape[0]):
            u[i] = np.random.rand(u.shape[1])
        u[0] = c["levels"][0]["u0"]
    oconv = op.solve()
    assert len(conv) == len(oconv)
    if len(conv):
        assert np.max(np.abs(conv - oconv) / oconv) <= 1e-10, (conv, oconv)
    ref = np.array(SOLVE[name]["conv"])
    assert np.all(np.abs(conv[:len(ref)] - ref) <= 1e-9 * ref + 2e-11)


def test_constructor_exceptions():
    """reference tests/core/test_mgrit.py:220-233 and mgrit.py:78-128"""
    d0 = Dahlquist(t_start=0, t_stop=5, nt=101)
    d1 = Dahlquist(t_start=0, t_stop=5, nt=51)
    with pytest.raises(Exception):
        Mgrit([d0, d1], transfer=[GridTransferCopy(), GridTransferCopy()], logging_lvl=30)
    with pytest.raises(Exception):
        Mgrit([d1, d0], logging_lvl=30)
    with pytest.raises(Exception):
        Mgrit([d0, d1], cycle_type='W', logging_lvl=30)
    with pytest.raises(Exception):
        Mgrit([d0, d1], output_lvl=3, logging_lvl=30)
    with pytest.raises(Exception):
        Mgrit([d0, Dahlquist(t_start=0, t_stop=5, nt=50)], logging_lvl=30)
    with pytest.raises(Exception):
        Mgrit([d0, d1], t_norm=4, logging_lvl=30)
    with pytest.raises(Exception):
        Mgrit([d0, d1], conv_crit=7, logging_lvl=30)
    with pytest.raises(Exception):
        Mgrit([d0, d1, Dahlquist(t_start=0, t_stop=5, nt=26)], cf_iter=[1], logging_lvl=30)
    with pytest.raises(Exception):
        Mgrit([d0, d1], cf_iter=1.5, logging_lvl=30)


def test_application_contract():
    """reference tests/core/test_application.py: required attributes are enforced after __init__"""
    from pymgrit_amd import Application

    class Bad(Application):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)

        def step(self, u_start, t_start, t_stop):
            return u_start

    with pytest.raises(ValueError):
        Bad(t_start=0, t_stop=1, nt=3)
    with pytest.raises(Exception):
        Dahlquist()
    with pytest.raises(Exception):
        Dahlquist(t_interval=[0, 1, 2])
    with pytest.raises(Exception):
        Dahlquist(method="XX", t_start=0, t_stop=1, nt=3)


def test_output_fcn_and_attributes():
    """output_fcn sees u, t, index_local, comm_time_rank, solve_iter"""
    seen = []

    def out(self):
        seen.append((self.solve_iter, self.comm_time_rank, [self.u[0][i].get_values() for i in self.index_local[0]][-1],
                     float(self.t[0][-1])))
    d = simple_setup_problem(Dahlquist(t_start=0, t_stop=5, nt=101), 2, 2)
    mg = Mgrit(d, tol=1e-10, output_fcn=out, output_lvl=2, logging_lvl=30)
    mg.solve()
    assert len(seen) == 1 + 5 and seen[-1][0] == 5 and seen[-1][3] == 5.0
    assert abs(seen[-1][2] - (1 / 1.05) ** 100) < 1e-10   # backward Euler at t=5
    assert mg.lvl_max == 2 and mg.m == [2, 1] and mg.cf_iter == [1, 1] and len(mg.conv) == 101


def test_log_wire_format(capsys):
    """the 'conv:' INFO lines are the wire format of the reference's MPI harness (tests/mpi/mpi.py:11,23-32)"""
    import logging
    d = simple_setup_problem(Dahlquist(t_start=0, t_stop=5, nt=101), 2, 2)
    logging.getLogger().handlers.clear()
    Mgrit(d, tol=1e-10, logging_lvl=20).solve()
    txt = capsys.readouterr().out
    vals = [float(v) for v in re.findall(r"conv: ([0-9.e+-]+)", txt)]
    ref = cases.load_json("ref_results.json")["tests_mpi_results"]["dahlquist"]
    np.testing.assert_allclose(vals, ref, rtol=1e-12)
    assert "Run parameter overview" in txt and "coarsening factors" in txt


def test_c_abi_exports_every_declared_symbol():
    """libmgrit_hip.so must load without a GPU and export every function include/mgrit_hip.h declares; the ctypes table
    of the product must cover the same set."""
    from pymgrit_amd.core import hip_lib
    header = open(os.path.join(ROOT, "include", "mgrit_hip.h")).read()
    declared = set(re.findall(r"\b(mgrit_hip_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    lib = ctypes.CDLL(hip_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(hip_lib.EXPORTS), declared ^ set(hip_lib.EXPORTS)
    loaded = hip_lib.load()
    assert loaded.mgrit_hip_abi_version() == 3
    assert loaded.mgrit_hip_row_stride(16382) == 16384 and loaded.mgrit_hip_row_stride(3) == 1024
    perm = hip_lib.row_permutation(16382)
    assert len(set(perm.tolist())) == 16382 and perm.max() < 16384


def test_device_application_without_gpu_fails_loudly():
    """no CPU fallback: a device application on a box without a GPU must raise, not silently compute on the host"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pymgrit_amd import Heat1D
    from pymgrit_amd.core.hip_lib import MgritHipError
    prob = [Heat1D(x_start=0, x_end=1, nx=17, a=1, t_start=0, t_stop=1, nt=n) for n in (17, 5)]
    with pytest.raises(MgritHipError):
        Mgrit(prob, logging_lvl=30)


def test_user_subclass_overriding_step_runs_its_own_step():
    """backend by type: a subclass of a device application that overrides step() runs through ITS method on the plugin path
    -- the kernels would silently ignore the override; a transfer that overrides restriction / interpolation runs through its
    methods on either path (on the device path between the kernels: tests/test_hip_user_transfer.py)"""
    from pymgrit_amd import GridTransferCopy, Heat1D, Mgrit
    calls = []

    class MyHeat(Heat1D):
        def step(self, u_start, t_start, t_stop):
            calls.append(t_stop)
            return super().step(u_start, t_start, t_stop)

    class MyCopy(GridTransferCopy):
        def restriction(self, u):
            calls.append("R")
            return super().restriction(u)
    grids = [np.linspace(0, 1, 17), np.linspace(0, 1, 5)]
    prob = [MyHeat(x_start=0, x_end=1, nx=9, a=1, t_interval=t) for t in grids]
    mg = Mgrit(prob, logging_lvl=30, max_iter=1)
    assert mg.backend.name == "plugin"
    mg.solve()
    assert calls
    del calls[:]
    prob = [Heat1D(x_start=0, x_end=1, nx=9, a=1, t_interval=t) for t in grids]
    for p in prob:
        p.device_stepper = lambda: None      # (no GPU here: keep the library applications on the host path too)
    mg = Mgrit(prob, transfer=[MyCopy()], logging_lvl=30, max_iter=1)
    mg.solve()
    assert "R" in calls


def test_library_method_detection():
    from pymgrit_amd import GridTransferCopy, Heat1D, Mgrit

    class Sub(Heat1D):
        pass

    class Over(Heat1D):
        def step(self, u_start, t_start, t_stop):
            return super().step(u_start, t_start, t_stop)
    t = np.linspace(0, 1, 5)
    assert Mgrit._library_method(Sub(x_start=0, x_end=1, nx=9, a=1, t_interval=t), "step")
    assert not Mgrit._library_method(Over(x_start=0, x_end=1, nx=9, a=1, t_interval=t), "step")
    plain = Heat1D(x_start=0, x_end=1, nx=9, a=1, t_interval=t)
    assert Mgrit._library_method(plain, "step")
    plain.step = lambda *a: None
    assert not Mgrit._library_method(plain, "step")
    assert Mgrit._library_method(GridTransferCopy(), "restriction")


def test_fused_pass_lists_of_the_device_path():
    """host side of the whole-level passes (Mgrit._level_intervals / _coarse_down): interval lists, which coarse rows the down
    pass must store (include/mgrit_hip.h, keep), run lists of the coarse-level passes -- checked against a stand-in backend that
    only answers the capability questions"""
    from pymgrit_amd import Heat1D, Mgrit
    grids = [np.linspace(0, 1, 65), np.linspace(0, 1, 17), np.linspace(0, 1, 5)]
    prob = [Heat1D(x_start=0, x_end=1, nx=9, a=1, t_interval=t) for t in grids]
    for p in prob:
        p.device_stepper = lambda: None          # host path: no GPU here
    mg = Mgrit(prob, logging_lvl=30, max_iter=1)

    class Caps:
        def __init__(self, real):
            self.real = real

        def __getattr__(self, name):
            return getattr(self.real, name)
        can_fuse_level = staticmethod(lambda lvl: lvl == 0)
        can_fuse_coarse_down = staticmethod(lambda lvl: lvl > 0)
        can_fuse_ec = staticmethod(lambda lvl: True)
    mg.backend = Caps(mg.backend)
    iv = mg._level_intervals(0)
    assert len(iv) == 16 and iv[0][:5] == (0, 4, -1, 1, 0) and iv[5][:5] == (20, 24, 5, 6, 5)
    keep = [t[5] for t in iv]
    # u of level 1 only at its C-points (every 4th point), v there too (the coarse level's first pass starts its runs from v);
    # conv_crit 0: the way up takes v from the fine C-point, so nowhere else
    assert keep == [3 if (k + 1) % 4 == 0 else 0 for k in range(16)]
    fc_runs, triples, head, skip_u = mg._coarse_down(1)
    assert fc_runs == [(1, 4), (5, 4), (9, 4), (13, 4)] and head == [(0, 0)] and skip_u
    assert triples == [(4, 0, 1), (8, 4, 2), (12, 8, 3), (16, 12, 4)]
    assert mg._coarse_down(0) is None and mg._coarse_down(2) is None
    # a jump criterion does not use the fused way up: v is needed at every closing C-point
    mg2 = Mgrit(prob, logging_lvl=30, max_iter=1, conv_crit=1)
    mg2.backend = Caps(mg2.backend)
    assert all(t[5] & 2 for t in mg2._level_intervals(0))
    # weight != 1, or cf_iter = 2 on level 1: no coarse-level passes
    mg3 = Mgrit(prob, logging_lvl=30, max_iter=1, cf_iter=[1, 2])
    mg3.backend = Caps(mg3.backend)
    assert mg3._coarse_down(1) is None and [t[5] for t in mg3._level_intervals(0)] == [1 if (k + 1) % 4 == 0 else 0 for k in range(16)]


def test_detect_separable_checks_every_time_point():
    """a forcing whose spatial shape deviates only BETWEEN the 16 full-shape probe times is refused (the general forcing rows take
    over), a separable one is accepted unchanged"""
    from pymgrit_amd.heat.heat_1d import NotSeparable, detect_separable
    x = np.linspace(0, 1, 67)[1:-1]
    t = np.linspace(0, 2, 257)
    s, tau = detect_separable(lambda xx, tt: np.sin(np.pi * xx) * np.cos(tt), x, t)
    assert len(s) == 1 and abs(tau[0](0.5) * s[0][10] - np.sin(np.pi * x[10]) * np.cos(0.5)) < 1e-14
    probes = set(np.unique(t[np.unique(np.linspace(0, len(t) - 1, 16).astype(int))]).tolist())
    odd = float([tp for tp in t if tp not in probes][40])

    def rhs(xx, tt):
        base = np.sin(np.pi * xx) * np.cos(tt)
        return base + (xx ** 2 if abs(tt - odd) < 1e-12 else 0.0)
    with pytest.raises(NotSeparable):
        detect_separable(rhs, x, t)


def test_time_factor_of_a_forcing_term_is_the_point_by_point_one():
    """backend_hip._time_factor tries the user's tau(t) on the whole time grid and keeps the result only where it provably is
    what the point-by-point calls of the reference (heat_1d.py:213, one call per step) give"""
    import math
    import numpy as np
    from pymgrit_amd.core.backend_hip import _time_factor
    t = np.linspace(0.0, 2.0, 4097)
    loop = lambda f: np.asarray([f(tt) for tt in t], dtype=np.float64)      # noqa: E731
    smooth = lambda x: np.sin(x) - np.pi ** 2 * np.cos(x)                   # noqa: E731 -- elementwise: the array call is taken
    assert _time_factor(smooth, t).tobytes() == loop(smooth).tobytes()
    branch = lambda x: 1.0 if x < 1.0 else math.exp(-x)                     # noqa: E731 -- raises on an array: point by point
    assert _time_factor(branch, t).tobytes() == loop(branch).tobytes()
    const = lambda x: 3.0                                                   # noqa: E731 -- a scalar back: point by point
    assert _time_factor(const, t).tobytes() == loop(const).tobytes()
    calls = []

    def grid_dependent(x):       # gives something else on an array than on its points: caught by the sampled comparison
        calls.append(np.ndim(x))
        return x - np.mean(x)
    assert _time_factor(grid_dependent, t).tobytes() == loop(grid_dependent).tobytes() and 1 in calls
    short = np.linspace(0.0, 1.0, 33)                                        # short grids are not worth the attempt
    assert _time_factor(smooth, short).tobytes() == np.asarray([smooth(tt) for tt in short]).tobytes()
    #  callables that keep the shape but are NOT position-independent, on a grid too long to compare every point:
    # the value depends on the entry's index parity (a SIMD-tail-like defect that 256 random samples could miss at a single
    # point), on the array's length, or on earlier calls -- the shifted / half-length / repeated calls catch them
    tl = np.linspace(0.0, 2.0, 65537)
    loopl = lambda f: np.asarray([f(tt) for tt in tl], dtype=np.float64)     # noqa: E731

    def last_entry_off(x):
        y = np.sin(x)
        if np.ndim(x):
            y = y.copy()
            y[-1] = np.nextafter(y[-1], 2.0)
        return y
    assert _time_factor(last_entry_off, tl).tobytes() == loopl(last_entry_off).tobytes()

    def length_dependent(x):
        return np.cos(x) * (1.0 + 1e-16 * np.size(x))
    assert _time_factor(length_dependent, tl).tobytes() == loopl(length_dependent).tobytes()
    seen = []

    def stateful(x):
        seen.append(1)
        return np.cos(x) + (1e-13 if len(seen) == 1 else 0.0)
    got = _time_factor(stateful, tl)
    del seen[:]
    seen.append(1)
    assert got.tobytes() == loopl(stateful).tobytes()
    # opting in (pymgrit_amd.elementwise) skips the shifted calls; options.time_factor = 'pointwise' never tries the array
    from pymgrit_amd import elementwise
    from pymgrit_amd.core.options import options
    assert _time_factor(elementwise(smooth), tl).tobytes() == loopl(smooth).tobytes()
    nd = []

    def probe(x):
        nd.append(np.ndim(x))
        return np.sin(x)
    try:
        options.time_factor = "pointwise"
        _time_factor(probe, t)
    finally:
        options.reset("time_factor")
    assert nd and max(nd) == 0


def test_index_array_reads_like_a_list_of_tuples():
    """core/layout.IndexArray (round 5): the run / pair / triple / interval lists of a level as ONE int64 array that reads like the
    list of tuples it replaced -- the plugin backend, the sharded schedules and the tests index, slice, iterate and compare it"""
    from pymgrit_amd.core.layout import IndexArray, as_index_array, consecutive_runs, member_mask
    runs = consecutive_runs(np.array([1, 2, 3, 5, 6, 9]))
    assert isinstance(runs, IndexArray) and runs == [(1, 3), (5, 2), (9, 1)] and runs != [(1, 3)] and len(runs) == 3 and bool(runs)
    assert runs[0] == (1, 3) and runs[-1] == (9, 1) and isinstance(runs[1][0], int)
    assert runs[1:] == [(5, 2), (9, 1)] and isinstance(runs[1:], IndexArray) and runs[:0] == [] and not runs[:0]
    assert [st + ln for st, ln in runs] == [4, 7, 10] and list(runs) == [(1, 3), (5, 2), (9, 1)]
    assert [(0, 0)] + runs == [(0, 0), (1, 3), (5, 2), (9, 1)] and runs + [(7, 7)] == [(1, 3), (5, 2), (9, 1), (7, 7)]
    assert dict(runs) == {1: 3, 5: 2, 9: 1}
    cols = runs.columns()
    assert [c.dtype for c in cols] == [np.int32, np.int32] and cols[0].tolist() == [1, 5, 9] and cols[1].flags["C_CONTIGUOUS"]
    pts = IndexArray(np.array([4, 8, 12]))
    assert pts == [4, 8, 12] and pts[1] == 8 and pts[1:] == [8, 12] and set(pts) == {4, 8, 12} and pts.columns()[0].tolist() == [4, 8, 12]
    assert consecutive_runs(pts) == [(4, 1), (8, 1), (12, 1)] and consecutive_runs(np.zeros(0, dtype=int)) == []
    runs.handle = 7                       # (device handles travel as attributes, as on Mgrit's IndexList)
    assert runs.handle == 7
    assert as_index_array([(1, 2), (3, 4)], 2).shape == (2, 2) and as_index_array([], 3).shape == (0, 3) and as_index_array(runs, 2) is runs.arr
    with pytest.raises(TypeError):
        hash(runs)
    # exact float membership in an ascending grid = np.isin; anything else falls back to it
    grid = np.linspace(0, 2, 17)
    vals = np.concatenate((grid[::4], grid[3:5] + 1e-16, [np.nan, -1.0, 5.0]))
    assert np.array_equal(member_mask(vals, grid), np.isin(vals, grid))
    assert np.array_equal(member_mask(vals, grid[::-1]), np.isin(vals, grid[::-1]))
