#!/usr/bin/env python3
"""Generate golden fixtures by running the *reference* (PyMGRIT, /root/reference) in the build container.

The reference is pure Python and cannot travel to the GPU box, so this script imports it here (with the
size-1 ``mpi4py`` stand-in in ``tests/golden/_mpi_stub``) and writes small data-only fixtures:

  tests/golden/layout.json      index sets / comm flags per (case, P, rank, level) via the rank-overwrite
                                trick of reference tests/core/test_mgrit.py:86-218
  tests/golden/phi.npz/.json    known-answer Application.step outputs (Heat1D, Advection1D, Dahlquist)
  tests/golden/solve.json       residual histories (full precision) + selected solution vectors
  tests/golden/solve_restated.json  the reference's Mgrit driven by a numpy restatement of the oracle's Thomas step (pins
                                the solver logic at 1e-10), and a general (non-separable) forcing
  tests/golden/ref_results.json the reference's own tests/mpi/results/* files (data) + literal KATs cited

Usage:  python tests/golden/make_golden.py            (needs /root/reference; ~2 min)
Nothing here is imported by the product; fixtures are data (inputs + expected outputs) only.
"""
import hashlib
import json
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("PYMGRIT_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "_mpi_stub"))
sys.path.insert(0, os.path.join(REF, "src"))
warnings.filterwarnings("ignore")

import numpy as np  # noqa: E402
from pymgrit.core.mgrit import Mgrit  # noqa: E402
from pymgrit.core.grid_transfer_copy import GridTransferCopy  # noqa: E402
from pymgrit.core.simple_setup_problem import simple_setup_problem  # noqa: E402
from pymgrit.dahlquist.dahlquist import Dahlquist  # noqa: E402
from pymgrit.heat.heat_1d import Heat1D, VectorHeat1D  # noqa: E402
from pymgrit.advection.advection_1d import Advection1D, VectorAdvection1D  # noqa: E402

QUIET = 30


def rhs(x, t):
    return - np.sin(np.pi * x) * (np.sin(t) - 1 * np.pi ** 2 * np.cos(t))


def init_cond(x):
    return np.sin(np.pi * x)


# --------------------------------------------------------------------------------------------------
# layout fixtures
# --------------------------------------------------------------------------------------------------
def digest(arr):
    a = np.ascontiguousarray(np.asarray(arr, dtype=np.int64))
    return hashlib.sha256(a.tobytes()).hexdigest()


def layout_for(problem, size, full):
    """rank-overwrite trick (reference tests/core/test_mgrit.py:109-129)."""
    mgrit = Mgrit(problem=problem, nested_iteration=False, logging_lvl=QUIET)
    out = []
    for rank in range(size):
        mgrit.comm_time_size = size
        mgrit.comm_time_rank = rank
        mgrit.int_start = 0
        mgrit.int_stop = 0
        for name in ("cpts", "comm_front", "comm_back", "index_local_c", "index_local_f", "index_local",
                     "first_is_f_point", "first_is_c_point", "last_is_f_point", "last_is_c_point",
                     "send_to", "get_from"):
            setattr(mgrit, name, [])
        for lvl in range(mgrit.lvl_max):
            mgrit.t.append(np.copy(mgrit.problem[lvl].t))
            mgrit.setup_points_and_comm_info(lvl=lvl)
        levels = []
        for lvl in range(mgrit.lvl_max):
            f_sorted = np.sort(np.asarray(mgrit.index_local_f[lvl], dtype=np.int64))
            rec = {
                "n_local": int(len(mgrit.t[lvl])),
                "t_first": float(mgrit.t[lvl][0]) if len(mgrit.t[lvl]) else None,
                "t_last": float(mgrit.t[lvl][-1]) if len(mgrit.t[lvl]) else None,
                "comm_front": bool(mgrit.comm_front[lvl]), "comm_back": bool(mgrit.comm_back[lvl]),
                "first_is_c_point": bool(mgrit.first_is_c_point[lvl]),
                "first_is_f_point": bool(mgrit.first_is_f_point[lvl]),
                "last_is_c_point": bool(mgrit.last_is_c_point[lvl]),
                "last_is_f_point": bool(mgrit.last_is_f_point[lvl]),
                "send_to": int(mgrit.send_to[lvl]), "get_from": int(mgrit.get_from[lvl]),
                "m": int(mgrit.m[lvl]),
            }
            arrays = {"cpts": mgrit.cpts[lvl], "index_local": mgrit.index_local[lvl],
                      "index_local_c": mgrit.index_local_c[lvl], "index_local_f_sorted": f_sorted}
            for k, v in arrays.items():
                v = np.asarray(v, dtype=np.int64)
                if full:
                    rec[k] = v.tolist()
                else:
                    rec[k + "_len"] = int(v.size)
                    rec[k + "_sha256"] = digest(v)
                    rec[k + "_first"] = int(v[0]) if v.size else None
                    rec[k + "_last"] = int(v[-1]) if v.size else None
            if full:
                # the exact (CPython set-iteration dependent) order of the reference, informational
                rec["index_local_f_ref_order"] = np.asarray(mgrit.index_local_f[lvl], dtype=np.int64).tolist()
            levels.append(rec)
        out.append(levels)
    return out


def dahl_levels(ts):
    return [Dahlquist(t_interval=t) for t in ts]


def make_layout():
    cases = {}

    def add(name, spec, ts, sizes, full=True):
        problem = dahl_levels(ts)
        cases[name] = {"spec": spec, "sizes": {}}
        for p in sizes:
            cases[name]["sizes"][str(p)] = layout_for(problem, p, full)

    t65 = np.linspace(0, 2, 65)
    add("nt65_m4_m4", {"t_stop": 2, "nt": [65, 17, 5], "kind": "linspace"},
        [t65, np.linspace(0, 2, 17), np.linspace(0, 2, 5)], [1, 2, 3, 4, 5, 7, 8])
    t101 = np.linspace(0, 5, 101)
    add("nt101_m2", {"t_stop": 5, "nt": [101, 51], "kind": "stride", "strides": [2]},
        [t101, t101[::2]], [1, 2, 3, 4, 5])
    add("nt101_m2_m2", {"t_stop": 5, "nt": [101, 51, 26], "kind": "stride", "strides": [2, 2]},
        [t101, t101[::2], t101[::4]], [1, 2, 3, 5])
    t129 = np.linspace(0, 5, 129)
    l1 = t129[::16]
    add("nt129_m16_2_2_2", {"t_stop": 5, "nt": [129], "kind": "stride", "strides": [16, 2, 2, 2]},
        [t129, l1, l1[::2], l1[::4], l1[::8]], [1, 2, 3, 4, 5, 6, 7])
    t65b = np.linspace(0, 5, 65)
    idx = [0, 3, 10, 12, 14, 17, 23, 27, 33, 34, 55, 57, 59, 61, 63, 64]
    v1 = t65b[idx]
    add("nt65_varying", {"t_stop": 5, "nt": [65], "kind": "index", "index": idx, "strides": [2, 2, 2]},
        [t65b, v1, v1[::2], v1[::4], v1[::8]], [1, 2, 3, 4, 5, 6, 7])
    t4097 = np.linspace(0, 2, 4097)
    add("nt4097_m4_m4", {"t_stop": 2, "nt": [4097], "kind": "stride", "strides": [4, 4]},
        [t4097, t4097[::4], t4097[::16]], [1, 3, 4, 8], full=False)
    t65537 = np.linspace(0, 2, 65537)
    add("nt65537_m4_m4", {"t_stop": 2, "nt": [65537], "kind": "stride", "strides": [4, 4]},
        [t65537, t65537[::4], t65537[::16]], [8], full=False)
    t16385 = np.linspace(0, 1, 16385)
    add("nt16385_m8", {"t_stop": 1, "nt": [16385], "kind": "stride", "strides": [8]},
        [t16385, t16385[::8]], [8], full=False)
    return cases


# --------------------------------------------------------------------------------------------------
# Phi known-answer fixtures
# --------------------------------------------------------------------------------------------------
def heat_input(x, k):
    return np.sin(np.pi * x) + 0.25 * np.sin(3 * np.pi * x + 0.1 * k) + 0.05 * np.cos(17.0 * x * x)


def make_phi():
    meta = {"heat1d": [], "advection1d": [], "dahlquist": []}
    arrays = {}
    # Heat1D: (nx, x_end, a, t_start, t_stop, forcing?)
    heat_cases = [(6, 1.0, 1.0, 0.0, 0.1, False), (6, 1.0, 1.0, 0.0, 0.1, True), (5, 2.0, 1.0, 0.5, 0.53125, True),
                  (17, 2.0, 1.0, 1.0, 1.015625, True), (33, 1.0, 0.7, 0.25, 0.375, True),
                  (1001, 1.0, 1.0, 0.0, 0.03125, True), (1024, 1.0, 1.0, 1.0, 1.0 + 2.0 / 4096, True),
                  (1024, 1.0, 1.0, 0.5, 0.5 + 2.0 / 256, True),
                  (4096, 1.0, 1.0, 0.5, 0.5 + 2.0 / 1024, True),
                  (16384, 1.0, 1.0, 1.0, 1.0 + 2.0 / 65536, True), (16384, 1.0, 1.0, 1.0, 1.0 + 2.0 / 4096, True)]
    for k, (nx, x_end, a, t0, t1, forcing) in enumerate(heat_cases):
        app = Heat1D(x_start=0, x_end=x_end, nx=nx, a=a, init_cond=init_cond,
                     rhs=rhs if forcing else (lambda x, t: x * 0), t_start=0, t_stop=2, nt=3)
        u = VectorHeat1D(app.nx)
        u.set_values(heat_input(app.x, k))
        out = app.step(u, t0, t1).get_values()
        key = f"heat1d_{k}"
        arrays[key] = out
        meta["heat1d"].append({"key": key, "nx": nx, "x_end": x_end, "a": a, "t_start": t0, "t_stop": t1,
                               "forcing": forcing, "input": "heat_input(x,k)", "k": k})
    # Advection1D
    adv_cases = [(6, 1.0, 0.0, 0.1), (129, 1.0, 0.0, 2.0 / 128), (129, 1.0, 1.0, 1.0 + 2.0 / 64), (8193, 1.0, 0.0, 2.0 / 32768),
                 (4097, 0.5, 0.0, 2.0 / 16384)]
    for k, (nx, c, t0, t1) in enumerate(adv_cases):
        app = Advection1D(c=c, x_start=-1, x_end=1, nx=nx, t_start=0, t_stop=2, nt=3)
        u = VectorAdvection1D(app.nx)
        u.set_values(np.exp(-app.x ** 2) + 0.1 * np.sin(5 * np.pi * app.x + k))
        out = app.step(u, t0, t1).get_values()
        key = f"advection1d_{k}"
        arrays[key] = out
        meta["advection1d"].append({"key": key, "nx": nx, "c": c, "t_start": t0, "t_stop": t1, "k": k})
    for method in ("BE", "FE", "TR", "MR"):
        app = Dahlquist(method=method, t_start=0, t_stop=5, nt=11)
        from pymgrit.dahlquist.dahlquist import VectorDahlquist
        vals = [float(app.step(VectorDahlquist(0.75), 0.3, 0.3 + h).get_values()) for h in (0.05, 0.5, 0.1)]
        meta["dahlquist"].append({"method": method, "u": 0.75, "t_start": 0.3, "h": [0.05, 0.5, 0.1], "out": vals})
    return meta, arrays


# --------------------------------------------------------------------------------------------------
# solve fixtures
# --------------------------------------------------------------------------------------------------
sys.path.insert(0, os.path.join(REF, "examples"))
from example_spatial_coarsening import GridTransferHeat  # noqa: E402  (reference class, imported not copied)


def run(problem, sample_pts=(), sample_lvl0=True, **kw):
    kw.setdefault("logging_lvl", QUIET)
    m = Mgrit(problem=problem, **kw)
    info = m.solve()
    rec = {"conv": [float(c) for c in info["conv"]]}
    if sample_lvl0:
        rec["samples"] = {}
        for i in sample_pts:
            vals = m.u[0][i].get_values()
            rec["samples"][str(i)] = np.asarray(vals, dtype=float).ravel().tolist()
    return rec


def heat_levels(nx, nts, x_end=1.0, a=1.0, t_stop=2.0, forcing=True):
    return [Heat1D(x_start=0, x_end=x_end, nx=nx, a=a, init_cond=init_cond,
                   rhs=rhs if forcing else (lambda x, t: x * 0), t_start=0, t_stop=t_stop, nt=nt) for nt in nts]


def make_solve(big=True):
    out = {}
    # --- Dahlquist family (config 1 = example_dahlquist.py)
    d = simple_setup_problem(Dahlquist(t_start=0, t_stop=5, nt=101), level=2, coarsening=2)
    out["dahlquist_config1"] = run(d, tol=1e-10, sample_pts=(0, 1, 50, 100))
    out["dahlquist_3lvl"] = run(simple_setup_problem(Dahlquist(t_start=0, t_stop=5, nt=101), 3, 2), tol=1e-10,
                                sample_pts=(100,))
    out["dahlquist_F"] = run(simple_setup_problem(Dahlquist(t_start=0, t_stop=5, nt=129), 4, 2), tol=1e-10,
                             cycle_type='F', sample_pts=(128,))
    out["dahlquist_time_integrators"] = run([Dahlquist(t_start=0, t_stop=5, nt=101, method='MR'),
                                             Dahlquist(t_start=0, t_stop=5, nt=51, method='BE')],
                                            sample_pts=(100,))
    for meth in ("FE", "TR"):
        out[f"dahlquist_{meth}"] = run(simple_setup_problem(Dahlquist(t_start=0, t_stop=5, nt=101, method=meth), 2, 2),
                                       tol=1e-10, sample_pts=(100,))
    t129 = np.linspace(0, 5, 129)
    l1 = t129[::16]
    out["dahlquist_procs_without_points"] = run(dahl_levels([t129, l1, l1[::2], l1[::4], l1[::8]]), tol=1e-10,
                                                sample_pts=(128,))
    t65b = np.linspace(0, 5, 65)
    v1 = t65b[[0, 3, 10, 12, 14, 17, 23, 27, 33, 34, 55, 57, 59, 61, 63, 64]]
    out["dahlquist_varying_coarsening"] = run(dahl_levels([t65b, v1, v1[::2], v1[::4], v1[::8]]), tol=1e-10,
                                              nested_iteration=False, sample_pts=(64,))
    # --- Heat1D small (tests/core/test_mgrit.py:59-70)
    h = heat_levels(5, [65, 17, 5], x_end=2.0)
    out["heat_nx5_test_mgrit"] = run(h, cf_iter=1, nested_iteration=True, max_iter=2, sample_pts=(1, 32, 64))
    out["heat_nx5_to_tol"] = run(h, cf_iter=1, nested_iteration=True, max_iter=12, tol=1e-12, sample_pts=(64,))
    # --- Heat1D nx=33, nt=65, 3 levels m=4: option sweep
    def h33():
        return heat_levels(33, [65, 17, 5])
    opts = {
        "V_nested": dict(), "V_nonested": dict(nested_iteration=False),
        "F_nested": dict(cycle_type='F'), "F_nonested": dict(cycle_type='F', nested_iteration=False),
        "V_weight13": dict(weight_c=1.3, nested_iteration=False), "V_cf2": dict(cf_iter=2),
        "V_cflist": dict(cf_iter=[2, 1, 1]), "V_cf0": dict(cf_iter=0), "V_tnorm1": dict(t_norm=1), "V_tnorm3": dict(t_norm=3),
        "V_jump": dict(conv_crit=1), "F_weight13_cf2": dict(cycle_type='F', weight_c=1.3, cf_iter=2),
    }
    for name, kw in opts.items():
        out["heat_nx33_" + name] = run(h33(), tol=1e-9, max_iter=8, sample_pts=(1, 33, 64), **kw)
    np.random.seed(0)
    out["heat_nx33_V_random"] = run(h33(), tol=1e-9, max_iter=8, random_init_guess=True, nested_iteration=False,
                                    sample_pts=(64,))
    out["heat_nx33_2lvl_m8"] = run(heat_levels(33, [65, 9]), tol=1e-9, max_iter=8, sample_pts=(64,))
    out["heat_nx33_1lvl"] = run(heat_levels(33, [65]), max_iter=2, sample_pts=(1, 64))
    out["heat_nx33_noforcing"] = run(heat_levels(33, [65, 17, 5], forcing=False), tol=1e-9, max_iter=8, sample_pts=(64,))
    # --- shrunken config 3: nx=257 nt=257 3-level m=4 (stiffer)
    out["heat_nx257_nt257"] = run(heat_levels(257, [257, 65, 17]), tol=1e-9, max_iter=10, sample_pts=(128, 256))
    # --- example_heat_1d.py / example_weighted_jacobi.py (tests/mpi/results/heat_1d, weighted_jacobi)
    if big:
        h5 = heat_levels(1001, [65, 33, 17, 9, 5])
        out["heat_example_F5"] = run(h5, cf_iter=1, cycle_type='F', nested_iteration=False, max_iter=10,
                                     random_init_guess=False, sample_pts=(64,))
        out["heat_example_F5_w13"] = run(h5, weight_c=1.3, tol=1e-8, cf_iter=1, cycle_type='F', nested_iteration=False,
                                         max_iter=10, sample_pts=(64,))
    # --- spatial coarsening (examples/example_spatial_coarsening.py:112-123)
    h0 = Heat1D(x_start=0, x_end=2, nx=2 ** 4 + 1, a=1, rhs=rhs, init_cond=init_cond, t_start=0, t_stop=2, nt=2 ** 7 + 1)
    h1 = Heat1D(x_start=0, x_end=2, nx=2 ** 3 + 1, a=1, rhs=rhs, init_cond=init_cond, t_interval=h0.t[::2])
    h2 = Heat1D(x_start=0, x_end=2, nx=2 ** 2 + 1, a=1, rhs=rhs, init_cond=init_cond, t_interval=h1.t[::2])
    h3 = Heat1D(x_start=0, x_end=2, nx=2 ** 2 + 1, a=1, rhs=rhs, init_cond=init_cond, t_interval=h2.t[::2])
    out["heat_spatial_coarsening"] = run([h0, h1, h2, h3],
                                         transfer=[GridTransferHeat(), GridTransferHeat(), GridTransferCopy()],
                                         sample_pts=(1, 64, 128))
    g0 = Heat1D(x_start=0, x_end=1, nx=129, a=1, rhs=rhs, init_cond=init_cond, t_start=0, t_stop=2, nt=129)
    g1 = Heat1D(x_start=0, x_end=1, nx=65, a=1, rhs=rhs, init_cond=init_cond, t_interval=g0.t[::4])
    g2 = Heat1D(x_start=0, x_end=1, nx=33, a=1, rhs=rhs, init_cond=init_cond, t_interval=g1.t[::4])
    out["heat_spatial_coarsening_F"] = run([g0, g1, g2], transfer=[GridTransferHeat(), GridTransferHeat()],
                                           cycle_type='F', tol=1e-9, max_iter=10, sample_pts=(128,))
    # --- advection (examples/example_advection.py)
    a0 = Advection1D(c=1, x_start=-1, x_end=1, nx=129, t_start=0, t_stop=2, nt=129)
    a1 = Advection1D(c=1, x_start=-1, x_end=1, nx=129, t_start=0, t_stop=2, nt=65)
    out["advection_example"] = run([a0, a1], cf_iter=1, nested_iteration=False, max_iter=12, sample_pts=(1, 128))
    b = [Advection1D(c=1, x_start=-1, x_end=1, nx=257, t_start=0, t_stop=2, nt=nt) for nt in (257, 65, 17)]
    out["advection_3lvl_F"] = run(b, cycle_type='F', max_iter=8, sample_pts=(256,))
    # --- config 2 (BASELINE.json configs[1]): heat_1d nx=1024 nt=4097 3-level m=4, V/FCF, nested
    if big:
        c2 = heat_levels(1024, [4097, 1025, 257])
        rec = run(c2, cf_iter=1, nested_iteration=True, max_iter=4, tol=1e-30, sample_pts=(2048, 4096))
        out["heat_config2"] = rec
    out.update(make_solve_wide())
    out.update(make_solve_block())
    return out


def make_solve_wide():
    """states wider than one group of 1024 values (several workgroups per state on the GPU): the coarsest
    level runs through the overlapped chain of the spec; full and partial last group, V and F cycles, with / without forcing"""
    out = {}
    out["heat_nx2050_wide"] = run(heat_levels(2050, [65, 17, 5]), tol=1e-9, max_iter=6, sample_pts=(1, 33, 64))
    out["heat_nx1500_wide_F"] = run(heat_levels(1500, [33, 9, 3], forcing=False), tol=1e-9, max_iter=6, cycle_type='F',
                                    nested_iteration=False, sample_pts=(16, 32))
    out["heat_nx3100_wide_2lvl"] = run(heat_levels(3100, [33, 9]), tol=1e-9, max_iter=2, cf_iter=2, sample_pts=(32,))
    return out


def make_solve_block():
    """coarsest levels long enough for the time-parallel forward solve of the device path (inputs restated in
    tests/cases.block_cases)"""
    out = {}
    out["heat_blk_nx257_3lvl"] = run(heat_levels(257, [1025, 257, 65]), tol=1e-9, max_iter=6, sample_pts=(1, 512, 1024))
    out["heat_blk_nx2050_2lvl"] = run(heat_levels(2050, [513, 129]), tol=1e-9, max_iter=4, cf_iter=2, sample_pts=(256, 512))
    t0 = 2 * np.linspace(0, 1, 401) ** 1.3
    nu = [Heat1D(x_start=0, x_end=1, nx=129, a=1, init_cond=init_cond, rhs=rhs, t_interval=t0[::s]) for s in (1, 2, 4)]
    out["heat_blk_nonuniform_F"] = run(nu, tol=1e-9, max_iter=6, cycle_type='F', sample_pts=(3, 200, 400))
    out["heat_blk_noforcing_cf0"] = run(heat_levels(65, [513, 129], forcing=False), tol=1e-9, max_iter=6, cf_iter=0,
                                        nested_iteration=False, sample_pts=(512,))

    def adv(nx, ts):
        return [Advection1D(c=1, x_start=-1, x_end=1, nx=nx, t_interval=t) for t in ts]
    out["advection_blk_nx257_3lvl_F"] = run(adv(257, [np.linspace(0, 2, nt) for nt in (1025, 257, 65)]), cycle_type='F', max_iter=6,
                                            sample_pts=(512, 1024))
    out["advection_blk_nx1025_2lvl"] = run(adv(1025, [np.linspace(0, 2, nt) for nt in (513, 129)]), max_iter=4, sample_pts=(512,))
    out["advection_blk_nonuniform"] = run(adv(129, [t0[::s] for s in (1, 2, 4)]), max_iter=6, nested_iteration=False,
                                          sample_pts=(7, 400))
    return out


# --------------------------------------------------------------------------------------------------
# Heat2D fixtures (heat_2d.py): Phi known answers for BE/CN/FE with and without boundary values, small MGRIT solves
# --------------------------------------------------------------------------------------------------
H2D_X_END, H2D_Y_END, H2D_A = 0.75, 1.5, 3.5


def h2d_rhs(x, y, t):   # forcing of examples/example_heat_2d.py:36-46
    return 5 * x * (H2D_X_END - x) * y * (H2D_Y_END - y) + 10 * H2D_A * t * (y * (H2D_Y_END - y) + x * (H2D_X_END - x))


def h2d_input(x, y, k):
    return np.sin(3 * x + 0.3 * k) * np.cos(2 * y) + 0.1 * x * y


H2D_BC = dict(bc_left=1.5, bc_right=lambda s: 2 + s, bc_top=-1.0, bc_bottom=lambda s: s * s)


def make_heat2d():
    from pymgrit.heat.heat_2d import Heat2D, VectorHeat2D
    meta, arrays = {"phi": [], "solve": {}}, {}
    k = 0
    for method in ("BE", "CN", "FE"):
        for with_bc in (False, True):
            for nx, ny in ((9, 12), (20, 17)):
                kw = H2D_BC if with_bc else {}
                app = Heat2D(x_start=0, x_end=H2D_X_END, y_start=0, y_end=H2D_Y_END, nx=nx, ny=ny, a=H2D_A, rhs=h2d_rhs,
                             method=method, t_start=0, t_stop=1, nt=33, **kw)
                u = VectorHeat2D(nx, ny)
                u.set_values(h2d_input(app.x_2d, app.y_2d, k) * np.ones((nx, ny)))
                out = app.step(u, app.t[3], app.t[4]).get_values()
                key = f"h2d_phi_{k}"
                arrays[key] = np.asarray(out).reshape(nx, ny)
                meta["phi"].append({"key": key, "method": method, "bc": with_bc, "nx": nx, "ny": ny, "k": k, "i_stop": 4})
                k += 1

    def levels(nx, ny, nts, method="BE", with_bc=False, a=H2D_A):
        kw = H2D_BC if with_bc else {}
        t0 = np.linspace(0, 1, nts[0])
        ts = [t0]
        for n in nts[1:]:
            ts.append(ts[-1][::(len(ts[-1]) - 1) // (n - 1)])
        return [Heat2D(x_start=0, x_end=H2D_X_END, y_start=0, y_end=H2D_Y_END, nx=nx, ny=ny, a=a, rhs=h2d_rhs,
                       method=method, t_interval=t, **kw) for t in ts]

    def solve(name, prob, **kw):
        m = Mgrit(problem=prob, logging_lvl=QUIET, **kw)
        info = m.solve()
        meta["solve"][name] = {"conv": [float(c) for c in info["conv"]]}
        arrays["h2d_solve_" + name] = np.asarray(m.u[0][len(prob[0].t) - 1].get_values())

    # examples/example_heat_2d.py shrunk (55x125x33 there); without nested iteration so that a history exists
    solve("example_small", levels(17, 21, [33, 17]), cycle_type='V', nested_iteration=False, tol=1e-9, max_iter=8)
    solve("be_3lvl_F_bc", levels(12, 10, [65, 17, 5], with_bc=True), cycle_type='F', tol=1e-9, max_iter=8)
    solve("cn_2lvl", levels(12, 10, [65, 33], method="CN", a=0.1), tol=1e-9, max_iter=8, nested_iteration=False)
    solve("be_weight_bc", levels(10, 14, [33, 9], with_bc=True), weight_c=1.2, tol=1e-9, max_iter=8, nested_iteration=False)
    solve("fe_2lvl", levels(8, 8, [257, 65], method="FE", a=0.05), tol=1e-9, max_iter=6, nested_iteration=False)
    # coarsest levels of >= 64 steps: the device path's time-parallel forward solve (backward Euler)
    solve("be_blk_2lvl_bc", levels(12, 10, [513, 65], with_bc=True), tol=1e-9, max_iter=6, nested_iteration=False)
    solve("be_blk_3lvl_F", levels(9, 14, [641, 161, 81]), cycle_type='F', tol=1e-9, max_iter=5, nested_iteration=False)
    # ... with Crank-Nicolson (round 5): boundary values, an F-cycle, and cf_iter = 0 with boundary values -- C-points that still hold
    # the zero initial guess (another rim than the boundary values) in the first cycle: the device path steps that solve
    solve("cn_blk_2lvl_bc", levels(12, 10, [513, 65], method="CN", with_bc=True, a=0.1), tol=1e-9, max_iter=6, nested_iteration=False)
    solve("cn_blk_3lvl_F", levels(9, 14, [641, 161, 81], method="CN", a=0.1), cycle_type='F', tol=1e-9, max_iter=5, nested_iteration=False)
    solve("cn_blk_cf0_bc", levels(10, 11, [513, 129], method="CN", with_bc=True, a=0.1), cf_iter=0, tol=1e-9, max_iter=6,
          nested_iteration=False)
    with open(os.path.join(HERE, "heat2d.json"), "w") as f:
        json.dump(meta, f, indent=1)
    np.savez_compressed(os.path.join(HERE, "heat2d.npz"), **arrays)
    print("wrote heat2d fixtures", {k: v["conv"][:2] for k, v in meta["solve"].items()})


def h2d_general_rhs(x, y, t):   # neither separable nor linear in time: what only a general callable can express
    return np.exp(x * t) * y + np.cos(3 * t + y) * x


def make_heat2d_general():
    """Heat2D with a forcing that is NOT of the form S0 + S1*t (the reference takes any callable, heat_2d.py:148,289-320):
    one known-answer step per method and small solves; tests/golden/heat2d_general.json/.npz"""
    from pymgrit.heat.heat_2d import Heat2D, VectorHeat2D
    meta, arrays = {"phi": [], "solve": {}}, {}
    for k, method in enumerate(("BE", "CN", "FE")):
        nx, ny = 11, 14
        app = Heat2D(x_start=0, x_end=H2D_X_END, y_start=0, y_end=H2D_Y_END, nx=nx, ny=ny, a=0.05 if method == "FE" else H2D_A,
                     rhs=h2d_general_rhs, method=method, t_start=0, t_stop=1, nt=33, **H2D_BC)
        u = VectorHeat2D(nx, ny)
        u.set_values(h2d_input(app.x_2d, app.y_2d, k) * np.ones((nx, ny)))
        arrays[f"phi_{method}"] = np.asarray(app.step(u, app.t[3], app.t[4]).get_values()).reshape(nx, ny)
        meta["phi"].append({"key": f"phi_{method}", "method": method, "nx": nx, "ny": ny, "k": k, "i_stop": 4})

    def solve(name, nx, ny, nts, method, a, **kw):
        t0 = np.linspace(0, 1, nts[0])
        ts = [t0]
        for n in nts[1:]:
            ts.append(ts[-1][::(len(ts[-1]) - 1) // (n - 1)])
        prob = [Heat2D(x_start=0, x_end=H2D_X_END, y_start=0, y_end=H2D_Y_END, nx=nx, ny=ny, a=a, rhs=h2d_general_rhs, method=method,
                       t_interval=t, **H2D_BC) for t in ts]
        m = Mgrit(problem=prob, logging_lvl=QUIET, **kw)
        info = m.solve()
        meta["solve"][name] = {"conv": [float(c) for c in info["conv"]], "nx": nx, "ny": ny, "nts": nts, "method": method, "a": a}
        arrays["solve_" + name] = np.asarray(m.u[0][len(prob[0].t) - 1].get_values())
    solve("general_be_3lvl", 12, 10, [65, 17, 5], "BE", H2D_A, tol=1e-9, max_iter=8)
    solve("general_cn_2lvl_F", 10, 13, [65, 33], "CN", 0.1, tol=1e-9, max_iter=8, nested_iteration=False)
    solve("general_fe_2lvl", 8, 8, [257, 65], "FE", 0.05, tol=1e-9, max_iter=6, nested_iteration=False)
    with open(os.path.join(HERE, "heat2d_general.json"), "w") as f:
        json.dump(meta, f, indent=1)
    np.savez_compressed(os.path.join(HERE, "heat2d_general.npz"), **arrays)
    print("wrote heat2d_general fixtures", {k: v["conv"][:2] for k, v in meta["solve"].items()})


# --------------------------------------------------------------------------------------------------
# Advection1D with periodic spatial coarsening (BASELINE config 5 shrunk). The reference has no periodic transfer class;
# the fixture runs the REFERENCE Mgrit with a user GridTransfer that states the periodic full-weighting / linear
# arithmetic on the reference's own vector class.
# --------------------------------------------------------------------------------------------------
def make_advection_sc():
    from pymgrit.core.grid_transfer import GridTransfer

    class PeriodicTransfer(GridTransfer):
        def restriction(self, u):
            f = u.get_values()
            n = len(f)
            ret = np.zeros(n // 2)
            for i in range(n // 2):
                ret[i] = f[(2 * i - 1) % n] * 1 / 4 + f[2 * i] * 1 / 2 + f[(2 * i + 1) % n] * 1 / 4
            out = VectorAdvection1D(n // 2)
            out.set_values(ret)
            return out

        def interpolation(self, u):
            c = u.get_values()
            nc = len(c)
            ret = np.zeros(2 * nc)
            for i in range(nc):
                ret[2 * i] += c[i]
                ret[2 * i + 1] += 1 / 2 * c[i]
                ret[(2 * i - 1) % (2 * nc)] += 1 / 2 * c[i]
            out = VectorAdvection1D(2 * nc)
            out.set_values(ret)
            return out

    out = {}
    t0 = np.linspace(0, 2, 129)
    for name, nxs, strides, kw in (("adv_sc_F", [129, 65, 33, 33], [2, 2, 2], dict(cycle_type='F', max_iter=8)),
                                   ("adv_sc_V", [65, 33], [4], dict(max_iter=8, nested_iteration=False))):
        ts = [t0]
        for st in strides:
            ts.append(ts[-1][::st])
        prob = [Advection1D(c=1, x_start=-1, x_end=1, nx=nx, t_interval=t) for nx, t in zip(nxs, ts)]
        tr = [PeriodicTransfer() if nxs[k] != nxs[k + 1] else GridTransferCopy() for k in range(len(nxs) - 1)]
        m = Mgrit(problem=prob, transfer=tr, logging_lvl=QUIET, **kw)
        info = m.solve()
        out[name] = {"nx": nxs, "strides": strides, "conv": [float(c) for c in info["conv"]],
                     "u_last": np.asarray(m.u[0][128].get_values()).tolist()}
    with open(os.path.join(HERE, "advection_sc.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote advection_sc", {k: v["conv"][:3] for k, v in out.items()})


# --------------------------------------------------------------------------------------------------
# Two-point BDF applications (heat_1d_2pts_bdf1.py / heat_1d_2pts_bdf2.py, example_heat_1d_bdf2.py)
# --------------------------------------------------------------------------------------------------
def bdf_rhs2(x, t):      # a second forcing form: two separable terms
    return rhs(x, t) + np.sin(2 * np.pi * x) * t


def make_bdf():
    from pymgrit.heat.heat_1d_2pts_bdf1 import Heat1DBDF1
    from pymgrit.heat.heat_1d_2pts_bdf2 import Heat1DBDF2
    cls = {1: Heat1DBDF1, 2: Heat1DBDF2}

    def levels(nx, n_pairs, orders, coarsening, forcing):
        """pairs (t, t + dtau) on n_pairs points of [0, 2], levels coarsened by `coarsening` (example_heat_1d_bdf2.py:56-66)"""
        dtau = 2.0 / (2 * (n_pairs - 1))
        t = np.linspace(0, 2, n_pairs)
        out = []
        for lvl, order in enumerate(orders):
            f = {"zero": lambda x, tt: x * 0, "one": rhs, "two": bdf_rhs2}[forcing]
            out.append(cls[order](x_start=0, x_end=1, nx=nx, a=1, dtau=dtau, rhs=f, init_cond=init_cond,
                                  t_interval=t[::coarsening ** lvl]))
        return out

    out = {"phi": {}, "solve": {}}
    # single Phi applications on a fixed input pair, uniform and non-uniform pair spacing
    x = np.linspace(0, 1, 35)[1:-1]
    for order in (1, 2):
        for forcing in ("zero", "one", "two"):
            app = levels(35, 17, [order], 2, forcing)[0]
            v = app.vector_template.clone_zero()
            v.set_values(heat_input(x, 0), heat_input(x, 1), app.vector_template.dtau)
            for name, (ta, tb) in {"uniform": (app.t[3], app.t[4]), "wide": (app.t[2], app.t[6])}.items():
                r = app.step(v, float(ta), float(tb))
                out["phi"][f"bdf{order}_{forcing}_{name}"] = {
                    "t_start": float(ta), "t_stop": float(tb),
                    "first": np.asarray(r.get_values()[0]).tolist(), "second": np.asarray(r.get_values()[1]).tolist()}
            out["phi"][f"bdf{order}_{forcing}_t0"] = {"first": np.asarray(app.vector_t_start.get_values()[0]).tolist(),
                                                      "second": np.asarray(app.vector_t_start.get_values()[1]).tolist()}
    cases = {
        "bdf2_example_small": dict(nx=35, n_pairs=33, orders=[2, 1, 1], coarsening=2, forcing="one", kw=dict(tol=1e-9, max_iter=10)),
        "bdf1_2lvl_m4": dict(nx=35, n_pairs=33, orders=[1, 1], coarsening=4, forcing="one", kw=dict(tol=1e-9, max_iter=10)),
        "bdf2_2lvl_F_two": dict(nx=19, n_pairs=33, orders=[2, 2, 1], coarsening=2, forcing="two",
                                kw=dict(tol=1e-9, max_iter=10, cycle_type='F', nested_iteration=False)),
        "bdf2_weighted_jump": dict(nx=19, n_pairs=17, orders=[2, 1], coarsening=2, forcing="zero",
                                   kw=dict(tol=1e-9, max_iter=8, weight_c=1.2, conv_crit=1)),
    }
    for name, c in cases.items():
        prob = levels(c["nx"], c["n_pairs"], c["orders"], c["coarsening"], c["forcing"])
        m = Mgrit(problem=prob, logging_lvl=QUIET, **c["kw"])
        info = m.solve()
        rec = {"conv": [float(v) for v in info["conv"]], "samples": {}}
        for i in (1, (c["n_pairs"] - 1) // 2, c["n_pairs"] - 1):
            a, b, _ = m.u[0][i].get_values()
            rec["samples"][str(i)] = [np.asarray(a).tolist(), np.asarray(b).tolist()]
        out["solve"][name] = rec
    with open(os.path.join(HERE, "bdf.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote bdf", {k: v["conv"][:3] for k, v in out["solve"].items()})


# --------------------------------------------------------------------------------------------------
# Local convergence criteria (conv_crit 2 / 3, mgrit.py:434-455) on one rank
# --------------------------------------------------------------------------------------------------
def make_local_conv():
    out = {}
    for crit in (2, 3):
        d = simple_setup_problem(Dahlquist(t_start=0, t_stop=5, nt=101), level=2, coarsening=2)
        out[f"dahlquist_crit{crit}"] = run(d, tol=1e-10, conv_crit=crit, sample_pts=(100,))
        out[f"heat_nx33_crit{crit}"] = run(heat_levels(33, [65, 17, 5]), tol=1e-7, max_iter=12, conv_crit=crit, sample_pts=(64,))
        out[f"heat_nx33_crit{crit}_maxiter"] = run(heat_levels(33, [65, 17, 5]), tol=1e-14, max_iter=3, conv_crit=crit,
                                                  sample_pts=(64,))
    with open(os.path.join(HERE, "local_conv.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote local_conv", {k: len(v["conv"]) for k, v in out.items()})


# --------------------------------------------------------------------------------------------------
# AT-MGRIT (core/at_mgrit.py), one rank
# --------------------------------------------------------------------------------------------------
def make_at_mgrit():
    from pymgrit.core.at_mgrit import AtMgrit
    out = {}

    def run_at(problem, k, sample_pts=(), **kw):
        kw.setdefault("logging_lvl", QUIET)
        m = AtMgrit(problem=problem, k=k, **kw)
        info = m.solve()
        rec = {"conv": [float(c) for c in info["conv"]], "samples": {}}
        for i in sample_pts:
            rec["samples"][str(i)] = np.asarray(m.u[0][i].get_values(), dtype=float).ravel().tolist()
        return rec
    for k in (1, 2, 3, 5):
        out[f"heat_nx33_k{k}"] = run_at(heat_levels(33, [65, 17, 5]), k, tol=1e-9, max_iter=8, sample_pts=(1, 33, 64))
    out["heat_nx33_k3_nonested_F"] = run_at(heat_levels(33, [129, 33, 9]), 3, tol=1e-9, max_iter=8, nested_iteration=False,
                                            cycle_type='F', sample_pts=(128,))
    out["heat_nx33_2lvl_k4_w13"] = run_at(heat_levels(33, [65, 17]), 4, tol=1e-9, max_iter=8, weight_c=1.3, sample_pts=(64,))
    out["heat_nx33_k2_jump"] = run_at(heat_levels(33, [65, 17, 5]), 2, tol=1e-9, max_iter=8, conv_crit=1, sample_pts=(64,))
    d = simple_setup_problem(Dahlquist(t_start=0, t_stop=5, nt=101), level=3, coarsening=2)
    out["dahlquist_k4"] = run_at(d, 4, tol=1e-10, sample_pts=(100,))
    with open(os.path.join(HERE, "at_mgrit.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote at_mgrit", {k: len(v["conv"]) for k, v in out.items()})


# --------------------------------------------------------------------------------------------------
# Local convergence criteria on SEVERAL ranks (the drain protocol, mgrit.py:434-455,648-691): the reference's true multi-rank
# path on the thread-backed MPI stand-in (tests/golden/_mpi_stub: run_world)
# --------------------------------------------------------------------------------------------------
def make_local_conv_ranks():
    from mpi4py import MPI
    cases = {
        "heat_crit2_V": (lambda: heat_levels(33, [65, 17, 5]), dict(tol=1e-7, max_iter=12, conv_crit=2)),
        "heat_crit3_V": (lambda: heat_levels(33, [65, 17, 5]), dict(tol=1e-7, max_iter=12, conv_crit=3)),
        "heat_crit2_F_nonested": (lambda: heat_levels(33, [65, 17, 5]), dict(tol=1e-7, max_iter=12, conv_crit=2, cycle_type='F',
                                                                         nested_iteration=False)),
        "heat_crit2_maxiter": (lambda: heat_levels(33, [65, 17, 5]), dict(tol=1e-14, max_iter=3, conv_crit=2)),
        "heat_crit2_2lvl_cf2": (lambda: heat_levels(33, [65, 9]), dict(tol=1e-7, max_iter=12, conv_crit=2, cf_iter=2)),
        "dahlquist_crit2": (lambda: simple_setup_problem(Dahlquist(t_start=0, t_stop=5, nt=101), level=3, coarsening=2),
                            dict(tol=1e-10, conv_crit=2)),
    }
    out = {}
    for name, (make, kw) in cases.items():
        for size in (2, 3, 4):
            def one(rank, make=make, kw=kw):
                m = Mgrit(problem=make(), logging_lvl=QUIET, **kw)
                info = m.solve()
                own = [int(i) for i in m.index_local[0]]
                pick = sorted({0, len(own) // 2, len(own) - 1})     # owned points sampled: first, middle, last
                vals = {str(j): np.asarray(m.u[0][own[j]].get_values(), dtype=float).ravel().tolist() for j in pick}
                return {"conv": [float(c) for c in info["conv"]], "n_owned": len(own), "u": vals}
            out[f"{name}_P{size}"] = MPI.run_world(size, one)
    with open(os.path.join(HERE, "local_conv_ranks.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote local_conv_ranks", {k: [len(r["conv"]) for r in v] for k, v in out.items()})


class Labelled:
    """a message of the reference together with the level it was sent for"""

    def __init__(self, data, lvl):
        self.data, self.lvl = data, lvl


def make_exchange_fuzz():
    """the reference on tests/fuzz_cases.py's random hierarchies at the SAME rank count (threads as ranks): with
    non-uniform coarsening its results depend on the rank count, so the one-rank run is no yardstick there"""
    from mpi4py import MPI
    sys.path.insert(0, os.path.dirname(HERE))
    from fuzz_cases import N_CASES, SEED0, random_case
    # the reference's tags are base[op] + (messages so far on THAT level, op) (mgrit.py:192-195): two levels can hold the
    # same tag at once, and the farewell messages of a finished rank (clean_up, mgrit.py:648-668) are then matched to the
    # wrong level's receive. Label every message with its level to notice; such runs are recorded but are no yardstick.
    crossed = []
    ref_send, ref_receive = Mgrit.send, Mgrit.receive

    def send(self, data, dest, lvl, op_id):
        ref_send(self, Labelled(data, lvl), dest, lvl, op_id)

    def receive(self, source, lvl, op_id):
        got = ref_receive(self, source, lvl, op_id)
        if got.lvl != lvl:
            crossed.append((lvl, got.lvl, op_id))
        return got.data
    Mgrit.send, Mgrit.receive = send, receive
    out = {}
    for seed in range(SEED0, SEED0 + N_CASES):
        grids, opts, size, _ = random_case(seed)
        size = min(size, len(grids[0]))
        del crossed[:]

        def one(rank, grids=grids, opts=opts):
            m = Mgrit(problem=[Dahlquist(t_interval=np.asarray(g)) for g in grids], logging_lvl=QUIET, **opts)
            info = m.solve()
            # the reference takes its F-point order from the iteration order of a Python set (mgrit.py:771-775); when the
            # local index span exceeds the set's table the order is scrambled and an F-point can be updated BEFORE the
            # F-point it depends on -- such runs of the reference are recorded but are no yardstick
            scrambled = False
            for order in m.index_local_f:
                seen = set()
                members = set(int(i) for i in order)
                for i in (int(i) for i in order):
                    scrambled |= (i - 1 in members and i - 1 not in seen)
                    seen.add(i)
            return {"conv": [float(c) for c in info["conv"]], "scrambled": bool(scrambled),
                    "u": [float(np.asarray(m.u[0][int(i)].get_values()).ravel()[0]) for i in m.index_local[0]]}
        ranks = MPI.run_world(size, one, timeout=120)
        out[str(seed)] = {"size": size, "n": [len(g) for g in grids], "scrambled": any(r.pop("scrambled") for r in ranks),
                          "crossed": bool(crossed), "ranks": ranks}
    Mgrit.send, Mgrit.receive = ref_send, ref_receive
    with open(os.path.join(HERE, "exchange_fuzz.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote exchange_fuzz", os.path.getsize(os.path.join(HERE, "exchange_fuzz.json")),
          "scrambled F order in", sum(v["scrambled"] for v in out.values()), "crossed messages in",
          sum(v["crossed"] for v in out.values()), "of", len(out))


# --------------------------------------------------------------------------------------------------
# SURVEY stage (B): the reference's OWN Mgrit (cycle logic, FAS operand order, transfers, stopping test) driven by a
# Heat1D whose step is the plain-numpy restatement of the oracle's variant-0 Phi (Thomas algorithm, same operation order as
# oracle/mgrit_oracle.c heat1d_rhs + thomas_toeplitz). With the SuperLU rounding out of the way, the oracle has to
# reproduce these residual histories to the north-star tolerance (1e-10 relative per iteration; in practice ~1e-15).
# Also: a general (non-separable) forcing, through the reference's SuperLU step and through the restated step.
# --------------------------------------------------------------------------------------------------
def general_rhs(x, t):      # not of the form s(x)*tau(t)
    return np.exp(-x * t) * np.cos(3.0 * x + t) + x * x * t


class RestatedHeat1D(Heat1D):
    """reference Heat1D with step = Thomas algorithm in the oracle's operation order (no SuperLU)"""

    def step(self, u_start, t_start, t_stop):
        dt = t_stop - t_start
        beta, diag = dt * self.fac_, dt * (2.0 * self.fac_) + 1.0
        d = u_start.get_values() + self.rhs(self.x, t_stop) * dt
        n = d.shape[0]
        cp = np.empty(n)
        piv = diag
        cp[0] = -beta / piv
        d[0] = d[0] / piv
        for j in range(1, n):
            piv = diag + beta * cp[j - 1]
            cp[j] = -beta / piv
            d[j] = (d[j] + beta * d[j - 1]) / piv
        out = np.empty(n)
        out[n - 1] = d[n - 1]
        for j in range(n - 2, -1, -1):
            out[j] = d[j] - cp[j] * out[j + 1]
        ret = VectorHeat1D(n)
        ret.set_values(out)
        return ret


def restated_levels(nxs, ts, x_end=1.0, forcing=rhs, cls=RestatedHeat1D):
    out = []
    for nx, t in zip(nxs, ts):
        app = cls(x_start=0, x_end=x_end, nx=nx, a=1.0, init_cond=init_cond, rhs=forcing, t_interval=t)
        app.fac_ = 1.0 / (app.x[1] - app.x[0]) ** 2      # a / dx^2 exactly as cases.heat_level_spec computes it
        out.append(app)
    return out


def make_restated():
    out = {}
    t0 = np.linspace(0, 2, 129)
    ts3 = [t0, t0[::4], t0[::16]]

    def rec(problem, sample_pts, **kw):
        r = run(problem, sample_pts=sample_pts, **kw)
        return r
    out["restated_V"] = rec(restated_levels([65] * 3, ts3), (1, 64, 128), tol=1e-13, max_iter=10)
    out["restated_F_w13"] = rec(restated_levels([65] * 3, ts3), (128,), tol=1e-13, max_iter=10, cycle_type='F', weight_c=1.3,
                                nested_iteration=False)
    out["restated_cf2_nonested"] = rec(restated_levels([33] * 3, ts3), (128,), tol=1e-13, max_iter=10, cf_iter=2,
                                       nested_iteration=False)
    tsc = [t0, t0[::2], t0[::4], t0[::8]]
    out["restated_spatial_coarsening"] = rec(restated_levels([17, 9, 5, 5], tsc, x_end=2.0), (128,),
                                             transfer=[GridTransferHeat(), GridTransferHeat(), GridTransferCopy()],
                                             tol=1e-13, max_iter=10)
    out["restated_general_forcing"] = rec(restated_levels([65] * 3, ts3, forcing=general_rhs), (1, 128), tol=1e-13, max_iter=10)
    out["superlu_general_forcing"] = rec(restated_levels([65] * 3, ts3, forcing=general_rhs, cls=Heat1D), (1, 128),
                                         tol=1e-9, max_iter=10)
    with open(os.path.join(HERE, "solve_restated.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("solve_restated.json:", {k: len(v["conv"]) for k, v in out.items()})


# --------------------------------------------------------------------------------------------------
# Stage (B) for the TIME-PARALLEL forward solve of the device path: the same reference-driven histories on
# hierarchies whose coarsest level has >= 64 steps, i.e. where the product's default forward_solve is the block form and no
# longer the reference's step-by-step loop (mgrit.py:459-486). The reference runs its own sequential forward_solve on the
# restated steps; the oracle's sequential form has to reproduce it to 1e-10 relative, the block form to
# 1e-10 * conv + K * eps * ||u|| (tests/test_oracle_golden.py states K).
# Same arithmetic as RestatedHeat1D, on Python floats (IEEE doubles, the same roundings as numpy's scalar operations) so that
# config 2's 80 000 steps of 1022 values run in minutes.
# --------------------------------------------------------------------------------------------------
class RestatedHeat1DFast(Heat1D):
    def step(self, u_start, t_start, t_stop):
        dt = t_stop - t_start
        beta, diag = dt * self.fac_, dt * (2.0 * self.fac_) + 1.0
        d = (u_start.get_values() + self.rhs(self.x, t_stop) * dt).tolist()
        n = len(d)
        cp = [0.0] * n
        piv = diag
        cp[0] = -beta / piv
        d[0] = d[0] / piv
        for j in range(1, n):
            piv = diag + beta * cp[j - 1]
            cp[j] = -beta / piv
            d[j] = (d[j] + beta * d[j - 1]) / piv
        out = [0.0] * n
        out[n - 1] = d[n - 1]
        for j in range(n - 2, -1, -1):
            out[j] = d[j] - cp[j] * out[j + 1]
        ret = VectorHeat1D(n)
        ret.set_values(np.array(out))
        return ret


class RestatedAdvection1D(Advection1D):
    """reference Advection1D with step = the oracle's variant-0 periodic solve (oracle/mgrit_oracle.c advection1d_step_natural:
    forward substitution of the bidiagonal part, closure through the last unknown), no SuperLU"""

    def step(self, u_start, t_start, t_stop):
        alpha = (t_stop - t_start) * self.fac_
        D = alpha + 1.0
        u = u_start.get_values().tolist()
        n = len(u)
        p, q = [0.0] * n, [0.0] * n
        p[0] = u[0] / D
        q[0] = alpha / D
        for j in range(1, n):
            p[j] = (u[j] + alpha * p[j - 1]) / D
            q[j] = alpha * q[j - 1] / D
        xl = p[n - 1] / (1.0 - q[n - 1])
        out = [p[j] + q[j] * xl for j in range(n - 1)] + [xl]
        ret = VectorAdvection1D(n)
        ret.set_values(np.array(out))
        return ret


def restated_advection_levels(nx, ts):
    out = []
    for t in ts:
        app = RestatedAdvection1D(c=1, x_start=-1, x_end=1, nx=nx, t_interval=t)
        app.fac_ = 1.0 / (app.x[1] - app.x[0])       # c / dx exactly as cases.advection_level_spec computes it
        out.append(app)
    return out


def make_restated_blk():
    """inputs restated in tests/cases.restated_blk_cases; about ten minutes (config 2's shape dominates)"""
    with open(os.path.join(HERE, "solve_restated.json")) as f:
        out = json.load(f)

    def lin(nts):
        return [np.linspace(0, 2, nt) for nt in nts]

    def heat(nx, ts):
        return restated_levels([nx] * len(ts), ts, cls=RestatedHeat1DFast)
    t0 = 2 * np.linspace(0, 1, 401) ** 1.3
    out["restated_blk_config2"] = run(heat(1024, lin([4097, 1025, 257])), sample_pts=(2048, 4096), cf_iter=1, max_iter=5,
                                      tol=1e-30)
    out["restated_blk_nx2050_2lvl"] = run(heat(2050, lin([513, 129])), sample_pts=(512,), tol=1e-30, max_iter=5, cf_iter=2)
    out["restated_blk_nonuniform_F"] = run(heat(129, [t0[::s] for s in (1, 2, 4)]), sample_pts=(400,), tol=1e-30, max_iter=6,
                                           cycle_type='F')
    out["restated_blk_nx257_3lvl"] = run(heat(257, lin([1025, 257, 65])), sample_pts=(1024,), tol=1e-30, max_iter=7)
    out["restated_blk_advection_F"] = run(restated_advection_levels(257, lin([1025, 257, 65])), sample_pts=(1024,),
                                          cycle_type='F', tol=1e-30, max_iter=6)
    out["restated_blk_advection_nonuniform"] = run(restated_advection_levels(129, [t0[::s] for s in (1, 2, 4)]),
                                                   sample_pts=(400,), tol=1e-30, max_iter=6, nested_iteration=False)
    with open(os.path.join(HERE, "solve_restated.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("solve_restated.json:", {k: v["conv"] for k, v in out.items() if k.startswith("restated_blk_")})


def ref_results():
    res = {}
    d = os.path.join(REF, "tests", "mpi", "results")
    for name in sorted(os.listdir(d)):
        with open(os.path.join(d, name)) as f:
            res[name] = [float(x) for x in f.read().split()]
    kats = {}
    return res, kats


def main():
    os.makedirs(HERE, exist_ok=True)
    if "--only-heat2d-general" in sys.argv:
        make_heat2d_general()
        sys.exit(0)
    if "--only-heat2d" in sys.argv:
        make_heat2d()
        return
    if "--only-advection-sc" in sys.argv:
        make_advection_sc()
        return
    if "--only-local-conv-ranks" in sys.argv:
        make_local_conv_ranks()
        return
    if "--only-solve-wide" in sys.argv:
        with open(os.path.join(HERE, "solve.json")) as f:
            sol = json.load(f)
        sol.update(make_solve_wide())
        with open(os.path.join(HERE, "solve.json"), "w") as f:
            json.dump(sol, f, separators=(",", ":"))
        return
    if "--only-adv-dft" in sys.argv:
        # round 5: Advection1D levels whose periodic grid is NOT a power of two (n = 200): the time-parallel forward solve takes its
        # transforms as ordered sums -- reference Mgrit with SuperLU into solve.json, on the restated steps into
        # solve_restated.json (inputs restated in tests/cases.block_cases / restated_blk_cases)
        ts = [np.linspace(0, 2, nt) for nt in (1025, 257, 65)]
        with open(os.path.join(HERE, "solve.json")) as f:
            sol = json.load(f)
        sol["advection_blk_nx201_3lvl_F"] = run([Advection1D(c=1, x_start=-1, x_end=1, nx=201, t_interval=t) for t in ts], cycle_type='F',
                                                max_iter=6, sample_pts=(512, 1024))
        with open(os.path.join(HERE, "solve.json"), "w") as f:
            json.dump(sol, f, separators=(",", ":"))
        with open(os.path.join(HERE, "solve_restated.json")) as f:
            out = json.load(f)
        out["restated_blk_advection_n200_F"] = run(restated_advection_levels(201, ts), sample_pts=(1024,), cycle_type='F', tol=1e-30,
                                                   max_iter=6)
        with open(os.path.join(HERE, "solve_restated.json"), "w") as f:
            json.dump(out, f, separators=(",", ":"))
        print(sol["advection_blk_nx201_3lvl_F"]["conv"], out["restated_blk_advection_n200_F"]["conv"])
        return
    if "--only-blk-wide-rank" in sys.argv:
        # round 5: a coarsest level whose blocks leave 127 sine modes above 2^-60 (the cap on the modes was 64 until then: DESIGN.md
        # 3.8) -- reference Mgrit with SuperLU into solve.json, reference Mgrit on the restated Thomas steps into solve_restated.json
        # (inputs restated in tests/cases.block_cases / restated_blk_cases)
        ts = [np.linspace(0, 0.02, nt) for nt in (1025, 257)]
        with open(os.path.join(HERE, "solve.json")) as f:
            sol = json.load(f)
        sol["heat_blk_r127_2lvl"] = run([Heat1D(x_start=0, x_end=1, nx=1025, a=1, init_cond=init_cond, rhs=rhs, t_interval=t) for t in ts],
                                        tol=1e-30, max_iter=4, sample_pts=(512, 1024))
        with open(os.path.join(HERE, "solve.json"), "w") as f:
            json.dump(sol, f, separators=(",", ":"))
        with open(os.path.join(HERE, "solve_restated.json")) as f:
            out = json.load(f)
        out["restated_blk_r127_2lvl"] = run(restated_levels([1025, 1025], ts, cls=RestatedHeat1DFast), sample_pts=(1024,), tol=1e-30,
                                            max_iter=5)
        with open(os.path.join(HERE, "solve_restated.json"), "w") as f:
            json.dump(out, f, separators=(",", ":"))
        print(sol["heat_blk_r127_2lvl"]["conv"], out["restated_blk_r127_2lvl"]["conv"])
        return
    if "--only-solve-block" in sys.argv:
        with open(os.path.join(HERE, "solve.json")) as f:
            sol = json.load(f)
        sol.update(make_solve_block())
        with open(os.path.join(HERE, "solve.json"), "w") as f:
            json.dump(sol, f, separators=(",", ":"))
        return
    if "--only-exchange-fuzz" in sys.argv:
        make_exchange_fuzz()
        return
    if "--only-at-mgrit" in sys.argv:
        make_at_mgrit()
        return
    if "--only-local-conv" in sys.argv:
        make_local_conv()
        return
    if "--only-bdf" in sys.argv:
        make_bdf()
        return
    if "--only-restated" in sys.argv:
        make_restated()
        make_restated_blk()
        return
    if "--only-restated-blk" in sys.argv:
        make_restated_blk()
        return
    big = "--small" not in sys.argv
    lay = make_layout()
    with open(os.path.join(HERE, "layout.json"), "w") as f:
        json.dump(lay, f, separators=(",", ":"))
    meta, arrays = make_phi()
    with open(os.path.join(HERE, "phi.json"), "w") as f:
        json.dump(meta, f, indent=1)
    np.savez_compressed(os.path.join(HERE, "phi.npz"), **arrays)
    sol = make_solve(big)
    with open(os.path.join(HERE, "solve.json"), "w") as f:
        json.dump(sol, f, separators=(",", ":"))
    make_heat2d()
    make_heat2d_general()
    make_advection_sc()
    make_bdf()
    make_local_conv()
    make_at_mgrit()
    make_local_conv_ranks()
    make_exchange_fuzz()
    make_restated()
    make_restated_blk()
    res, kats = ref_results()
    with open(os.path.join(HERE, "ref_results.json"), "w") as f:
        json.dump({"tests_mpi_results": res}, f, indent=1)
    print("wrote fixtures:", {k: os.path.getsize(os.path.join(HERE, k)) for k in
                              ("layout.json", "phi.json", "phi.npz", "solve.json", "ref_results.json")})


if __name__ == "__main__":
    main()
