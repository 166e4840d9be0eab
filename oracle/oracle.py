"""ctypes front-end of the parity oracle (oracle/mgrit_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg.
The product package (pymgrit_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

E, LANES = 16, 64
GROUP = E * LANES


class LayoutInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_local", "ghost", "first_owned", "n_owned", "comm_front", "comm_back", "first_is_c_point",
        "first_is_f_point", "last_is_c_point", "last_is_f_point", "send_to", "get_from", "n_c", "n_f", "m")]


def build(force=False):
    """Compile liboracle.so with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "mgrit_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.orc_problem_create.restype = C.c_void_p
        L.orc_problem_create.argtypes = [C.c_int]
        L.orc_problem_destroy.argtypes = [C.c_void_p]
        L.orc_problem_set_level_heat1d.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, C.c_int, C.c_double, C.c_int,
                                                   dp, dp, dp, C.c_int]
        L.orc_problem_set_forcing_rows.argtypes = [C.c_void_p, C.c_int, dp]
        L.orc_problem_set_forcing_rows_2d.argtypes = [C.c_void_p, C.c_int, dp]
        L.orc_problem_set_level_heat1d_2pts.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, C.c_int, C.c_double, C.c_double,
                                                        C.c_int, C.c_int, dp, dp, dp, dp, C.c_int]
        L.orc_sumsq_spec_2pts.restype = C.c_double
        L.orc_sumsq_spec_2pts.argtypes = [dp, C.c_int]
        L.orc_problem_set_level_advection1d.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, C.c_int, C.c_double, dp,
                                                        C.c_int]
        L.orc_problem_set_level_heat2d.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, C.c_int, C.c_int, C.c_double, C.c_double,
                                                   C.c_double, dp, C.c_int, dp, dp, dp]
        L.orc_sumsq_rows.restype = C.c_double
        L.orc_sumsq_rows.argtypes = [dp, C.c_int, C.c_int]
        L.orc_problem_set_level_dahlquist.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, C.c_double, C.c_int,
                                                      C.c_double]
        L.orc_problem_set_transfer.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_problem_set_at.argtypes = [C.c_void_p, C.c_int]
        L.orc_problem_set_block_solve.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_h2d_threads.restype = C.c_int
        L.orc_set_h2d_threads.argtypes = [C.c_int]
        L.orc_block_solve_rank.restype = C.c_int
        L.orc_block_solve_rank.argtypes = [C.c_int, C.c_double, C.c_int, dp, dp]
        L.orc_at_forward_solve.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_problem_set_threads.restype = C.c_int
        L.orc_problem_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.orc_problem_set_options.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_double, C.c_int]
        L.orc_problem_init_state.argtypes = [C.c_void_p]
        L.orc_state_ptr.restype = dp
        L.orc_state_ptr.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_phi_count.restype = C.c_int64
        L.orc_phi_count.argtypes = [C.c_void_p, C.c_int]
        L.orc_phi.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, dp]
        for name in ("orc_f_relax", "orc_c_relax", "orc_forward_solve", "orc_fas_residual", "orc_error_correction"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int]
        L.orc_iteration.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_nested_iteration.argtypes = [C.c_void_p]
        L.orc_setup.argtypes = [C.c_void_p]
        L.orc_solve.restype = C.c_int
        L.orc_solve.argtypes = [C.c_void_p, dp]
        L.orc_compute_residual.restype = C.c_int
        L.orc_compute_residual.argtypes = [C.c_void_p, dp]
        L.orc_time_norm.restype = C.c_double
        L.orc_time_norm.argtypes = [dp, C.c_int, C.c_int]
        L.orc_sumsq_spec.restype = C.c_double
        L.orc_sumsq_spec.argtypes = [dp, C.c_int]
        L.orc_restrict.argtypes = [C.c_int, dp, C.c_int, dp, C.c_int]
        L.orc_interp.argtypes = [C.c_int, dp, C.c_int, dp, C.c_int]
        L.orc_split_into.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int32)]
        L.orc_layout.restype = C.c_int
        L.orc_layout.argtypes = [C.c_int, C.POINTER(C.c_int32), dp, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(LayoutInfo)] + [C.POINTER(C.c_int64)] * 4
        L.orc_cset_dump.restype = C.c_int
        L.orc_cset_dump.argtypes = [C.c_void_p, C.c_int, C.c_double, dp, dp]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def block_solve_rank(n, fac, t):
    """modes kept by the time-parallel forward solve of a Heat1D level (0: the level is solved step by step), DESIGN.md 3.8"""
    t = _f64(t)
    return int(lib().orc_block_solve_rank(int(n), float(fac), t.size, _dp(t), None))


def set_h2d_threads(threads):
    """OpenMP threads over the independent columns of the Heat2D sine transforms (same bits for every count); returns the count
    in effect. Process-wide."""
    return int(lib().orc_set_h2d_threads(int(threads)))


def split_into(n_points, n_procs):
    out = np.zeros(n_procs, dtype=np.int32)
    lib().orc_split_into(n_points, n_procs, out.ctypes.data_as(C.POINTER(C.c_int32)))
    return out


def layout(t_levels, lvl, rank, size):
    """Restated Mgrit.setup_points_and_comm_info for (rank, size) on level lvl. Returns dict."""
    nt = np.array([len(t) for t in t_levels], dtype=np.int32)
    t_all = _f64(np.concatenate([np.asarray(t, dtype=np.float64) for t in t_levels]))
    cap = int(nt[lvl]) + 1
    bufs = [np.zeros(cap, dtype=np.int64) for _ in range(4)]
    info = LayoutInfo()
    rc = lib().orc_layout(len(t_levels), nt.ctypes.data_as(C.POINTER(C.c_int32)), _dp(t_all), lvl, rank, size,
                          C.byref(info), *[b.ctypes.data_as(C.POINTER(C.c_int64)) for b in bufs])
    if rc != 0:
        raise RuntimeError("orc_layout failed")
    d = {n: getattr(info, n) for n, _ in LayoutInfo._fields_}
    d["cpts"] = bufs[0][:info.n_c].copy()
    d["index_local"] = bufs[1][:info.n_owned].copy()
    d["index_local_c"] = bufs[2][:info.n_c].copy()
    d["index_local_f"] = bufs[3][:info.n_f].copy()
    return d


def sumsq_spec(r):
    r = _f64(r)
    return lib().orc_sumsq_spec(_dp(r), r.size)


def sumsq_spec_2pts(r):
    r = _f64(np.asarray(r).ravel())
    return lib().orc_sumsq_spec_2pts(_dp(r), r.size // 2)


def restrict(kind, f, nc):
    f = _f64(f)
    c = np.zeros(nc)
    lib().orc_restrict(kind, _dp(f), f.size, _dp(c), nc)
    return c


def interp(kind, c, nf):
    c = _f64(c)
    f = np.zeros(nf)
    lib().orc_interp(kind, _dp(c), c.size, _dp(f), nf)
    return f


METHODS = {"BE": 0, "FE": 1, "TR": 2, "MR": 3}


class OracleProblem:
    """Single-rank MGRIT oracle over a level hierarchy. Level specs are dicts:
       {"kind": "heat1d", "t": array, "n": int, "fac": a/dx^2, "s": [K][n] or None, "tau": [K][nt] or None, "u0": [n]}
       {"kind": "heat1d_2pts", "t": array, "n": int (per time point of the pair), "fac", "dtau", "order": 1|2, "s", "tau",
        "tau2": [K][nt] tau_k(t_i + dtau), "u0": [2][n]}   (state = [first | second])
       {"kind": "advection1d", "t": array, "n": int, "fac": c/dx, "u0": [n]}
       {"kind": "dahlquist", "t": array, "lambda": float, "method": "BE", "u0": float}
       {"kind": "heat2d", "t": array, "nx", "ny", "fx", "fy", "theta", "bc": [nx][ny], "S": [K][nx-2][ny-2], "tau": [K][nt],
        "u0": [nx][ny]}
    """

    def __init__(self, levels, transfer=None, variant=1, weight_c=1.0, cf_iter=1, cycle_type='V', nested_iteration=True,
                 t_norm=2, conv_crit=0, max_iter=100, tol=1e-7, norm_spec=True, block_solve=True):
        L = lib()
        self.L = L
        self.n_levels = len(levels)
        self.h = C.c_void_p(L.orc_problem_create(self.n_levels))
        self.nt, self.n = [], []
        for lvl, s in enumerate(levels):
            t = _f64(s["t"])
            self.nt.append(t.size)
            if s["kind"] == "heat1d":
                n = int(s["n"])
                K = 0 if s.get("s") is None else int(np.asarray(s["s"]).reshape(-1, n).shape[0])
                sa = _f64(np.asarray(s["s"]).reshape(K, n)) if K else np.zeros(1)
                ta = _f64(np.asarray(s["tau"]).reshape(K, t.size)) if K else np.zeros(1)
                u0 = _f64(s["u0"])
                L.orc_problem_set_level_heat1d(self.h, lvl, t.size, _dp(t), n, float(s["fac"]), K, _dp(sa), _dp(ta),
                                               _dp(u0), int(variant))
                if s.get("b_rows") is not None:   # general forcing: [nt][n] rows rhs(x, t_i) * dt_i
                    rows = _f64(np.asarray(s["b_rows"]).reshape(t.size, n))
                    L.orc_problem_set_forcing_rows(self.h, lvl, _dp(rows))
            elif s["kind"] == "heat1d_2pts":
                m = int(s["n"])
                n = 2 * m
                K = 0 if s.get("s") is None else int(np.asarray(s["s"]).reshape(-1, m).shape[0])
                sa = _f64(np.asarray(s["s"]).reshape(K, m)) if K else np.zeros(1)
                ta = _f64(np.asarray(s["tau"]).reshape(K, t.size)) if K else np.zeros(1)
                tb = _f64(np.asarray(s["tau2"]).reshape(K, t.size)) if K else np.zeros(1)
                u0 = _f64(np.asarray(s["u0"]).ravel())
                assert u0.size == n
                L.orc_problem_set_level_heat1d_2pts(self.h, lvl, t.size, _dp(t), m, float(s["fac"]), float(s["dtau"]),
                                                    int(s["order"]), K, _dp(sa), _dp(ta), _dp(tb), _dp(u0), int(variant))
            elif s["kind"] == "advection1d":
                n = int(s["n"])
                u0 = _f64(s["u0"])
                L.orc_problem_set_level_advection1d(self.h, lvl, t.size, _dp(t), n, float(s["fac"]), _dp(u0),
                                                    int(variant))
            elif s["kind"] == "heat2d":
                nx, ny = int(s["nx"]), int(s["ny"])
                n = nx * ny
                Sx = np.asarray(s.get("S", np.zeros((0, nx - 2, ny - 2))), dtype=np.float64).reshape(-1, nx - 2, ny - 2)
                K = Sx.shape[0]
                sa = _f64(Sx) if K else np.zeros(1)
                ta = _f64(np.asarray(s["tau"]).reshape(K, t.size)) if K else np.zeros(1)
                bc, u0 = _f64(s["bc"]), _f64(np.asarray(s["u0"]).ravel())
                L.orc_problem_set_level_heat2d(self.h, lvl, t.size, _dp(t), nx, ny, float(s["fx"]), float(s["fy"]),
                                               float(s["theta"]), _dp(bc), K, _dp(sa), _dp(ta), _dp(u0))
                if s.get("b_rows") is not None:   # general forcing: [nt][nx-2][ny-2] values rhs(x, y, t_i)
                    rows = _f64(np.asarray(s["b_rows"]).reshape(t.size, (nx - 2) * (ny - 2)))
                    L.orc_problem_set_forcing_rows_2d(self.h, lvl, _dp(rows))
            elif s["kind"] == "dahlquist":
                n = 1
                L.orc_problem_set_level_dahlquist(self.h, lvl, t.size, _dp(t), float(s["lambda"]),
                                                  METHODS[s.get("method", "BE")], float(s.get("u0", 1.0)))
            else:
                raise ValueError(s["kind"])
            self.n.append(n)
        if transfer is not None:
            for lvl, k in enumerate(transfer):
                L.orc_problem_set_transfer(self.h, lvl, int(k))
        if isinstance(cf_iter, int):
            cf_iter = [cf_iter] * self.n_levels
        cf = np.array(list(cf_iter) + [1] * (self.n_levels - len(cf_iter)), dtype=np.int32)
        self.max_iter = max_iter
        L.orc_problem_set_options(self.h, float(weight_c), cf.ctypes.data_as(C.POINTER(C.c_int32)),
                                  1 if cycle_type == 'F' else 0, int(bool(nested_iteration)), int(t_norm),
                                  int(conv_crit), int(max_iter), float(tol), int(bool(norm_spec)))
        L.orc_problem_set_block_solve(self.h, int(bool(block_solve)))   # time-parallel forward solve where eligible
        L.orc_problem_init_state(self.h)

    def __del__(self):
        try:
            self.L.orc_problem_destroy(self.h)
        except Exception:
            pass

    def set_at(self, k):
        """AT-MGRIT: truncated coarsest-level solves of distance k (core/at_mgrit.py)"""
        self.L.orc_problem_set_at(self.h, int(k))

    def at_forward_solve(self, lvl, k): self.L.orc_at_forward_solve(self.h, lvl, int(k))

    def set_threads(self, threads):
        """threaded sweeps for timing (bench.py cpu_baseline): returns the thread count in effect"""
        return int(self.L.orc_problem_set_threads(self.h, int(threads)))

    def state(self, which, lvl):
        """numpy view (no copy) of slab 'u' | 'v' | 'g' on level lvl, shape [nt][n]."""
        idx = {"u": 0, "v": 1, "g": 2}[which]
        p = self.L.orc_state_ptr(self.h, idx, lvl)
        return np.ctypeslib.as_array(p, shape=(self.nt[lvl], self.n[lvl]))

    def phi(self, lvl, i, u):
        u = _f64(u)
        out = np.zeros(self.n[lvl])
        self.L.orc_phi(self.h, lvl, i, _dp(u), _dp(out))
        return out

    def phi_count(self, lvl):
        return int(self.L.orc_phi_count(self.h, lvl))

    def f_relax(self, lvl): self.L.orc_f_relax(self.h, lvl)
    def c_relax(self, lvl): self.L.orc_c_relax(self.h, lvl)
    def forward_solve(self, lvl): self.L.orc_forward_solve(self.h, lvl)
    def fas_residual(self, lvl): self.L.orc_fas_residual(self.h, lvl)
    def error_correction(self, lvl): self.L.orc_error_correction(self.h, lvl)

    def iteration(self, lvl, cycle_type, iteration, first_f):
        self.L.orc_iteration(self.h, lvl, 1 if cycle_type == 'F' else 0, iteration, int(first_f))

    def nested_iteration(self): self.L.orc_nested_iteration(self.h)
    def setup(self): self.L.orc_setup(self.h)

    def residual_norms(self):
        out = np.zeros(self.nt[0])
        cnt = self.L.orc_compute_residual(self.h, _dp(out))
        return out[:cnt].copy()

    def solve(self):
        """setup (nested iteration) + MGRIT iterations; returns conv history like Mgrit.solve()['conv']."""
        self.setup()
        conv = np.zeros(max(self.max_iter, 1))
        n = self.L.orc_solve(self.h, _dp(conv))
        c = conv[:n]
        return c[c != 0]

    def cset(self, lvl, dt):
        sc = np.zeros(4 + 17 + 6 + 64)
        tab = np.zeros(self.n[lvl])
        k = self.L.orc_cset_dump(self.h, lvl, float(dt), _dp(sc), _dp(tab))
        assert k == sc.size
        return {"rho": sc[0], "ik": sc[1], "scal": sc[2], "gc": sc[3], "pw": sc[4:21].copy(), "sc": sc[21:27].copy(),
                "lp": sc[27:].copy(), "tab": tab}
