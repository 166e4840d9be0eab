"""The C ABI of include/mgrit_hip.h driven directly -- raw device pointers, plain ints and doubles, the way the binding stub
of INTEGRATION.md does it, without the pymgrit_amd host layer: one F-relaxation + C-relaxation + residual of a Heat1D level
against the oracle, and the error behaviour of the entry points (negative return code + mgrit_hip_last_error, no exceptions,
no crash on misuse)."""
import ctypes as C

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


def _dev(t):
    return C.c_void_p(t.data_ptr())


@pytest.fixture()
def lib():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from pymgrit_amd.core import hip_lib
    return hip_lib.load()


def test_direct_abi_relaxation_matches_oracle(lib, oracle):
    nx, nt, m = 131, 33, 4
    x, _ = cases.heat_grid(nx)
    n = nx - 2
    t = cases.lin(2, nt)
    spec = cases.heat_level_spec(nx, t)
    ld = lib.mgrit_hip_row_stride(n)
    assert ld == 1024 and lib.mgrit_hip_row_position(n, 17) == ((((17 >> 10) * 8 + ((17 & 15) >> 1)) * 64 + ((17 >> 4) & 63)) << 1) + 1
    perm = np.array([lib.mgrit_hip_row_position(n, j) for j in range(n)])
    stream = torch.cuda.current_stream()
    eng = C.c_void_p()
    assert lib.mgrit_hip_create(C.byref(eng), 1, C.c_void_p(stream.cuda_stream)) == 0
    try:
        s = np.ascontiguousarray(np.asarray(spec["s"], dtype=np.float64).reshape(1, n))
        tau = np.ascontiguousarray(np.asarray(spec["tau"], dtype=np.float64).reshape(1, nt))
        tt = np.ascontiguousarray(t)
        assert lib.mgrit_hip_level_heat1d(eng, 0, nt, _ptr(tt), n, ld, float(spec["fac"]), 1, _ptr(s), _ptr(tau)) == 0
        rng = np.random.default_rng(3)
        u_host = rng.standard_normal((nt, n))
        slab = np.zeros((nt, ld))
        slab[:, perm] = u_host
        u = torch.from_numpy(slab).cuda()
        assert lib.mgrit_hip_level_bind(eng, 0, _dev(u), C.c_void_p(0), C.c_void_p(0)) == 0
        # F-intervals (3 points after every C-point), C-points, as run lists
        f_start = np.arange(1, nt, m, dtype=np.int32)
        f_len = np.full(f_start.size, m - 1, dtype=np.int32)
        c_start = np.arange(m, nt, m, dtype=np.int32)
        c_len = np.ones(c_start.size, dtype=np.int32)
        fid, cid = C.c_int(-1), C.c_int(-1)
        assert lib.mgrit_hip_runs_create(eng, 0, f_start.size, _ptr(f_start), _ptr(f_len), C.byref(fid)) == 0
        assert lib.mgrit_hip_runs_create(eng, 0, c_start.size, _ptr(c_start), _ptr(c_len), C.byref(cid)) == 0
        assert lib.mgrit_hip_relax(eng, 0, fid.value, 0, 1.0) == 0          # MGRIT_HIP_RELAX_F
        sumsq = torch.zeros(c_start.size, dtype=torch.float64, device="cuda")
        assert lib.mgrit_hip_residual(eng, 0, cid.value, _dev(sumsq)) == 0
        assert lib.mgrit_hip_relax(eng, 0, cid.value, 1, 1.0) == 0          # MGRIT_HIP_RELAX_C
        assert lib.mgrit_hip_sync(eng) == 0
        # oracle: same level as the fine level of a 2-level hierarchy with m = 4
        op = oracle.OracleProblem([spec, cases.heat_level_spec(nx, t[::m])], variant=1, nested_iteration=False)
        op.state("u", 0)[:] = u_host
        op.f_relax(0)
        ref_norms = op.residual_norms()
        op.c_relax(0)
        got = u.cpu().numpy()
        assert np.array_equal(got[:, perm], op.state("u", 0))
        pad = np.ones(ld, dtype=bool)
        pad[perm] = False
        assert np.isfinite(got[:, pad]).all()    # padding positions: unspecified finite values (the norms mask them)
        assert np.array_equal(np.sqrt(sumsq.cpu().numpy()), ref_norms)
    finally:
        assert lib.mgrit_hip_destroy(eng) == 0


def test_abi_misuse_returns_error_codes(lib):
    err = lambda: lib.mgrit_hip_last_error().decode()
    eng = C.c_void_p()
    assert lib.mgrit_hip_create(C.byref(eng), 2, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    try:
        t = np.ascontiguousarray(cases.lin(1, 9))
        n, ld = 100, lib.mgrit_hip_row_stride(100)
        rid = C.c_int(-1)
        assert lib.mgrit_hip_level_heat1d(eng, 5, 9, _ptr(t), n, ld, 1.0, 0, C.c_void_p(0), C.c_void_p(0)) < 0 and "level 5" in err()
        assert lib.mgrit_hip_relax(eng, 0, 0, 0, 1.0) < 0 and "no stepper" in err()
        assert lib.mgrit_hip_level_heat1d(eng, 0, 9, _ptr(t), n, ld + 16, 1.0, 0, C.c_void_p(0), C.c_void_p(0)) < 0 and "row_stride" in err()
        assert lib.mgrit_hip_level_heat1d(eng, 0, 9, _ptr(t), 70000, 70656, 1.0, 0, C.c_void_p(0), C.c_void_p(0)) < 0      # above the wide limit
        assert lib.mgrit_hip_level_advection1d(eng, 0, 9, _ptr(t), 70000, 70656, 1.0) < 0                                  # likewise for Advection1D
        assert lib.mgrit_hip_level_heat1d(eng, 0, 9, _ptr(t), n, ld, 1.0, 3, C.c_void_p(0), C.c_void_p(0)) < 0 and "forcing" in err()
        assert lib.mgrit_hip_level_heat1d(eng, 0, 9, _ptr(t), n, ld, 1.0, 0, C.c_void_p(0), C.c_void_p(0)) == 0
        assert lib.mgrit_hip_level_heat1d(eng, 0, 9, _ptr(t), n, ld, 1.0, 0, C.c_void_p(0), C.c_void_p(0)) < 0 and "already" in err()
        start, ln = np.array([1], dtype=np.int32), np.array([20], dtype=np.int32)
        assert lib.mgrit_hip_runs_create(eng, 0, 1, _ptr(start), _ptr(ln), C.byref(rid)) < 0     # run leaves the local grid
        start0 = np.array([0], dtype=np.int32)
        assert lib.mgrit_hip_runs_create(eng, 0, 1, _ptr(start0), _ptr(np.array([1], dtype=np.int32)), C.byref(rid)) < 0  # no predecessor
        ok_len = np.array([3], dtype=np.int32)
        assert lib.mgrit_hip_runs_create(eng, 0, 1, _ptr(start), _ptr(ok_len), C.byref(rid)) == 0
        assert lib.mgrit_hip_relax(eng, 0, rid.value, 0, 1.0) < 0 and "not bound" in err()       # slabs missing
        assert lib.mgrit_hip_relax(eng, 0, rid.value + 7, 0, 1.0) < 0                            # unknown list id
        assert lib.mgrit_hip_level_transfer(eng, 0, 0) < 0 and "describe level 1" in err()
        u = torch.zeros((9, ld), dtype=torch.float64, device="cuda")
        assert lib.mgrit_hip_level_bind(eng, 0, _dev(u), C.c_void_p(0), C.c_void_p(0)) == 0
        assert lib.mgrit_hip_relax(eng, 0, rid.value, 9, 1.0) < 0 and "mode" in err()
        assert lib.mgrit_hip_residual(eng, 0, rid.value, C.c_void_p(0)) < 0
        assert lib.mgrit_hip_relax(eng, 0, rid.value, 0, 1.0) == 0 and lib.mgrit_hip_sync(eng) == 0
    finally:
        assert lib.mgrit_hip_destroy(eng) == 0
    lib.mgrit_hip_destroy(C.c_void_p(0))   # a null engine must not crash
