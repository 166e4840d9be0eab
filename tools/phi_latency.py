#!/usr/bin/env python3
"""Time of one Phi of a 1024-thread workgroup when nothing else runs: mgrit_hip_relax mode F on ONE run of 4096 points (one
workgroup steps, one row store per step, no load), and on 256 such runs side by side (every CU busy). GPU box."""
import ctypes as C
import os
import sys
import time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pymgrit_amd.core import hip_lib
lib = hip_lib.load()
_ptr = lambda a: C.c_void_p(a.ctypes.data)


def run(n, K, n_runs, run_len=1024):
    nt = n_runs * run_len + 1
    eng = C.c_void_p()
    st = torch.cuda.current_stream()
    assert lib.mgrit_hip_create(C.byref(eng), 1, C.c_void_p(st.cuda_stream)) == 0
    t = np.ascontiguousarray(np.linspace(0, 2.0 * (nt - 1) / 65536, nt))
    ld = lib.mgrit_hip_row_stride(n)
    s = np.ascontiguousarray(np.random.rand(max(K, 1), n))
    tau = np.ascontiguousarray(np.random.rand(max(K, 1), nt))
    assert lib.mgrit_hip_level_heat1d(eng, 0, nt, _ptr(t), n, ld, float((n + 1) ** 2), K, _ptr(s), _ptr(tau)) == 0, lib.mgrit_hip_last_error()
    u = torch.zeros(nt, ld, dtype=torch.float64, device="cuda")
    u[:, :n] = 1.0
    assert lib.mgrit_hip_level_bind(eng, 0, C.c_void_p(u.data_ptr()), C.c_void_p(0), C.c_void_p(0)) == 0
    start = np.arange(1, nt, run_len, dtype=np.int32)
    ln = np.full(start.size, run_len, dtype=np.int32)
    rid = C.c_int(-1)
    assert lib.mgrit_hip_runs_create(eng, 0, start.size, _ptr(start), _ptr(ln), C.byref(rid)) == 0
    for _ in range(2):
        assert lib.mgrit_hip_relax(eng, 0, rid.value, 0, 1.0) == 0
    lib.mgrit_hip_sync(eng)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        lib.mgrit_hip_relax(eng, 0, rid.value, 0, 1.0)
    lib.mgrit_hip_sync(eng)
    el = (time.perf_counter() - t0) / reps
    lib.mgrit_hip_destroy(eng)
    return el / run_len * 1e6


for n in (16382, 8190, 4094, 1022):
    for K in (0, 1, 2):
        print(f"n={n:6d} forcing terms {K}: one workgroup {run(n, K, 1):6.2f} us per Phi+store, 256 workgroups {run(n, K, 256, 256):6.2f} us", flush=True)
