#!/usr/bin/env python3
"""Where the time of the level-0 way down goes, measured INSIDE cfas_kernel: an experiment build of the library in which the
`/*STAMP n*/` markers of pymgrit_amd/csrc/mgrit_hip.hip are wall-clock stamps (s_memrealtime, 100 MHz, thread 0 of every
workgroup, summed per phase) and `/*DRAIN*/` is a wait for every vector memory operation in flight. The product library has
neither (the markers are comments there). The stamps cost time of their own (cfas 1.8 -> 2.4 ms): read the SHARES.

    python tools/cfas_timeline.py build      # here (hipcc): scratch/cfas_timeline/libmgrit_hip_stamps.so
    gpurun -- 'python tools/cfas_timeline.py run > gpurun_out/cfas_timeline.txt'     # on the GPU box: config 3, program order
"""
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORK = os.path.join(ROOT, "scratch", "cfas_timeline")
LIB = os.path.join(WORK, "libmgrit_hip_stamps.so")
PHASES = ["coarse Phi + park", "F-steps of the interval", "request of the old last F-point's row", "its arrival + row stores of the C-point",
          "residual Phi (solve part)", "arithmetic + store of g", "chunk start (per chunk)", "tail", "DRAIN at the top of the interval",
          "own coefficient set back (load_coef)"]


def build():
    src = os.path.join(WORK, "src")
    shutil.rmtree(src, ignore_errors=True)
    shutil.copytree(os.path.join(ROOT, "pymgrit_amd", "csrc"), src)
    p = os.path.join(src, "mgrit_hip.hip")
    s = open(p).read()
    head = "template <int FORCE>\n__global__ void __launch_bounds__(1024) cfas_kernel("
    assert s.count(head) == 1
    s = s.replace(head, "__device__ unsigned long long g_stamp[16][256];\n"
                  "#define STAMP(k) do { if (t == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); "
                  "g_stamp[k][blockIdx.x & 255] += now_ - last_; last_ = now_; } } while (0)\n" + head)
    body = s.index(head)
    anchor = s.index("constexpr int KIND = MGRIT_HIP_STEPPER_HEAT1D;", body)
    s = s[:anchor] + "unsigned long long last_ = __builtin_amdgcn_s_memrealtime();\n    " + s[anchor:]
    s = re.sub(r"/\*STAMP (\d+)\*/", lambda m: f"STAMP({m.group(1)});", s)
    s = s.replace("/*DRAIN*/", "__builtin_amdgcn_s_waitcnt(0x0070);")
    s = s.rstrip() + '''

extern "C" int mgrit_hip_debug_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 16 * 256) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long z[16 * 256];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
'''
    open(p, "w").write(s)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "include"), "-shared", "-o", LIB, "mgrit_hip.hip"], cwd=src)
    print("built", LIB)


def run():
    os.environ["PYMGRIT_AMD_LIB"] = LIB
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import numpy as np
    import bench
    from pymgrit_amd import Heat1D, Mgrit
    from pymgrit_amd.core import hip_lib
    nx, nt0 = 16384, 65537
    t0 = np.linspace(0, 2.0, nt0)
    problem = [Heat1D(x_start=0, x_end=1, nx=nx, a=1, init_cond=bench.init_cond, rhs_separable=[(bench.rhs_space, bench.rhs_time)],
                      t_interval=g) for g in (t0, t0[::4], t0[::16])]
    mg = Mgrit(problem, cf_iter=1, cycle_type='V', nested_iteration=False, max_iter=3, tol=0.0, logging_lvl=30, plan_blocks=1)
    be, lib = mg.backend, hip_lib.load()

    def cycle(it):
        mg.iteration(lvl=0, cycle_type='V', iteration=it, first_f=True)
        mg.convergence_criterion(iteration=1)
    cycle(0); cycle(1); be.sync()
    buf = (C.c_ulonglong * (16 * 256))()
    assert lib.mgrit_hip_debug_stamps(buf, 1) == 0
    n = 5
    be.set_timing(True); be.timing_drain()
    for _ in range(n):
        cycle(1)
    recs = be.timing_drain(); be.set_timing(False); be.sync()
    assert lib.mgrit_hip_debug_stamps(buf, 1) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(16, 256).astype(np.float64) * 0.01    # us
    per = a / (n * 64.0)      # 64 intervals per workgroup and launch (16384 intervals, 256 workgroups)
    ms = [m for k, _, m in recs if k == "cf_fas"]
    print(f"cfas_kernel with stamps: {np.mean(ms):.3f} ms per launch (config 3, heat_1d nx=16384 nt=65537 m=4, steady-state cycle)")
    print("per interval of level 0 (3 F-points, the C-point that closes it), mean over the 256 workgroups [min .. max]:")
    for k, name in enumerate(PHASES):
        print(f"  {name:48s} {per[k].mean():6.2f} us  [{per[k].min():6.2f} .. {per[k].max():6.2f}]")
    print(f"  sum {per[:10].sum(axis=0).mean():.2f} us per interval")


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
