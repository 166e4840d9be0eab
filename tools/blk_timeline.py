#!/usr/bin/env python3
"""Where the time of the block solve's two passes goes on a RANK of a sharded level (32 blocks on 256 CUs), measured inside the
kernels: an experiment build in which the `/*BSTAMP n*/` markers of pymgrit_amd/csrc/mgrit_hip_blk.inc are wall-clock stamps
(s_memrealtime, 100 MHz, thread 0 of every workgroup, summed per phase). The product library has none (comments there).

    python tools/blk_timeline.py build
    gpurun -- 'python tools/blk_timeline.py run > gpurun_out/blk_timeline.txt'
"""
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORK = os.path.join(ROOT, "scratch", "blk_timeline")
LIB = os.path.join(WORK, "libmgrit_hip_stamps.so")
NAMES = {0: "finish: block start row + first g requested", 1: "finish: Phi", 2: "finish: add, next g requested, store issued", 3: "finish: tail",
         8: "one launch: prologue + pass 1 (16 Phi)", 9: "one launch: amplitudes (1024-long chains)", 10: "one launch: first device-wide barrier",
         11: "one launch: recurrence + block end", 12: "one launch: second device-wide barrier", 13: "one launch: pass 2 (15 Phi)",
         4: "local: prologue + first row", 5: "local: g requested + Phi", 6: "local: u_i requested, arrived, arithmetic", 7: "local: W_b stored, tail",
         }


def build():
    src = os.path.join(WORK, "src")
    shutil.rmtree(src, ignore_errors=True)
    shutil.copytree(os.path.join(ROOT, "pymgrit_amd", "csrc"), src)
    p = os.path.join(src, "mgrit_hip_blk.inc")
    s = open(p).read()
    s = ("__device__ unsigned long long g_bstamp[16][256];\n"
         "#define BSTAMP(k) do { if (t == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); "
         "g_bstamp[k][blockIdx.x & 255] += now_ - blast_; blast_ = now_; } } while (0)\n") + s
    s = s.replace("/*BSTAMP_INIT*/", "unsigned long long blast_ = __builtin_amdgcn_s_memrealtime();")
    s = re.sub(r"/\*BSTAMP (\d+)\*/", lambda m: f"BSTAMP({m.group(1)});", s)
    open(p, "w").write(s)
    q = os.path.join(src, "mgrit_hip.hip")
    m = open(q).read().rstrip() + '''

extern "C" int mgrit_hip_debug_bstamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bstamp), sizeof(unsigned long long) * 16 * 256) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long z[16 * 256];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_bstamp), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
'''
    open(q, "w").write(m)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "include"), "-shared", "-o", LIB, "mgrit_hip.hip"], cwd=src)
    print("built", LIB)


def run(small=False):
    os.environ["PYMGRIT_AMD_LIB"] = LIB
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import numpy as np
    import bench
    from pymgrit_amd import Heat1D, Mgrit
    from pymgrit_amd.core import hip_lib
    # a rank's share of config 3's coarsest level as a coarsest level of its own: 513 points, 32 blocks, same step size
    nx, nt0 = 16384, 8193
    t0 = np.linspace(0, 2.0 * (nt0 - 1) / 65536, nt0)
    if small:   # BASELINE config 2: the coarsest level (257 points of 1022 values) takes the one-launch form
        nx, nt0 = 1024, 4097
        t0 = np.linspace(0, 2.0 * (nt0 - 1) / 65536, nt0)
    problem = [Heat1D(x_start=0, x_end=1, nx=nx, a=1, init_cond=bench.init_cond, rhs_separable=[(bench.rhs_space, bench.rhs_time)],
                      t_interval=g) for g in (t0, t0[::4], t0[::16])]
    mg = Mgrit(problem, cf_iter=1, cycle_type='V', nested_iteration=False, max_iter=3, tol=0.0, logging_lvl=30, plan_blocks=1)
    be, lib = mg.backend, hip_lib.load()

    def cycle(it):
        mg.iteration(lvl=0, cycle_type='V', iteration=it, first_f=True)
        mg.convergence_criterion(iteration=1)
    cycle(0); cycle(1); be.sync()
    buf = (C.c_ulonglong * (16 * 256))()
    assert lib.mgrit_hip_debug_bstamps(buf, 1) == 0
    n = 10
    be.set_timing(True); be.timing_drain()
    for _ in range(n):
        cycle(1)
    recs = be.timing_drain(); be.set_timing(False); be.sync()
    assert lib.mgrit_hip_debug_bstamps(buf, 1) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(16, 256).astype(np.float64) * 0.01    # us
    B = (len(mg.t[2]) - 1) // 16
    per = a[:, :B] / n       # per launch and workgroup (one block each)
    ms = [m for k, lv, m in recs if k == "chain"]
    print(f"block solve with stamps: {np.mean(ms) * 1e3:.1f} us per solve ({B} blocks of 16 steps, n = {nx - 2}); per workgroup and launch, mean [min .. max]:")
    for k in sorted(NAMES):
        if per[k].sum() > 0:
            steps = 16 if k in (5, 6) else 15 if k in (1, 2) else 1
            print(f"  {NAMES[k]:52s} {per[k].mean():7.2f} us  [{per[k].min():7.2f} .. {per[k].max():7.2f}]   = {per[k].mean() / steps:5.2f} us x {steps}")


if __name__ == "__main__":
    {"build": build, "run": run, "run_small": lambda: run(True)}[sys.argv[1]]()
