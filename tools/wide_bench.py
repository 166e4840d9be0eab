#!/usr/bin/env python3
"""Level-0 F-relaxation of a WIDE Heat1D state (n > 16384: three launches per Phi over rows in HBM, csrc/mgrit_hip_wide.inc):
device time per launch group and the HBM rate of the rows it moves by construction (per Phi: the local-scan launch reads u and
writes the scanned row, the finishing launch reads it back and writes the result: 4 rows of 8 n bytes; SURVEY 8d counts 2)."""
import json
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from pymgrit_amd import Heat1D, Mgrit
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 32770
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 4097
t0 = np.linspace(0, 2.0 * (nt - 1) / 65536, nt)
prob = [Heat1D(x_start=0, x_end=1, nx=nx, a=1, init_cond=bench.init_cond, rhs_separable=[(bench.rhs_space, bench.rhs_time)], t_interval=g)
        for g in (t0, t0[::4], t0[::16])]
mg = Mgrit(prob, cf_iter=1, nested_iteration=False, max_iter=3, tol=0.0, logging_lvl=30)
be = mg.backend
runs = mg._f_runs(0)
n_f = sum(r[1] for r in runs)
be.relax(0, runs, 'F'); be.sync()
be.set_timing(True); be.timing_drain()
for _ in range(5):
    be.relax(0, runs, 'F')
recs = be.timing_drain()
ms = float(np.mean([m for k, _, m in recs if k == "relax_f"][1:]))
dof = nx - 2
conv = mg.solve()["conv"]
out = {"workload": f"heat_1d nx={nx} nt={nt} 3-level m=4: level-0 F-relaxation, wide state ({-(-dof // 1024)} groups)", "f_relax_ms": ms,
       "phi": n_f, "algorithmic_GBps": n_f * 16.0 * dof / (ms * 1e-3) / 1e9, "rows_moved_GBps": n_f * 32.0 * dof / (ms * 1e-3) / 1e9,
       "rows_moved_frac_of_8TBps": n_f * 32.0 * dof / (ms * 1e-3) / 1e9 / 8000.0, "updates_per_s": n_f * dof / (ms * 1e-3),
       "solve_conv": [float(c) for c in conv]}
print(json.dumps(out))
