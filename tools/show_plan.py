#!/usr/bin/env python3
"""Print the planned cycle of the bench workload (GPU box): issue order, stream, estimated start, predecessors."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from pymgrit_amd import Heat1D, Mgrit
nx, nt0 = 16384, int(sys.argv[2]) if len(sys.argv) > 2 else 65537
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 4
t0 = np.linspace(0, 2.0 * (nt0 - 1) / 65536, nt0)
problem = [Heat1D(x_start=0, x_end=1, nx=nx, a=1, init_cond=bench.init_cond, rhs_separable=[(bench.rhs_space, bench.rhs_time)],
                  t_interval=g) for g in (t0, t0[::4], t0[::16])]
mg = Mgrit(problem, cf_iter=1, cycle_type='V', nested_iteration=False, max_iter=3, tol=0.0, logging_lvl=30, plan_blocks=blocks)
mg.iteration(lvl=0, cycle_type='V', iteration=0, first_f=True)
mg.iteration(lvl=0, cycle_type='V', iteration=1, first_f=True)
plan = [p for k, p in mg._plans.items() if p is not None and not k[1]][0]
for n in plan.order:
    print(f"{n.start*1e3:7.3f}-{n.finish*1e3:7.3f} {n.stream:5s} #{n.idx:3d} {n.name:14s} L{n.lvl} b{n.chunk}  preds {sorted(plan.nodes[p].idx for p in n.preds)}")
