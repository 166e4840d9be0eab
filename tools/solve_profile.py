#!/usr/bin/env python3
"""Where the host time of Mgrit(...).solve() goes (config 3 by default): cProfile of the constructor and of solve() of a WARM run
(the second of two; slabs from the allocator's cache), top functions by cumulative and by own time, beside the wall clock and the
solver's own breakdown (Mgrit.solve_breakdown).   python tools/solve_profile.py [--nx 16384 --nt 65537] [--emulate-rank r/P]"""
import argparse
import cProfile
import gc
import io
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=16384)
    ap.add_argument("--nt", type=int, default=65537)
    ap.add_argument("--top", type=int, default=28)
    ap.add_argument("--no-profile", action="store_true")
    args = ap.parse_args()
    import torch
    import bench
    from pymgrit_amd import Heat1D, Mgrit
    nt0 = args.nt
    t0 = np.linspace(0, 2.0 * (nt0 - 1) / 65536, nt0)
    problem = [Heat1D(x_start=0, x_end=1, nx=args.nx, a=1, init_cond=bench.init_cond, rhs_separable=[(bench.rhs_space, bench.rhs_time)],
                      t_interval=g) for g in (t0, t0[::4], t0[::16])]
    for run in range(3):
        prof = cProfile.Profile()
        torch.cuda.synchronize()
        w0 = time.perf_counter()
        if run == 2 and not args.no_profile:
            prof.enable()
        mg = Mgrit(problem, cf_iter=1, cycle_type='V', nested_iteration=True, max_iter=30, tol=1e-10, logging_lvl=30)
        w1 = time.perf_counter()
        res = mg.solve()
        torch.cuda.synchronize()
        w2 = time.perf_counter()
        if run == 2 and not args.no_profile:
            prof.disable()
        print(f"run {run}: setup {1e3 * (w1 - w0):.2f} ms  solve {1e3 * (w2 - w1):.2f} ms  iterations {len(res['conv'])}  "
              f"breakdown {getattr(mg, 'solve_breakdown', None)}", flush=True)
        if run == 2 and not args.no_profile:
            for key in ("cumulative", "tottime"):
                s = io.StringIO()
                pstats.Stats(prof, stream=s).sort_stats(key).print_stats(args.top)
                print(s.getvalue())
        del mg
        gc.collect()


if __name__ == "__main__":
    main()
