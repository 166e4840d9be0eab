import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pymgrit_amd import Heat1D, Mgrit
import bench
from scratch.gap import timeit
def one_level(nt, nx, forcing=True):
    t0 = np.linspace(0, 2.0 * (nt - 1) / 4096, nt)
    kw = dict(rhs_separable=[(bench.rhs_space, bench.rhs_time)]) if forcing else {}
    return Mgrit([Heat1D(x_start=0, x_end=1, nx=nx, a=1, init_cond=bench.init_cond, t_interval=t0, **kw)], nested_iteration=False, max_iter=1, logging_lvl=30)
for nx in ((16384, 8192, 4096, 1024) if __name__ == "__main__" else ()):
    for forcing in (True, False):
        mg = one_level(4097, nx, forcing)
        ms = timeit(lambda: mg.forward_solve(0), 3)
        print(f"chain no-g nx={nx} forcing={forcing}: {ms:.2f} ms = {ms/4096*1e3:.2f} us/step, store {nx*8/ (ms/4096*1e-3)/1e9:.1f} GB/s")
        del mg
