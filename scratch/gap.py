import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pymgrit_amd import Heat1D, Mgrit
import bench
def build(nt0, nx=16384):
    t0 = np.linspace(0, 2.0 * (nt0 - 1) / 65536, nt0)
    grids = [t0, t0[::4], t0[::16]]
    problem = [Heat1D(x_start=0, x_end=1, nx=nx, a=1, init_cond=bench.init_cond, rhs_separable=[(bench.rhs_space, bench.rhs_time)], t_interval=g) for g in grids]
    return Mgrit(problem, nested_iteration=False, max_iter=1, tol=0.0, logging_lvl=30)
def timeit(fn, n=4):
    fn(); torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter()-t)/n*1e3
for nt in ([int(a) for a in sys.argv[1:]] if __name__ == "__main__" else []):
    mg = build(nt)
    mg.iteration(0,'V',0,True); torch.cuda.synchronize()
    full = timeit(lambda: (mg.iteration(0,'V',1,True), mg.convergence_criterion(1)))
    nocheck = timeit(lambda: mg.iteration(0,'V',1,True))
    sync_each = timeit(lambda: (mg.iteration(0,'V',1,True), torch.cuda.synchronize()))
    chain = timeit(lambda: mg.forward_solve(2))
    chain_sync = timeit(lambda: (mg.forward_solve(2), torch.cuda.synchronize()))
    f0 = timeit(lambda: mg.f_relax(0))
    resid = timeit(lambda: mg.convergence_criterion(1))
    print(f"nt={nt}: cycle+check {full:.2f} | cycle only (async) {nocheck:.2f} | cycle+sync {sync_each:.2f} | chain {chain:.2f} | chain+sync {chain_sync:.2f} | f_relax0 {f0:.2f} | check alone {resid:.2f}", flush=True)
    del mg; torch.cuda.empty_cache()
