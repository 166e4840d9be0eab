import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pymgrit_amd import Mgrit
from pymgrit_amd.heat.heat_2d import Heat2D
from scratch.gap import timeit
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 1025
t0 = np.linspace(0, 1, nt)
prob = [Heat2D(x_start=0, x_end=1, y_start=0, y_end=1, nx=nx, ny=nx, a=1.0, method="BE",
               init_cond=lambda x, y: np.sin(np.pi * x) * np.sin(np.pi * y), t_interval=t) for t in (t0, t0[::8])]
mg = Mgrit(prob, nested_iteration=False, max_iter=1, tol=0.0, logging_lvl=30)
ms = timeit(lambda: mg.f_relax(0), 3)
nphi = (nt - 1) * 7 // 8
M = ((nx - 2 + 63) // 64) * 64
flops = nphi * 4 * 2 * M ** 3
print(f"heat2d nx={nx} nt={nt}: f_relax {ms:.2f} ms, {nphi} Phi, {ms/nphi*1e3:.1f} us/Phi, GEMM rate {flops/ms/1e9:.1f} TFLOP/s (padded M={M})")
ms = timeit(lambda: (mg.iteration(0, 'V', 1, True), mg.convergence_criterion(1)), 2)
print(f"V-cycle + residual: {ms:.1f} ms")
