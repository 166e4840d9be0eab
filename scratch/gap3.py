import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from scratch.gap import build, timeit
from pymgrit_amd.core.hip_lib import check
mg = build(65537)
be = mg.backend
mg.iteration(0,'V',0,True); torch.cuda.synchronize()
pts = mg._c_points(0)
buf = torch.zeros(len(pts), dtype=torch.float64, device='cuda')
rid = be._point_run_id(0, pts)
host = np.empty(len(pts))
def resid_dev():
    check(be.lib.mgrit_hip_residual(be.h, 0, rid, C.c_void_p(buf.data_ptr())))
def resid_host():
    check(be.lib.mgrit_hip_residual_host(be.h, 0, rid, C.c_void_p(host.ctypes.data)))
def t(label, fn, n=3):
    print(label, "%.2f ms" % timeit(fn, n), flush=True)
t("cycle + resid_dev + sync + buf.cpu()", lambda: (mg.iteration(0,'V',1,True), resid_dev(), torch.cuda.synchronize(), buf.cpu()))
t("cycle + resid_host", lambda: (mg.iteration(0,'V',1,True), resid_host()))
t("cycle + sync + resid_host", lambda: (mg.iteration(0,'V',1,True), torch.cuda.synchronize(), resid_host()))
t("cycle + compute_residual", lambda: (mg.iteration(0,'V',1,True), mg.compute_residual()))
t("cycle + convergence_criterion", lambda: (mg.iteration(0,'V',1,True), mg.convergence_criterion(1)))
for _ in range(3):
    t0=time.perf_counter(); mg.iteration(0,'V',1,True); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter(); resid_host(); t3=time.perf_counter()
    print("enqueue %.2f sync %.2f resid_host %.2f"%((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3))
for _ in range(3):
    t0=time.perf_counter(); mg.iteration(0,'V',1,True); t1=time.perf_counter(); resid_host(); t3=time.perf_counter()
    print("enqueue %.2f resid_host(no sync before) %.2f"%((t1-t0)*1e3,(t3-t1)*1e3))
