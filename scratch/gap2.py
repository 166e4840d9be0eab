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
def resid_dev():
    check(be.lib.mgrit_hip_residual(be.h, 0, rid, C.c_void_p(buf.data_ptr())))
def t(label, fn, n=3):
    print(label, "%.2f ms" % timeit(fn, n), flush=True)
t("cycle + resid kernel (device out) + sync", lambda: (mg.iteration(0,'V',1,True), resid_dev(), torch.cuda.synchronize()))
t("cycle + sync + resid + sync", lambda: (mg.iteration(0,'V',1,True), torch.cuda.synchronize(), resid_dev(), torch.cuda.synchronize()))
t("cycle + f_relax + sync", lambda: (mg.iteration(0,'V',1,True), mg.f_relax(0), torch.cuda.synchronize()))
t("cycle + c_relax + sync", lambda: (mg.iteration(0,'V',1,True), mg.c_relax(0), torch.cuda.synchronize()))
t("chain + resid + sync", lambda: (mg.forward_solve(2), resid_dev(), torch.cuda.synchronize()))
t("chain + f_relax0 + resid + sync", lambda: (mg.forward_solve(2), mg.f_relax(0), resid_dev(), torch.cuda.synchronize()))
t("f_relax0 + resid + sync", lambda: (mg.f_relax(0), resid_dev(), torch.cuda.synchronize()))
t("fas0 + resid + sync", lambda: (mg.fas_residual(0), resid_dev(), torch.cuda.synchronize()))
t("ec0 + resid + sync", lambda: (mg.error_correction(0), resid_dev(), torch.cuda.synchronize()))
t("ec0 + f_relax+ resid + sync", lambda: (mg.error_correction(0), mg.f_relax(0), resid_dev(), torch.cuda.synchronize()))
t("resid x1 + sync", lambda: (resid_dev(), torch.cuda.synchronize()))
