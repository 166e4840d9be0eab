"""Worker for the multi-process tests: one rank of a torch.distributed job running Mgrit on the plugin path (CPU, gloo)
or on the HIP path (GPU). Results go to a per-rank .npz so the parent can compare them with the single-rank run."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def build_problem(case_name, mode):
    import cases
    if case_name.startswith("h2d:"):
        prob, opts = cases.h2d_solve_problem(case_name[4:])
        return prob, None, opts
    if case_name.startswith("at:"):      # AT-MGRIT: (problem, transfer, options incl. the distance k under "_at_k")
        import test_at_mgrit as at
        nx, nts, k, opts = at.CASES[case_name[3:]]
        return at.heat(nx, nts, mode == "plugin"), None, dict(opts, _at_k=k)
    if case_name.startswith("lc:"):      # local stopping criteria on several ranks (tests/test_local_conv.py)
        import test_local_conv as lc
        make, opts = lc._ranks_problem(case_name[3:])
        prob = make()
        if mode != "plugin":
            for p in prob:
                del p.device_stepper      # back to the class's declarative description: HIP path
        return prob, None, dict(opts)
    if case_name.startswith("bdf:"):
        c = cases.BDF_CASES[case_name[4:]]
        prob = cases.bdf_levels(c["nx"], c["n_pairs"], c["orders"], c["coarsening"], c["forcing"])
        if mode == "plugin":
            for p in prob:
                p.device_stepper = lambda: None   # host steppers through the plugin backend
        return prob, None, dict(c["kw"])
    if case_name.startswith("advsc:"):
        from pymgrit_amd import Advection1D as A1, GridTransferAdvection as GA, GridTransferCopy as GC
        rec, nxs, ts, transfer, opts = cases.adv_sc_case(case_name[6:])
        return ([A1(c=1, x_start=-1, x_end=1, nx=nx, t_interval=t) for nx, t in zip(nxs, ts)],
                [GA() if k == 2 else GC() for k in transfer], opts)
    from pymgrit_amd import Advection1D, Dahlquist, GridTransferCopy, GridTransferHeat, Heat1D
    c = {**cases.solve_cases(), **cases.extra_cases()}[case_name]
    tr = None
    if c.get("transfer") is not None:
        tr = [GridTransferHeat() if k == 1 else GridTransferCopy() for k in c["transfer"]]
    prob = []
    for s in c["levels"]:
        if s["kind"] == "dahlquist":
            prob.append(Dahlquist(constant_lambda=s["lambda"], method=s["method"], t_interval=np.asarray(s["t"])))
        elif mode == "plugin":
            from oracle import oracle as orc
            from oracle_apps import OracleApp
            prob.append(OracleApp(orc, s))
        elif s["kind"] == "heat1d":
            nx = s["n"] + 2
            x_end = float(np.round((nx - 1) / np.sqrt(s["fac"]), 9))  # a = 1 in every fixture case
            kw = dict(rhs_separable=[(cases.rhs_space, cases.rhs_time)]) if s.get("s") is not None else {}
            prob.append(Heat1D(x_start=0, x_end=x_end, nx=nx, a=1, init_cond=cases.init_cond,
                               t_interval=np.asarray(s["t"]), **kw))
        else:
            prob.append(Advection1D(c=1, x_start=-1, x_end=1, nx=s["n"] + 1, t_interval=np.asarray(s["t"])))
    return prob, tr, dict(c["opts"])


def run(rank, world, port, case_name, mode, out_dir, backend):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1:
        dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    if mode == "hip":
        torch.cuda.set_device(rank % max(torch.cuda.device_count(), 1))
    from pymgrit_amd import AtMgrit, Mgrit
    prob, tr, opts = build_problem(case_name, mode)
    if "_at_k" in opts:
        k = opts.pop("_at_k")
        Mgrit = lambda *a, **kw: AtMgrit(k, 0 if "conv_crit" not in kw else kw.pop("conv_crit"), *a, **kw)  # noqa: E731
    if os.environ.get("MGRIT_TEST_PIPELINE_DEPTH") is not None:
        opts["pipeline_depth"] = int(os.environ["MGRIT_TEST_PIPELINE_DEPTH"])
    mg = Mgrit(prob, transfer=tr, logging_lvl=30, **opts)
    if world > 1 and os.environ.get("MGRIT_TEST_PIPELINE_DEPTH") is not None:
        assert mg.pipeline_depth() == int(os.environ["MGRIT_TEST_PIPELINE_DEPTH"]) or mg.conv_crit != 0
    slow = os.environ.get("MGRIT_TEST_SLOW_RANK")
    if slow is not None and world > 1 and rank == int(slow) % world:
        import time
        relax = mg.backend.relax

        def slow_relax(*a, **k):   # one rank far behind its neighbours: the others run ahead as far as the depth allows
            time.sleep(0.004)
            return relax(*a, **k)
        mg.backend.relax = slow_relax
    conv = mg.solve()["conv"]
    owned = [int(i) for i in mg.index_local[0]]
    vals = np.array([np.asarray(mg.u[0][i].pack(), dtype=np.float64).ravel() for i in owned])
    t_owned = np.asarray(mg.t[0])[owned]
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), conv=conv, u=vals, t=t_owned)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    run(rank, world, port, sys.argv[4], sys.argv[5], sys.argv[6], sys.argv[7])
