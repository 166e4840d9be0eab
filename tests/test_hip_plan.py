"""GPU: the planned cycle (core/cycle_plan.py: sweeps and coarsest-level chain parts of different blocks of time points on two
streams) against the program order and against the oracle. Everything must be bit-identical: the plan launches the same
kernels with the same arguments, only in another order and with the chain cut into parts that hand their state on."""
import numpy as np
import pytest

import cases
import dist_worker

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def solve(case, blocks, **extra):
    from pymgrit_amd import Mgrit
    prob, tr, opts = dist_worker.build_problem(case, "hip")
    opts.update(extra)
    mg = Mgrit(prob, transfer=tr, logging_lvl=30, plan_blocks=blocks, **opts)
    conv = mg.solve()["conv"]
    return conv, [mg.backend.natural("u", lvl) for lvl in range(mg.lvl_max)], mg


CASES = ["heat_nx33_V_nested", "heat_nx33_F_weight13_cf2", "heat_nx33_V_cf0", "heat_nx257_nt257", "heat_spatial_coarsening",
         "heat_spatial_coarsening_F", "advection_3lvl_F", "heat_nx2050_wide", "heat_nx1500_wide_F", "heat_nx3100_wide_2lvl",
         "heat_config2"]


@pytest.mark.parametrize("case", CASES)
def test_planned_cycle_bit_identical(case):
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    conv0, u0, _ = solve(case, 1)
    for blocks in (2, 4, 8):
        conv, u, mg = solve(case, blocks)
        if len(mg.t[-1]) >= 3:
            assert any(p is not None and p.n_blocks > 1 for p in mg._plans.values()), "no plan was recorded"
        assert np.array_equal(conv, conv0), (case, blocks, conv, conv0)
        for a, b in zip(u, u0):
            assert np.array_equal(a, b), (case, blocks)


@pytest.mark.parametrize("merge", [None, "2", "5"])
def test_merged_way_up_bit_identical(monkeypatch, merge):
    """eight blocks: the way up of the first blocks is ONE launch per level (cycle_plan.Recorder.up_merge: half of the blocks
    by default from five blocks on, with the coarser levels' way up through the interval pass); same numbers as program order"""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    conv0, u0, _ = solve("heat_config2", 1)
    if merge is None:
        monkeypatch.delenv("PYMGRIT_AMD_PLAN_UP_MERGE", raising=False)
    else:
        monkeypatch.setenv("PYMGRIT_AMD_PLAN_UP_MERGE", merge)
    conv, u, mg = solve("heat_config2", 8)
    plan = next(p for p in mg._plans.values() if p is not None and p.n_blocks > 1)
    want = plan.n_blocks - (int(merge) if merge else plan.n_blocks // 2) + 1
    for lvl in (0, 1):
        ups = [n for n in plan.nodes if n.name == "ec_relax_res" and n.lvl == lvl]
        assert len(ups) == want, (lvl, [n.chunk for n in ups])
    assert np.array_equal(conv, conv0)
    for a, b in zip(u, u0):
        assert np.array_equal(a, b)


def test_planned_wide_chain_against_oracle(oracle, sequential_coarse):
    """wide states, many coarsest points, the step-by-step forward solve: the chain parts continue each other through the hand-over
    state; residual history equal to the oracle's (same arithmetic spec) and the state bit-exact"""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from pymgrit_amd import Heat1D, Mgrit
    nx, nts = 2050, (1025, 257, 65)
    grids = [cases.lin(2, nt) for nt in nts]
    prob = [Heat1D(x_start=0, x_end=1, nx=nx, a=1, init_cond=cases.init_cond, rhs_separable=[(cases.rhs_space, cases.rhs_time)],
                   t_interval=g) for g in grids]
    mg = Mgrit(prob, logging_lvl=30, plan_blocks=8, max_iter=3, tol=0.0)
    conv = mg.solve()["conv"]
    plan = next(p for p in mg._plans.values() if p is not None)
    assert sum(1 for n in plan.order if n.stream == "chain") == 8
    op = oracle.OracleProblem([cases.heat_level_spec(nx, g) for g in grids], variant=1, max_iter=3, tol=0.0, block_solve=False)
    ref = op.solve()
    assert np.max(np.abs(conv - ref) / ref) <= 1e-10, (conv, ref)
    assert np.array_equal(mg.backend.natural("u", 0), op.state("u", 0))


def test_debug_lines_carry_device_times(caplog):
    """logging.DEBUG: the reference's per-sweep lines (mgrit.py:333,370,486,549), here with the device time of the kernels"""
    import logging
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from pymgrit_amd import Mgrit
    prob, tr, opts = dist_worker.build_problem("heat_nx257_nt257", "hip")
    opts.update(max_iter=1)
    with caplog.at_level(logging.DEBUG):
        Mgrit(prob, transfer=tr, logging_lvl=logging.DEBUG, **opts).solve()
    lines = [r.getMessage() for r in caplog.records]
    for what, kind in (("F-relax", "relax_f"), ("C-relax", "relax_c"), ("Fas residual", "fas_fused"), ("Forward solve", "chain"),
                       ("Convergence criterion", "residual")):
        hit = [ln for ln in lines if ln.startswith(what) and "| device:" in ln and kind in ln]
        assert hit, (what, lines[:20])
        ms = float(hit[0].split(kind)[1].split()[1])
        assert 0.0 < ms < 100.0
