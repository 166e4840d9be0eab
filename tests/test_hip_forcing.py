"""GPU: a general (non-separable) forcing on the HIP engine -- an unmodified reference-style ``Heat1D(rhs=lambda x, t: ...)``
hierarchy runs from precomputed rows rhs(x, t_i)*dt_i (FORCE == 3, include/mgrit_hip.h: mgrit_hip_level_forcing_rows).
Against the oracle (same rows, same arithmetic spec: bit-exact) and against the reference's own run (solve_restated.json)."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
RESTATED = cases.load_json("solve_restated.json")


def problem(nx, grids):
    from pymgrit_amd import Heat1D
    return [Heat1D(x_start=0, x_end=1, nx=nx, a=1, init_cond=cases.init_cond, rhs=cases.general_rhs, t_interval=np.asarray(t))
            for t in grids]


def test_general_forcing_solve(oracle):
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from pymgrit_amd import Mgrit
    c = cases.restated_cases()["superlu_general_forcing"]
    grids = [s["t"] for s in c["levels"]]
    prob = problem(65, grids)
    assert prob[0].device_stepper()["forcing_rows"] is not None          # detected as not separable
    mg = Mgrit(prob, logging_lvl=30, **c["opts"])
    assert mg.backend.name == "hip"
    conv = mg.solve()["conv"]
    op = oracle.OracleProblem(c["levels"], variant=1, **c["opts"])
    oconv = op.solve()
    assert len(conv) == len(oconv) and np.max(np.abs(conv - oconv) / oconv) <= 1e-10, (conv, oconv)
    assert np.array_equal(mg.backend.natural("u", 0), op.state("u", 0))
    ref = np.array(RESTATED["superlu_general_forcing"]["conv"])
    assert np.all(np.abs(conv - ref) <= 1e-9 * ref + 2e-11), (conv, ref)
    for key, vals in RESTATED["superlu_general_forcing"]["samples"].items():
        got = np.asarray(mg.u[0][int(key)].get_values())
        assert np.max(np.abs(got - np.array(vals))) <= 1e-10 * max(1.0, np.max(np.abs(vals)))


@pytest.mark.parametrize("nx,cycle", [(1500, 'V'), (2050, 'F'), (300, 'V')])
def test_general_forcing_wide_and_sweeps(oracle, nx, cycle):
    """wider states (several groups: the plain cross-workgroup chain with a forcing row per step), F-cycle, every sweep"""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from pymgrit_amd import Mgrit
    grids = [cases.lin(2, 65), cases.lin(2, 17), cases.lin(2, 5)]
    mg = Mgrit(problem(nx, grids), logging_lvl=30, max_iter=3, tol=0.0, cycle_type=cycle)
    conv = mg.solve()["conv"]
    op = oracle.OracleProblem([cases.heat_level_spec_general(nx, t) for t in grids], variant=1, max_iter=3, tol=0.0,
                              cycle_type=cycle)
    oconv = op.solve()
    assert np.max(np.abs(conv - oconv) / oconv) <= 1e-10, (conv, oconv)
    for lvl in range(3):
        assert np.array_equal(mg.backend.natural("u", lvl), op.state("u", lvl)), lvl


def test_replaced_rhs_is_used():
    """a caller that replaces .rhs after construction gets the new forcing on the device too (never the stale declaration)"""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from pymgrit_amd import Heat1D, Mgrit
    grids = [cases.lin(2, 33), cases.lin(2, 9)]
    a = [Heat1D(x_start=0, x_end=1, nx=65, a=1, init_cond=cases.init_cond, rhs_separable=[(cases.rhs_space, cases.rhs_time)],
                t_interval=t) for t in grids]
    for p in a:
        p.rhs = cases.general_rhs
    conv_a = Mgrit(a, logging_lvl=30, max_iter=3, tol=0.0).solve()["conv"]
    conv_b = Mgrit(problem(65, grids), logging_lvl=30, max_iter=3, tol=0.0).solve()["conv"]
    assert np.array_equal(conv_a, conv_b)
