"""output_fcn on the HIP path: the reference's post-processing idiom -- per-rank ``np.save`` of
``[[t_i, u_i] for i in index_local[0]]`` with ``u_i`` the application's Vector (examples/example_heat_2d.py:26-33) -- runs
unchanged on lazily materialised slab rows."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("family", ["heat1d", "heat2d", "bdf"])
def test_reference_output_function_runs_unchanged(tmp_path, family):
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from pymgrit_amd import Heat1D, Mgrit
    calls = []

    def output_fcn(self):   # the body of examples/example_heat_2d.py:26-33, path aside
        np.save(str(tmp_path / f"sol-rank{self.comm_time_rank}"),
                np.array([[self.t[0][i], self.u[0][i]] for i in self.index_local[0]], dtype=object))
        calls.append(self.solve_iter)

    if family == "heat1d":
        prob = [Heat1D(x_start=0, x_end=1, nx=65, a=1, init_cond=cases.init_cond, rhs_separable=[(cases.rhs_space, cases.rhs_time)],
                       t_interval=t) for t in (cases.lin(2, 33), cases.lin(2, 9))]
    elif family == "heat2d":
        prob, _ = cases.h2d_solve_problem("cn_2lvl")
    else:
        prob = cases.bdf_levels(35, 17, [2, 1], 2, "one")
    mg = Mgrit(prob, logging_lvl=30, max_iter=3, tol=1e-30, output_fcn=output_fcn, output_lvl=2)
    assert mg.backend.name == "hip"
    mg.solve()
    assert len(calls) >= 3   # output_lvl 2: after every iteration (mgrit.py:609-610)
    saved = np.load(str(tmp_path / "sol-rank0.npy"), allow_pickle=True)
    assert saved.shape == (len(mg.t[0]), 2)
    nat = mg.backend.natural("u", 0)
    for i in (0, 1, len(mg.t[0]) // 2, len(mg.t[0]) - 1):
        assert saved[i][0] == mg.t[0][i]
        vec = saved[i][1]
        assert type(vec) is type(prob[0].vector_template)
        assert np.array_equal(np.asarray(vec.pack(), dtype=np.float64).ravel(), nat[i])


def test_output_after_every_iteration_sees_every_f_point(monkeypatch):
    """output_lvl 2 on the whole-level passes: the way up stores C-points and the last F-point of every interval only
    (DESIGN.md section 2); what output_fcn reads after EVERY iteration is what an every-point store leaves"""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from pymgrit_amd import Heat1D, Mgrit

    def run(store_all):
        if store_all:
            monkeypatch.setenv("PYMGRIT_AMD_STORE_ALL_F", "1")
        else:
            monkeypatch.delenv("PYMGRIT_AMD_STORE_ALL_F", raising=False)
        seen = []

        def output_fcn(self):
            seen.append(np.array([self.u[0][i].get_values() for i in (1, 2, 3, 4, 6, len(self.t[0]) - 2)]))
        prob = [Heat1D(x_start=0, x_end=1, nx=1200, a=1, init_cond=cases.init_cond, rhs_separable=[(cases.rhs_space, cases.rhs_time)],
                       t_interval=t) for t in (cases.lin(2, 65), cases.lin(2, 17), cases.lin(2, 5))]
        mg = Mgrit(prob, logging_lvl=30, max_iter=4, tol=0.0, output_fcn=output_fcn, output_lvl=2)
        assert mg._level_intervals(0) is not None
        conv = mg.solve()["conv"]
        return conv, seen
    conv0, seen0 = run(True)
    conv1, seen1 = run(False)
    assert np.array_equal(conv0, conv1) and len(seen0) == len(seen1) >= 4
    for a, b in zip(seen0, seen1):
        assert np.array_equal(a, b)
