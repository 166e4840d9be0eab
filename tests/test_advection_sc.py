"""Advection1D with periodic spatial coarsening (BASELINE config 5 family, shrunk): oracle vs the reference run with a
user periodic transfer (CPU), product vs oracle (GPU)."""
import numpy as np
import pytest

import cases


@pytest.mark.parametrize("name", sorted(cases.ADV_SC))
@pytest.mark.parametrize("variant", [0, 1])
def test_oracle_matches_reference(oracle, name, variant):
    rec, nxs, ts, transfer, opts = cases.adv_sc_case(name)
    p = oracle.OracleProblem([cases.advection_level_spec(nx, t) for nx, t in zip(nxs, ts)], transfer=transfer,
                             variant=variant, **opts)
    conv = p.solve()
    ref = np.array(rec["conv"])
    assert len(conv) == len(ref) and np.all(np.abs(conv - ref) <= 1e-9 * ref + 2e-11), (conv, ref)
    assert np.abs(p.state("u", 0)[128] - np.array(rec["u_last"])).max() <= 1e-11


def test_periodic_transfer_class_matches_oracle(oracle):
    from pymgrit_amd import GridTransferAdvection
    from pymgrit_amd.advection.advection_1d import VectorAdvection1D
    g = GridTransferAdvection()
    f = np.cos(np.arange(64.0)) * 3 + np.arange(64.0) ** 0.5
    v = VectorAdvection1D(64)
    v.set_values(f)
    c = g.restriction(v).get_values()
    assert np.array_equal(c, oracle.restrict(2, f, 32))
    w = VectorAdvection1D(32)
    w.set_values(c)
    assert np.array_equal(g.interpolation(w).get_values(), oracle.interp(2, c, 64))


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(cases.ADV_SC))
def test_gpu_matches_oracle_and_reference(oracle, name):
    import torch
    assert torch.cuda.is_available()
    from pymgrit_amd import Advection1D, GridTransferAdvection, GridTransferCopy, Mgrit
    rec, nxs, ts, transfer, opts = cases.adv_sc_case(name)
    prob = [Advection1D(c=1, x_start=-1, x_end=1, nx=nx, t_interval=t) for nx, t in zip(nxs, ts)]
    tr = [GridTransferAdvection() if k == 2 else GridTransferCopy() for k in transfer]
    mg = Mgrit(prob, transfer=tr, logging_lvl=30, **opts)
    conv = mg.solve()["conv"]
    op = oracle.OracleProblem([cases.advection_level_spec(nx, t) for nx, t in zip(nxs, ts)], transfer=transfer, variant=1,
                              **opts)
    oconv = op.solve()
    assert len(conv) == len(oconv) and np.max(np.abs(conv - oconv) / oconv) <= 1e-10
    assert np.array_equal(mg.backend.natural("u", 0), op.state("u", 0))
    ref = np.array(rec["conv"])
    assert np.all(np.abs(conv - ref) <= 1e-9 * ref + 2e-11)


@pytest.mark.gpu
def test_config5_scale_sweeps_bit_exact(oracle):
    """config 5 spatial sizes (8192 -> 4096 -> 2048 periodic points), short time grid, every sweep vs the oracle"""
    import torch
    assert torch.cuda.is_available()
    from test_hip_parity import assert_state_equal, randomize
    from pymgrit_amd import Advection1D, GridTransferAdvection, Mgrit
    t0 = np.linspace(0, 2.0 * 32 / 32768, 33)
    ts, nxs = [t0, t0[::2], t0[::4]], [8193, 4097, 2049]
    prob = [Advection1D(c=1, x_start=-1, x_end=1, nx=nx, t_interval=t) for nx, t in zip(nxs, ts)]
    mg = Mgrit(prob, transfer=[GridTransferAdvection(), GridTransferAdvection()], nested_iteration=False, logging_lvl=30)
    op = oracle.OracleProblem([cases.advection_level_spec(nx, t) for nx, t in zip(nxs, ts)], transfer=[2, 2], variant=1,
                              nested_iteration=False)
    randomize(mg, op, seed=5)
    for lvl in (0, 1):
        mg.f_relax(lvl); op.f_relax(lvl)
        mg.c_relax(lvl); op.c_relax(lvl)
        mg.fas_residual(lvl); op.fas_residual(lvl)
        assert_state_equal(mg, op)
    mg.forward_solve(2); op.forward_solve(2)
    for lvl in (1, 0):
        mg.error_correction(lvl); op.error_correction(lvl)
        assert_state_equal(mg, op)


@pytest.mark.gpu
def test_config5_full_size_properties(oracle):
    """BASELINE config 5 at FULL size: 8192 periodic points, nt=32769, 4 levels m=2, periodic coarsening on the first two
    level pairs. F-cycle runs and contracts; F-relax idempotent; restriction of a constant is that constant and
    interpolation reproduces constants (partition of unity) on the full slabs; a sampled step equals the oracle."""
    import torch
    from pymgrit_amd import Advection1D, GridTransferAdvection, GridTransferCopy, Mgrit
    t0 = np.linspace(0, 2, 32769)
    ts, nxs = [t0, t0[::2], t0[::4], t0[::8]], [8193, 4097, 2049, 2049]
    prob = [Advection1D(c=1, x_start=-1, x_end=1, nx=nx, t_interval=t) for nx, t in zip(nxs, ts)]
    tr = [GridTransferAdvection(), GridTransferAdvection(), GridTransferCopy()]
    mg = Mgrit(prob, transfer=tr, cycle_type='F', nested_iteration=True, max_iter=2, tol=0.0, logging_lvl=30)
    conv = mg.solve()["conv"]
    assert len(conv) == 2 and conv[1] < conv[0]
    before = mg.backend.U[0].clone()
    mg.f_relax(0)
    once = mg.backend.U[0].clone()
    mg.f_relax(0)
    assert torch.equal(once, mg.backend.U[0]) and before.shape == once.shape
    # sampled step vs the oracle
    op = oracle.OracleProblem([cases.advection_level_spec(8193, t0[:3])], variant=1)
    x = mg.backend.natural("u", 0)[16384]
    assert np.array_equal(mg.backend.natural("u", 0)[16385], op.phi(0, 1, x))
    # transfers on constants
    mg.backend.U[0].zero_()
    mg.backend.U[0][:, mg.backend.perm[0]] = 3.0
    mg.backend.restrict_u(0, mg._pairs(0, skip_first=False))
    got = mg.backend.natural("u", 1)
    assert np.all(got == 3.0)
    mg.backend.interpolate(0, mg._pairs(0, skip_first=True))
    assert np.all(mg.backend.natural("u", 0) == 3.0)
