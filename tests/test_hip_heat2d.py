"""GPU parity for Heat2D (BASELINE config 4 family): the MFMA fast-diagonalisation stepper and every sweep on 2-D states
against the oracle (same arithmetic: fma dot products in ascending k = v_mfma_f64_16x16x4 chains), bit for bit; solves
against the oracle (1e-10 rel) and the reference fixtures (tests/golden/heat2d.*)."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

H2D = cases.load_json("heat2d.json")
ARR = np.load(cases.GOLDEN + "/heat2d.npz")


def _pair(oracle, prob, **opts):
    from pymgrit_amd import Mgrit
    opts.setdefault("nested_iteration", False)
    mg = Mgrit(prob, logging_lvl=30, **opts)
    op = oracle.OracleProblem([cases.h2d_level_spec(a) for a in prob], **opts)
    return mg, op


def _randomize(mg, op, seed):
    rng = np.random.default_rng(seed)
    for lvl in range(mg.lvl_max):
        for name, slabs in (("u", mg.backend.U), ("v", mg.backend.V), ("g", mg.backend.G)):
            if slabs[lvl] is None:
                continue
            ref = op.state(name, lvl)
            ref[:] = rng.standard_normal(ref.shape)
            mg.backend.set_natural(name, lvl, ref)


def _equal(mg, op):
    for lvl in range(mg.lvl_max):
        for name, slabs in (("u", mg.backend.U), ("v", mg.backend.V), ("g", mg.backend.G)):
            if slabs[lvl] is None:
                continue
            ref, got = op.state(name, lvl), mg.backend.natural(name, lvl)
            assert np.array_equal(got, ref), (name, lvl, np.abs(got - ref).max())


SHAPES = [("BE", False, 9, 12), ("BE", True, 20, 17), ("CN", True, 12, 10), ("FE", False, 8, 8), ("BE", True, 66, 67),
          ("CN", False, 70, 40), ("BE", False, 130, 131)]


@pytest.mark.parametrize("method,with_bc,nx,ny", SHAPES, ids=[f"{m}-bc{int(b)}-{x}x{y}" for m, b, x, y in SHAPES])
def test_sweeps_bit_exact(oracle, method, with_bc, nx, ny):
    assert torch.cuda.is_available()
    a = 0.05 if method == "FE" else cases.H2D_A
    prob = [cases.h2d_app(nx, ny, t, method, with_bc, a) for t in cases.h2d_grids([33, 9, 3])]
    for w in (1.0, 1.3):
        mg, op = _pair(oracle, prob, weight_c=w)
        _randomize(mg, op, nx + ny)
        _every_sweep_once(mg, op)


def _every_sweep_once(mg, op):
    for lvl in range(mg.lvl_max - 1):
        mg.f_relax(lvl); op.f_relax(lvl)
        _equal(mg, op)
        mg.c_relax(lvl); op.c_relax(lvl)
        _equal(mg, op)
        mg.fas_residual(lvl); op.fas_residual(lvl)
        _equal(mg, op)
    mg.forward_solve(mg.lvl_max - 1); op.forward_solve(mg.lvl_max - 1)
    _equal(mg, op)
    for lvl in range(mg.lvl_max - 2, -1, -1):
        mg.error_correction(lvl); op.error_correction(lvl)
        _equal(mg, op)
    got, ref = np.array(mg.compute_residual()), op.residual_norms()
    assert np.array_equal(got, ref), np.abs(got - ref).max()


@pytest.mark.parametrize("nx,ny", [(9, 12), (20, 17), (4, 3), (66, 67), (67, 130), (131, 129), (200, 66)])
def test_homogeneous_backward_euler_sweeps_bit_exact(oracle, nx, ny):
    """the reference's default right-hand side (zero, heat_2d.py:148) with zero boundary values: the step's right-hand side is the
    interior of u itself, and the first transform reads it from the state rows (no rhs launch; h2d_kloop GRID) -- sizes with odd
    and even interiors, below / at / above the 64-wide tiles, random states with random rims (which must not leak in)"""
    assert torch.cuda.is_available()
    prob = [cases.h2d_app(nx, ny, t, "BE", False, forcing=False) for t in cases.h2d_grids([33, 9, 3])]
    d = prob[0].device_stepper()
    assert len(d["forcing_time"]) == 0 and not np.any(d["bc"])
    mg, op = _pair(oracle, prob)
    _randomize(mg, op, nx + ny)
    _every_sweep_once(mg, op)
    for it in range(2):
        mg.iteration(lvl=0, cycle_type='V', iteration=it, first_f=True); op.iteration(0, 'V', it, True)
        _equal(mg, op)


BLK_SHAPES = [("be_bc_66x67_uniform", True, 66, 67, None), ("be_130x131_nonuniform", False, 130, 131, 1.3), ("be_bc_20x17_rem", True, 20, 17, None),
              # Crank-Nicolson (round 5): the random states have random rims, so the first solves are stepped (both sides decide the
              # same from the same bits); the cycle after them runs on states whose rims hold the boundary values: the modal form
              ("cn_bc_66x67_uniform", True, 66, 67, None), ("cn_40x35_nonuniform", False, 40, 35, 1.3)]


@pytest.mark.parametrize("name,with_bc,nx,ny,power", BLK_SHAPES, ids=[s[0] for s in BLK_SHAPES])
def test_time_parallel_forward_solve_bit_exact(oracle, name, with_bc, nx, ny, power):
    """coarsest levels of >= 64 steps, backward Euler: the time-parallel forward solve (blocks of 16 steps from
    zero states, the block ends through the full sine spectrum) against the oracle's statement of it, random states; uniform
    steps, every step its own size, a last block that takes the remainder (77 steps = 3 x 16 + 29); and the step-by-step form
    on request"""
    from pymgrit_amd.core.options import options
    assert torch.cuda.is_available()
    t0 = np.linspace(0, 1, 309) if power is None else np.linspace(0, 1, 321) ** power
    ts = [t0, t0[::4]]
    method = "CN" if name.startswith("cn_") else "BE"
    kw = dict(a=0.1) if method == "CN" else {}
    prob = [cases.h2d_app(nx, ny, t, method, with_bc, **kw) for t in ts]
    mg, op = _pair(oracle, prob)
    assert mg.backend.block_r[1] == (nx - 2) * (ny - 2)
    _randomize(mg, op, nx)
    for rep in range(2):
        mg.forward_solve(1); op.forward_solve(1)
        _equal(mg, op)
    for it in range(2):
        mg.iteration(lvl=0, cycle_type='V', iteration=it, first_f=True); op.iteration(0, 'V', it, True)
        _equal(mg, op)
    if method == "CN":     # the cycles above did take the modal form: the oracle that steps through the level differs in the last bits
        op_seq = oracle.OracleProblem([cases.h2d_level_spec(a) for a in prob], nested_iteration=False, block_solve=False)
        _randomize(mg, op_seq, nx)
        for rep in range(2):
            op_seq.forward_solve(1)
        for it in range(2):
            op_seq.iteration(0, 'V', it, True)
        assert not np.array_equal(op_seq.state("u", 1), op.state("u", 1))
        assert np.abs(op_seq.state("u", 1) - op.state("u", 1)).max() <= 1e-9 * np.abs(op.state("u", 1)).max()
        return
    try:
        options.coarse_solve = "sequential"
        mg2, _ = _pair(oracle, [cases.h2d_app(nx, ny, t, "BE", with_bc) for t in ts])
    finally:
        options.reset("coarse_solve")
    op2 = oracle.OracleProblem([cases.h2d_level_spec(a) for a in prob], nested_iteration=False, block_solve=False)
    assert mg2.backend.block_r[1] == 0
    _randomize(mg2, op2, nx)
    mg2.forward_solve(1); op2.forward_solve(1)
    _equal(mg2, op2)


@pytest.mark.parametrize("name", sorted(cases.H2D_SOLVE))
def test_solve_matches_oracle_and_reference(oracle, name):
    from pymgrit_amd import Mgrit
    prob, opts = cases.h2d_solve_problem(name)
    mg = Mgrit(prob, logging_lvl=30, **opts)
    conv = mg.solve()["conv"]
    op = oracle.OracleProblem([cases.h2d_level_spec(a) for a in prob], **opts)
    oconv = op.solve()
    assert len(conv) == len(oconv) and np.max(np.abs(conv - oconv) / oconv) <= 1e-10, (conv, oconv)
    ref = np.array(H2D["solve"][name]["conv"])
    assert np.all(np.abs(conv - ref) <= 1e-9 * ref + 2e-11), (conv, ref)
    assert np.array_equal(mg.backend.natural("u", 0), op.state("u", 0))
    last = mg.u[0][len(prob[0].t) - 1].get_values()
    refu = ARR["h2d_solve_" + name]
    assert last.shape == refu.shape and np.abs(last - refu).max() <= 1e-11 * max(1.0, np.abs(refu).max())


def test_reference_unit_test_kats_on_gpu():
    """reference tests/heat/test_heat_2d.py:230-293 through the GPU engine: one step of a 1-level hierarchy"""
    from pymgrit_amd import Mgrit
    from pymgrit_amd.heat.heat_2d import Heat2D
    from test_heat2d_cpu import REF_KAT
    for method in ("BE", "CN", "FE"):
        app = Heat2D(a=1, x_start=0, x_end=1, y_start=3, y_end=4, nx=5, ny=5, method=method, rhs=lambda x, y, t: 2 * x * y,
                     t_start=0, t_stop=1, nt=11)
        mg = Mgrit([app], logging_lvl=30)   # nested iteration on one level = time stepping
        np.testing.assert_almost_equal(mg.u[0][1].get_values(), np.array(REF_KAT[method]))


def test_config4_size_properties():
    """512x512 states (config 4 spatial size), short time grid: F-relax is idempotent, leaves zero residual at F-points,
    and with zero forcing / zero BC Phi is linear under exact scaling by a power of two."""
    from pymgrit_amd import Mgrit
    from pymgrit_amd.heat.heat_2d import Heat2D
    ts = cases.h2d_grids([33, 5])
    prob = [Heat2D(x_start=0, x_end=1, y_start=0, y_end=1, nx=512, ny=512, a=1.0, method="BE", t_interval=t) for t in ts]
    mg = Mgrit(prob, nested_iteration=False, logging_lvl=30)
    U = mg.backend.U[0]
    gen = torch.Generator(device="cpu").manual_seed(1)
    c = torch.randn((len(ts[1]), 512, 512), generator=gen, dtype=torch.float64)
    c[:, 0, :] = 0; c[:, -1, :] = 0; c[:, :, 0] = 0; c[:, :, -1] = 0
    U[::8, :512 * 512] = c.reshape(len(ts[1]), -1).to(U.device)
    mg.f_relax(0)
    first = U.clone()
    mg.f_relax(0)
    assert torch.equal(first, U)
    fpts = [int(i) for i in np.sort(mg.index_local_f[0])]
    assert max(mg.backend.residual_norms(fpts)) == 0.0
    U[::8, :512 * 512] = (4.0 * c).reshape(len(ts[1]), -1).to(U.device)
    mg.f_relax(0)
    assert torch.equal(first * 4.0, U)


def _oracle_threads(oracle):
    return oracle.set_h2d_threads(cases.usable_cpus(16))


def test_config4_full_size_properties(oracle):
    """BASELINE config 4 at FULL size: 512x512 states, nt=16385, m=8 (34 GB level-0 slab). F-relax idempotence and zero
    F-point residual over the whole grid, exact scaling linearity, and one sampled level-0 interval (its 7 F-points) bit for bit
    the ORACLE's steps at 512 x 512 (reference heat_2d.py:322-366)."""
    from pymgrit_amd import Mgrit
    from pymgrit_amd.heat.heat_2d import Heat2D
    free, _ = torch.cuda.mem_get_info()
    if free < 70 * 2 ** 30:
        pytest.skip("needs ~60 GB of free HBM")
    ts = cases.h2d_grids([16385, 2049])
    prob = [Heat2D(x_start=0, x_end=1, y_start=0, y_end=1, nx=512, ny=512, a=1.0, method="BE", t_interval=t) for t in ts]
    mg = Mgrit(prob, nested_iteration=False, logging_lvl=30)
    U = mg.backend.U[0]
    gen = torch.Generator(device="cpu").manual_seed(2)
    base = torch.randn((512, 512), generator=gen, dtype=torch.float64)
    base[0, :] = 0; base[-1, :] = 0; base[:, 0] = 0; base[:, -1] = 0
    scale = torch.linspace(0.5, 1.5, len(ts[1]), dtype=torch.float64)
    U[::8, :512 * 512] = (scale[:, None] * base.reshape(1, -1)).to(U.device)
    mg.f_relax(0)
    ref_rows = U[8 * 1000 + 1:8 * 1000 + 8].clone()
    first_sum = U.sum().item()
    mg.f_relax(0)
    assert torch.equal(ref_rows, U[8 * 1000 + 1:8 * 1000 + 8]) and U.sum().item() == first_sum
    fpts = [int(i) for i in np.sort(mg.index_local_f[0])]
    assert max(mg.backend.residual_norms(fpts[:3000] + fpts[-3000:])) == 0.0
    # the oracle on one interval, AT this size: a two-level oracle problem over the 9 time points of interval 1000 (Phi depends on
    # the step's end points only: the same doubles), its 7 F-steps from the same C-point -- every value bit for bit
    _oracle_threads(oracle)
    t_int = np.ascontiguousarray(ts[0][8000:8009])
    op = oracle.OracleProblem([cases.h2d_level_spec(Heat2D(x_start=0, x_end=1, y_start=0, y_end=1, nx=512, ny=512, a=1.0, method="BE",
                                                             t_interval=t)) for t in (t_int, t_int[::8])], nested_iteration=False)
    v = (scale[1000] * base).numpy().ravel()
    for k in range(1, 8):
        v = op.phi(0, k, v)
        got = U[8000 + k, :512 * 512].cpu().numpy()
        assert np.array_equal(got, v), (k, np.abs(got - v).max())
    U[::8, :512 * 512] = (2.0 * scale[:, None] * base.reshape(1, -1)).to(U.device)
    mg.f_relax(0)
    assert torch.equal(ref_rows * 2.0, U[8 * 1000 + 1:8 * 1000 + 8])


def test_config4_size_block_solve_matches_the_oracle(oracle):
    """512 x 512 states (BASELINE config 4's size, forcing and boundary values of the fixtures), 2 levels m = 8, nt = 513: the coarsest
    level's 64 steps = 4 blocks of the time-parallel forward solve on the full 510 x 510 sine spectrum, from random
    u / g rows, against the oracle's block solve at the same size -- every row of the level bit for bit; then a whole V-cycle with
    its per-point residual norms. (The oracle's transforms run on up to 16 host threads: same bits, 30 ms per step.)"""
    assert torch.cuda.is_available()
    _oracle_threads(oracle)
    prob = [cases.h2d_app(512, 512, t, "BE", True) for t in cases.h2d_grids([513, 65])]
    mg, op = _pair(oracle, prob)
    assert mg.backend.block_r[1] == 510 * 510
    _randomize(mg, op, 512)
    mg.forward_solve(1); op.forward_solve(1)
    got, ref = mg.backend.natural("u", 1), op.state("u", 1)
    assert np.array_equal(got, ref), np.abs(got - ref).max()
    mg.iteration(lvl=0, cycle_type='V', iteration=0, first_f=True); op.iteration(0, 'V', 0, True)
    got, ref = np.array(mg.compute_residual()), op.residual_norms()
    assert np.array_equal(got, ref), np.abs(got - ref).max()


@pytest.mark.parametrize("method", ["BE", "CN", "FE"])
def test_general_forcing_sweeps_bit_exact(oracle, method):
    """rhs(x, y, t) that is not S0 + S1*t (the reference takes any callable, heat_2d.py:148,289-320): the engine streams
    precomputed rows (mgrit_hip_level_heat2d_forcing_rows); every sweep against the oracle's rows form, bit for bit"""
    assert torch.cuda.is_available()
    a = 0.05 if method == "FE" else cases.H2D_A
    prob = [cases.h2d_general_app(20, 17, t, method, a) for t in cases.h2d_grids([33, 9, 3])]
    mg, op = _pair(oracle, prob)
    assert mg.backend.desc[0]["forcing_rows"] is not None and type(mg.backend).__name__ == "HipBackend"
    _randomize(mg, op, 5)
    for lvl in range(mg.lvl_max - 1):
        mg.f_relax(lvl); op.f_relax(lvl)
        _equal(mg, op)
        mg.c_relax(lvl); op.c_relax(lvl)
        _equal(mg, op)
        mg.fas_residual(lvl); op.fas_residual(lvl)
        _equal(mg, op)
    mg.forward_solve(mg.lvl_max - 1); op.forward_solve(mg.lvl_max - 1)
    _equal(mg, op)
    got, ref = np.array(mg.compute_residual()), op.residual_norms()
    assert np.array_equal(got, ref)


def test_general_forcing_solves_match_oracle_and_reference(oracle):
    from pymgrit_amd import Mgrit
    assert torch.cuda.is_available()
    meta, arr = cases.load_json("heat2d_general.json"), np.load(cases.GOLDEN + "/heat2d_general.npz")
    for name, rec in meta["solve"].items():
        prob = [cases.h2d_general_app(rec["nx"], rec["ny"], t, rec["method"], rec["a"]) for t in cases.h2d_grids(rec["nts"])]
        opts = dict(tol=1e-9, max_iter=8 if "fe" not in name else 6, nested_iteration=name == "general_be_3lvl")
        mg = Mgrit(prob, logging_lvl=30, **opts)
        conv = mg.solve()["conv"]
        oconv = oracle.OracleProblem([cases.h2d_level_spec(a) for a in prob], **opts).solve()
        assert len(conv) == len(oconv) and np.max(np.abs(conv - oconv) / oconv) <= 1e-10, (name, conv, oconv)
        ref = np.asarray(rec["conv"])
        assert len(conv) == len(ref) and np.all(np.abs(conv - ref) <= 1e-9 * ref + 2e-11), (name, conv, ref)
        last = np.asarray(mg.u[0][len(prob[0].t) - 1].get_values())
        assert np.max(np.abs(last - arr["solve_" + name])) <= 1e-10 * max(1.0, np.max(np.abs(last))), name


@pytest.mark.parametrize("name", ["be_3lvl_F_bc", "cn_2lvl", "example_small"])
def test_planned_cycle_on_cu_partitions_is_bit_identical(name, monkeypatch):
    """Heat2D planned cycle: blocks of time points, the coarsest-level solve of a block beside the sweeps of the others, each side
    on CUs of its own (mgrit_hip_stream_create_masked; the solve has work buffers of its own) -- same bits as the program order"""
    from pymgrit_amd import Mgrit
    assert torch.cuda.is_available()
    out = []
    for blocks in (1, 3, 5):
        monkeypatch.setenv("PYMGRIT_AMD_PLAN_BLOCKS_HEAT2D", str(blocks))
        prob, opts = cases.h2d_solve_problem(name)
        mg = Mgrit(prob, logging_lvl=30, **opts)
        conv = mg.solve()["conv"]
        if blocks > 1 and mg.plan_blocks() > 1:
            assert any(p is not None for p in mg._plans.values()) and mg.backend._masked_streams() is not None
        out.append((conv, mg.backend.natural("u", 0)))
    for conv, u in out[1:]:
        assert np.array_equal(conv, out[0][0]) and np.array_equal(u, out[0][1])
