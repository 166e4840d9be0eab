"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI via pymgrit_amd, against the
parity oracle (variant 1 = the arithmetic spec) on the same seeded inputs.

Bars: state vectors and per-point residual norms BIT-EXACT vs the oracle (same arithmetic spec);
residual-norm history within 1e-10 relative per iteration (north_star), in practice identical; and within the
reference-fixture tolerance (1e-9 rel + 2e-11 abs, see test_oracle_golden.py) of the reference's own numbers.
"""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible (HIP path has no CPU fallback)")


def heat_problem(nx, grids, x_end=1.0, a=1.0, forcing=True):
    from pymgrit_amd import Heat1D
    kw = dict(rhs_separable=[(cases.rhs_space, cases.rhs_time)]) if forcing else {}
    return [Heat1D(x_start=0, x_end=x_end, nx=nx, a=a, init_cond=cases.init_cond, t_interval=np.asarray(t), **kw)
            for t in grids]


def advection_problem(nx, grids, c=1.0):
    from pymgrit_amd import Advection1D
    return [Advection1D(c=c, x_start=-1, x_end=1, nx=nx, t_interval=np.asarray(t)) for t in grids]


def make_pair(oracle, kind, nx, grids, transfer=None, x_end=1.0, forcing=True, **opts):
    """(product Mgrit on the GPU, oracle problem) for the same hierarchy; no nested iteration unless asked."""
    from pymgrit_amd import GridTransferAdvection, GridTransferCopy, GridTransferHeat, Mgrit
    opts.setdefault("nested_iteration", False)
    nxs = nx if isinstance(nx, (list, tuple)) else [nx] * len(grids)
    if kind == "heat":
        prob = [heat_problem(n, [t], x_end=x_end, forcing=forcing)[0] for n, t in zip(nxs, grids)]
        specs = [cases.heat_level_spec(n, t, x_end=x_end, forcing=forcing) for n, t in zip(nxs, grids)]
    else:
        prob = [advection_problem(n, [t])[0] for n, t in zip(nxs, grids)]
        specs = [cases.advection_level_spec(n, t) for n, t in zip(nxs, grids)]
    tr = None
    if transfer is not None:
        tr = [GridTransferHeat() if k == 1 else GridTransferAdvection() if k == 2 else GridTransferCopy() for k in transfer]
    mg = Mgrit(prob, transfer=tr, logging_lvl=30, **opts)
    oopts = {k: v for k, v in opts.items() if k != "random_init_guess"}
    op = oracle.OracleProblem(specs, transfer=transfer, variant=1, **oopts)
    return mg, op


def randomize(mg, op, seed=0):
    """same random u, v, g on both sides"""
    rng = np.random.default_rng(seed)
    for lvl in range(mg.lvl_max):
        for name, slabs in (("u", mg.backend.U), ("v", mg.backend.V), ("g", mg.backend.G)):
            if slabs[lvl] is None:
                continue
            ref = op.state(name, lvl)
            ref[:] = rng.standard_normal(ref.shape)
            mg.backend.set_natural(name, lvl, ref)


def assert_state_equal(mg, op, what=("u", "v", "g")):
    for lvl in range(mg.lvl_max):
        for name, slabs in (("u", mg.backend.U), ("v", mg.backend.V), ("g", mg.backend.G)):
            if name not in what or slabs[lvl] is None:
                continue
            ref = op.state(name, lvl)
            got = mg.backend.natural(name, lvl)
            assert np.array_equal(got, ref), (name, lvl, np.abs(got - ref).max())
            pad = torch.ones(slabs[lvl].shape[1], dtype=torch.bool, device=slabs[lvl].device)
            pad[mg.backend.perm[lvl]] = False
            # padding positions of a row: unspecified FINITE values (a Phi whose rank-one correction is evaluated in closed form
            # leaves -z0 * w there instead of 0; the kernels zero them between their scans and mask them in the norms)
            assert torch.isfinite(slabs[lvl][:, pad]).all().item(), ("padding positions must stay finite", name, lvl)


GRIDS3 = [cases.lin(2, 65), cases.lin(2, 17), cases.lin(2, 5)]

SWEEP_SHAPES = [
    ("heat", 5, GRIDS3), ("heat", 33, GRIDS3), ("heat", 1024, GRIDS3), ("heat", 1027, GRIDS3), ("heat", 2050, GRIDS3),
    ("heat", 4099, GRIDS3), ("heat", 16384, [cases.lin(2, 17), cases.lin(2, 5), cases.lin(2, 3)]),
    ("heat", 16386, [cases.lin(2, 9), cases.lin(2, 5), cases.lin(2, 3)]),        # n = 16384: the largest supported state
    ("heat", 300, [cases.lin(5, 101), cases.lin(5, 51), cases.lin(5, 26)]),      # several distinct dt per level
    ("advection", 6, GRIDS3), ("advection", 129, GRIDS3), ("advection", 1026, GRIDS3), ("advection", 8193, GRIDS3),
]


@pytest.mark.parametrize("kind,nx,grids", SWEEP_SHAPES, ids=[f"{k}-nx{n}-nt{len(g[0])}" for k, n, g in SWEEP_SHAPES])
def test_sweeps_bit_exact(oracle, kind, nx, grids):
    """every sweep of the hot path, on every level, against the oracle: bit-exact states"""
    _need_gpu()
    mg, op = make_pair(oracle, kind, nx, grids, weight_c=1.0)
    randomize(mg, op, seed=nx)
    for lvl in range(mg.lvl_max - 1):
        mg.f_relax(lvl); op.f_relax(lvl)
        assert_state_equal(mg, op)
        mg.c_relax(lvl); op.c_relax(lvl)
        assert_state_equal(mg, op)
        mg.fas_residual(lvl); op.fas_residual(lvl)
        assert_state_equal(mg, op)
    mg.forward_solve(mg.lvl_max - 1); op.forward_solve(mg.lvl_max - 1)
    assert_state_equal(mg, op)
    for lvl in range(mg.lvl_max - 2, -1, -1):
        mg.error_correction(lvl); op.error_correction(lvl)
        assert_state_equal(mg, op)
    got = np.array(mg.compute_residual())
    ref = op.residual_norms()
    assert np.array_equal(got, ref), np.abs(got - ref).max()


@pytest.mark.parametrize("w", [1.3, 0.7])
def test_weighted_c_relax_bit_exact(oracle, w):
    _need_gpu()
    mg, op = make_pair(oracle, "heat", 257, GRIDS3, weight_c=w)
    randomize(mg, op, seed=7)
    for lvl in (0, 1):
        mg.c_relax(lvl); op.c_relax(lvl)
        assert_state_equal(mg, op)


def test_spatial_coarsening_sweeps_bit_exact(oracle):
    """full-weighting restriction / linear interpolation kernels (example_spatial_coarsening.py:33-82)"""
    _need_gpu()
    t0 = cases.lin(2, 129)
    for nxs in ([17, 9, 5, 5], [2049, 1025, 513, 513]):
        mg, op = make_pair(oracle, "heat", nxs, [t0, t0[::2], t0[::4], t0[::8]], transfer=[1, 1, 0], x_end=2.0)
        randomize(mg, op, seed=3)
        for lvl in range(3):
            mg.fas_residual(lvl); op.fas_residual(lvl)
            assert_state_equal(mg, op)
        for lvl in (2, 1, 0):
            mg.error_correction(lvl); op.error_correction(lvl)
            assert_state_equal(mg, op)
        mg.nested_iteration(); op.nested_iteration()
        assert_state_equal(mg, op)


def test_non_uniform_coarsening_adjacent_c_points(oracle):
    """C-points 33,34 adjacent (reference tests/mpi/varying_coarsening.py:14): the second sees the first's NEW value"""
    _need_gpu()
    t0 = cases.lin(5, 65)
    v1 = t0[[0, 3, 10, 12, 14, 17, 23, 27, 33, 34, 55, 57, 59, 61, 63, 64]]
    mg, op = make_pair(oracle, "heat", 65, [t0, v1, v1[::2], v1[::4]])
    randomize(mg, op, seed=11)
    for lvl in range(3):
        mg.f_relax(lvl); op.f_relax(lvl)
        mg.c_relax(lvl); op.c_relax(lvl)
        assert_state_equal(mg, op)
        mg.fas_residual(lvl); op.fas_residual(lvl)
        assert_state_equal(mg, op)


SOLVE = cases.load_json("solve.json")
DEVICE_CASES = [n for n in cases.solve_cases() if n.startswith(("heat_", "advection_"))]


def build_product(name):
    """product hierarchy for a solve case, rebuilt from the same parameters make_golden.py used"""
    from pymgrit_amd import Advection1D, GridTransferCopy, GridTransferHeat, Heat1D
    P = {  # name -> (kind, nx list, x_end, forcing)
        "heat_nx5_test_mgrit": ("heat", 5, 2.0, True), "heat_nx5_to_tol": ("heat", 5, 2.0, True),
        "heat_nx33_2lvl_m8": ("heat", 33, 1.0, True), "heat_nx33_1lvl": ("heat", 33, 1.0, True),
        "heat_nx33_noforcing": ("heat", 33, 1.0, False), "heat_nx257_nt257": ("heat", 257, 1.0, True),
        "heat_example_F5": ("heat", 1001, 1.0, True), "heat_example_F5_w13": ("heat", 1001, 1.0, True),
        "heat_spatial_coarsening": ("heat", [17, 9, 5, 5], 2.0, True),
        "heat_spatial_coarsening_F": ("heat", [129, 65, 33], 1.0, True),
        "advection_example": ("advection", 129, None, None), "advection_3lvl_F": ("advection", 257, None, None),
        "heat_config2": ("heat", 1024, 1.0, True),
        "heat_nx2050_wide": ("heat", 2050, 1.0, True), "heat_nx1500_wide_F": ("heat", 1500, 1.0, False),
        "heat_nx3100_wide_2lvl": ("heat", 3100, 1.0, True),
        # coarsest levels whose forward solve takes the time-parallel form
        "heat_blk_nx257_3lvl": ("heat", 257, 1.0, True), "heat_blk_nx2050_2lvl": ("heat", 2050, 1.0, True),
        "heat_blk_nonuniform_F": ("heat", 129, 1.0, True), "heat_blk_noforcing_cf0": ("heat", 65, 1.0, False),
        "heat_blk_r127_2lvl": ("heat", 1025, 1.0, True),
        "advection_blk_nx257_3lvl_F": ("advection", 257, None, None), "advection_blk_nx1025_2lvl": ("advection", 1025, None, None),
        "advection_blk_nonuniform": ("advection", 129, None, None), "advection_blk_nx201_3lvl_F": ("advection", 201, None, None),
    }
    c = cases.solve_cases()[name]
    if name not in P and not name.startswith("heat_nx33_"):
        raise KeyError(f"solve case {name!r} has no entry in test_hip_parity.build_product")
    kind, nx, x_end, forcing = P.get(name, ("heat", 33, 1.0, True))
    grids = [spec["t"] for spec in c["levels"]]
    nxs = nx if isinstance(nx, list) else [nx] * len(grids)
    if kind == "heat":
        prob = [heat_problem(n, [t], x_end=x_end, forcing=forcing)[0] for n, t in zip(nxs, grids)]
    else:
        prob = [advection_problem(n, [t])[0] for n, t in zip(nxs, grids)]
    tr = None
    if c.get("transfer") is not None:
        tr = [GridTransferHeat() if k == 1 else GridTransferCopy() for k in c["transfer"]]
    return prob, tr, c


@pytest.mark.parametrize("name", DEVICE_CASES)
def test_solve_matches_oracle_and_reference(oracle, name):
    """Mgrit.solve() on the GPU: residual history within 1e-10 rel of the oracle per iteration (north_star), within
    the fixture tolerance of the reference, sampled solution vectors bit-exact vs the oracle."""
    _need_gpu()
    from pymgrit_amd import Mgrit
    prob, tr, c = build_product(name)
    opts = dict(c["opts"])
    if c.get("seed") is not None:
        np.random.seed(c["seed"])
    mg = Mgrit(prob, transfer=tr, logging_lvl=30, **opts)
    conv = mg.solve()["conv"]
    oopts = {k: v for k, v in opts.items() if k != "random_init_guess"}
    op = oracle.OracleProblem(c["levels"], transfer=c.get("transfer"), variant=1, **oopts)
    if opts.get("random_init_guess"):
        np.random.seed(c["seed"])
        u = op.state("u", 0)
        for i in range(u.shape[0]):
            u[i] = np.random.rand(u.shape[1])
        u[0] = c["levels"][0]["u0"]
    oconv = op.solve()
    assert len(conv) == len(oconv), (conv, oconv)
    if len(conv):
        assert np.max(np.abs(conv - oconv) / oconv) <= 1e-10, (conv, oconv)
    ref = np.array(SOLVE[name]["conv"])
    n = min(len(ref), len(conv))
    assert np.all(np.abs(conv[:n] - ref[:n]) <= 1e-9 * ref[:n] + 2e-11), (name, conv, ref)
    assert np.array_equal(mg.backend.natural("u", 0), op.state("u", 0))
    if cases.takes_block_solve(name):
        # the coarsest level took the time-parallel forward solve, the reference steps through it
        # (mgrit.py:459-486): residual history within 1e-10 relative + BLK_K eps ||u|| of the reference's, and of the SAME engine
        # with the step-by-step solve within BLK_K_FORM eps ||u|| (tests/cases.py states the constants)
        assert mg.backend.block_r[mg.lvl_max - 1] > 0
        floor = cases.EPS * cases.spacetime_norm(op.state("u", 0))
        assert np.all(np.abs(conv[:n] - ref[:n]) <= 1e-10 * ref[:n] + cases.BLK_K * floor), (name, conv, ref, floor)
        from pymgrit_amd.core.options import options
        try:
            options.coarse_solve = "sequential"
            prob2, tr2, _ = build_product(name)
            seq = Mgrit(prob2, transfer=tr2, logging_lvl=30, **opts).solve()["conv"]
        finally:
            options.reset("coarse_solve")
        assert len(seq) == len(conv) and np.all(np.abs(conv - seq) <= 1e-10 * seq + cases.BLK_K_FORM * floor), (name, conv, seq)
    for k, v in SOLVE[name].get("samples", {}).items():
        v = np.array(v)
        assert np.abs(mg.u[0][int(k)].get_values() - v).max() <= 1e-11 * max(1.0, np.abs(v).max())


def test_reference_unit_test_conv_golden_on_gpu():
    """reference tests/core/test_mgrit.py:59-70 re-run against the GPU engine with the reference's own literals"""
    _need_gpu()
    from pymgrit_amd import Mgrit
    prob, tr, c = build_product("heat_nx5_test_mgrit")
    res = Mgrit(prob, cf_iter=1, nested_iteration=True, max_iter=2, random_init_guess=False, logging_lvl=30).solve()
    np.testing.assert_almost_equal(np.array([0.00267692, 0.00018053]), res["conv"])


def test_one_level_equals_time_stepping():
    """reference tests/core/test_mgrit.py:72-84: a 1-level hierarchy is sequential time stepping"""
    _need_gpu()
    from pymgrit_amd import Mgrit
    heat0 = heat_problem(5, [cases.lin(2, 65)], x_end=2.0)[0]
    mg = Mgrit([heat0], cf_iter=1, nested_iteration=True, max_iter=2, logging_lvl=30)
    res = mg.solve()
    assert len(res["conv"]) == 0
    cur = heat0.vector_t_start
    for i in range(1, len(heat0.t)):
        cur = heat0.step(u_start=cur, t_start=heat0.t[i - 1], t_stop=heat0.t[i])
        np.testing.assert_almost_equal(mg.u[0][i].get_values(), cur.get_values())


def test_auto_detected_separable_forcing():
    """a plain callable rhs(x,t) (reference signature) is accepted when it is rank-one separable"""
    _need_gpu()
    from pymgrit_amd import Heat1D, Mgrit
    grids = [cases.lin(2, 65), cases.lin(2, 17), cases.lin(2, 5)]
    prob = [Heat1D(x_start=0, x_end=1, nx=33, a=1, init_cond=cases.init_cond, rhs=cases.rhs, t_interval=t) for t in grids]
    conv = Mgrit(prob, tol=1e-9, max_iter=8, logging_lvl=30).solve()["conv"]
    ref = np.array(SOLVE["heat_nx33_V_nested"]["conv"])
    assert np.all(np.abs(conv - ref) <= 1e-9 * ref + 2e-11)


def test_full_size_properties_config3(oracle):
    """BASELINE config 3 at FULL size (nx=16384, nt=65537: 8.6 GB level-0 slab, 2 levels m=4 here): size-independent
    properties
      * F-relax is idempotent bit for bit and leaves zero residual at F-points;
      * with zero forcing Phi is linear: scaling the state by 2^k scales the result exactly;
      * a sampled set of intervals equals the oracle bit for bit."""
    _need_gpu()
    from pymgrit_amd import Mgrit
    nt = 65537
    grids = [cases.lin(2.0 * (nt - 1) / 65536, nt), cases.lin(2.0 * (nt - 1) / 65536, (nt - 1) // 4 + 1)]
    prob = heat_problem(16384, grids, forcing=False)
    mg = Mgrit(prob, nested_iteration=False, logging_lvl=30)
    U = mg.backend.U[0]
    n = 16382
    perm = mg.backend.perm[0]
    gen = torch.Generator(device="cpu").manual_seed(5)
    cvals = torch.randn((len(grids[1]), n), generator=gen, dtype=torch.float64)

    def set_c(vals):
        rows = torch.zeros((vals.shape[0], U.shape[1]), dtype=torch.float64, device=U.device)
        rows[:, perm] = vals.to(U.device)
        U[::4] = rows

    set_c(cvals)
    mg.f_relax(0)
    first = U.clone()
    mg.f_relax(0)
    assert torch.equal(first, U), "F-relax must be idempotent"
    # zero residual at F-points: run the residual kernel on F-points through the C ABI run list
    fpts = [int(i) for i in np.sort(mg.index_local_f[0])][:4096]
    assert max(mg.backend.residual_norms(fpts)) == 0.0
    # linearity under exact scaling
    set_c(cvals * 8.0)
    mg.f_relax(0)
    assert torch.equal(first * 8.0, U)
    # sampled intervals vs oracle
    spec = cases.heat_level_spec(16384, grids[0], forcing=False)
    op = oracle.OracleProblem([spec, cases.heat_level_spec(16384, grids[1], forcing=False)], variant=1,
                              nested_iteration=False)
    for c_idx in (0, 8191, 16383):
        x = (cvals[c_idx] * 8.0).numpy()
        for k in range(1, 4):
            x = op.phi(0, 4 * c_idx + k, x)
            assert np.array_equal(U[4 * c_idx + k][perm].cpu().numpy(), x), (c_idx, k)


def test_unsupported_sizes_fail_loudly():
    """states beyond the engine's limits (Heat1D: n > 65536; see tests/test_hip_wide.py for 16384 < n <= 65536) are rejected with
    an error, never computed elsewhere"""
    _need_gpu()
    from pymgrit_amd import Mgrit
    from pymgrit_amd.core.hip_lib import MgritHipError
    prob = heat_problem(70000, [cases.lin(2, 9), cases.lin(2, 3)])
    with pytest.raises(MgritHipError):
        Mgrit(prob, logging_lvl=30)
