"""GPU twin of test_distributed.py: the time-sharded HIP path with TWO (or three) ranks sharing the one GPU of the test
box, ghost rows exchanged over torch.distributed (gloo transport here, staged through the host; on a multi-GPU node the
same schedule runs over RCCL). Sharded runs must equal the single-rank GPU run bit for bit."""
import numpy as np
import pytest

from test_distributed import launch

pytestmark = pytest.mark.gpu

# at most 4 ranks: the GPU box allows six processes on the card at once and the test process itself holds it too

CASES = [("heat_nx33_V_nested", [2, 3]), ("heat_nx257_nt257", [2]), ("heat_nx33_F_nonested", [3]),
         ("heat_nx33_V_jump", [2]), ("heat_spatial_coarsening", [2]), ("advection_3lvl_F", [2]),
         ("h2d:be_3lvl_F_bc", [2, 3, 4]), ("h2d:cn_2lvl", [2]), ("advsc:adv_sc_F", [3]),
         ("heat_nx33_procs_without_points", [4]), ("bdf:bdf2_example_small", [2, 3]), ("bdf:bdf2_weighted_jump", [3])]   # first factor 16: ranks that own no coarse point at all


@pytest.mark.parametrize("case,sizes", CASES, ids=[c for c, _ in CASES])
def test_sharded_hip_equals_single_rank(case, sizes):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    conv1, u1 = launch(1, case, mode="hip")
    for world in sizes:
        conv, u = launch(world, case, mode="hip", backend="gloo")
        assert np.array_equal(conv, conv1), (case, world, conv, conv1)
        assert np.array_equal(u, u1), (case, world, np.abs(u - u1).max())


@pytest.mark.parametrize("case,world,depth", [("heat_nx33_V_nested", 3, 1), ("heat_nx33_V_nested", 2, 0),
                                              ("h2d:be_3lvl_F_bc", 2, 2), ("bdf:bdf2_example_small", 3, 4)])
def test_pipelined_hip_solve_is_bit_identical(case, world, depth):
    """stopping value examined `depth` iterations late on the HIP path: C-point snapshots / rollback inside HBM"""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    conv1, u1 = launch(1, case, mode="hip")
    conv, u = launch(world, case, mode="hip", backend="gloo", depth=depth)
    assert np.array_equal(conv, conv1), (conv, conv1)
    assert np.array_equal(u, u1)


@pytest.mark.parametrize("case,world", [("heat_nx2050_wide", 3), ("heat_nx1500_wide_F", 2)])
def test_wide_states_across_ranks(case, world, monkeypatch):
    """states wider than one group of 1024 values: the coarsest level runs the overlapped chain, whose
    running form is carried across a rank boundary (the hand-over of op 5 takes the chain's state along with the last
    point), so sharded runs stay bit-identical to the one-rank run; likewise with the plain per-step chain
    (MGRIT_HIP_CHAIN_PLAIN=1), and the two forms agree to rounding."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    monkeypatch.delenv("MGRIT_HIP_CHAIN_PLAIN", raising=False)
    conv1, u1 = launch(1, case, mode="hip")
    conv, u = launch(world, case, mode="hip", backend="gloo")
    assert np.array_equal(conv, conv1), (conv, conv1)
    assert np.array_equal(u, u1), np.max(np.abs(u - u1))
    conv_d, u_d = launch(world, case, mode="hip", backend="gloo", depth=2)     # pipelined loop, rollback included
    assert np.array_equal(conv_d, conv1) and np.array_equal(u_d, u1)
    monkeypatch.setenv("MGRIT_HIP_CHAIN_PLAIN", "1")
    conv1p, u1p = launch(1, case, mode="hip")
    convp, up = launch(world, case, mode="hip", backend="gloo")
    assert np.array_equal(convp, conv1p) and np.array_equal(up, u1p)
    # rounding differences between the two forms enter with the conditioning of the implicit step (eps * cond ~ 1e-10 here)
    assert np.allclose(conv1p, conv1, rtol=1e-10, atol=1e-12) and np.max(np.abs(u1p - u1)) <= 1e-11 * max(1.0, np.max(np.abs(u1)))


@pytest.mark.parametrize("case,world", [("heat_nx257_nt257", 2), ("heat_nx2050_wide", 3), ("heat_spatial_coarsening", 2)])
def test_sharded_hip_matches_the_oracle(oracle, case, world):
    """several ranks on the HIP path against the ORACLE (not only against the one-rank HIP run): residual history within 1e-10
    relative per iteration, level-0 solution bit for bit"""
    import torch
    import cases
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    c = cases.solve_cases()[case]
    conv, u = launch(world, case, mode="hip", backend="gloo")
    op = oracle.OracleProblem(c["levels"], transfer=c.get("transfer"), variant=1, **dict(c["opts"]))
    oconv = op.solve()
    assert len(conv) == len(oconv) and np.max(np.abs(conv - oconv) / oconv) <= 1e-10, (conv, oconv)
    assert np.array_equal(u, op.state("u", 0))


@pytest.mark.parametrize("case,plain", [("advection_nx2049_wide", False), ("advection_nx4000_wide_F", False), ("heat_nx2050_wide", True),
                                        ("heat_nx1500_wide_F", True)])
def test_chain_inside_one_workgroup_is_bit_identical(oracle, case, plain, monkeypatch):
    """states of 2..4 groups: the coarsest-level chain with all its workers in ONE workgroup and the group totals exchanged
    through LDS (chain_local_kernel, the default) gives the bits of the chain with one workgroup per group and the totals
    exchanged through L2 (MGRIT_HIP_CHAIN_LOCAL_G=0) -- and of the oracle; on two ranks as well"""
    import torch
    import cases
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    if plain:
        monkeypatch.setenv("MGRIT_HIP_CHAIN_PLAIN", "1")     # Heat1D: the per-step form (the overlapped chain is a kernel of its own)
    monkeypatch.delenv("MGRIT_HIP_CHAIN_LOCAL_G", raising=False)
    conv, u = launch(1, case, mode="hip")
    conv2, u2 = launch(2, case, mode="hip", backend="gloo")
    monkeypatch.setenv("MGRIT_HIP_CHAIN_LOCAL_G", "0")
    conv0, u0 = launch(1, case, mode="hip")
    assert np.array_equal(conv, conv0) and np.array_equal(u, u0)
    assert np.array_equal(conv2, conv0) and np.array_equal(u2, u0)
    if not plain:
        c = {**cases.solve_cases(), **cases.extra_cases()}[case]
        op = oracle.OracleProblem(c["levels"], transfer=c.get("transfer"), variant=1, **dict(c["opts"]))
        oconv = op.solve()
        assert len(conv) == len(oconv) and np.max(np.abs(conv - oconv) / oconv) <= 1e-10, (conv, oconv)
        assert np.array_equal(u, op.state("u", 0))
