"""The N>1 path on CPU: world_size 2/3/5 torch.distributed jobs (gloo, 127.0.0.1) running the SAME Mgrit host logic and
exchange schedule as the multi-GPU runs, on the plugin path. The reference is bit-identical for every P with uniform
coarsening; so must we be: residual history and every solution vector equal the single-rank run
bit for bit. The HIP twin of this test (two ranks sharing one GPU, gloo transport) lives in test_hip_distributed.py."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


_ONE_RANK = {}     # results of one-rank launches, the baseline several tests compare their sharded runs with


def launch(world, case, mode="plugin", backend="gloo", timeout=600, depth=None, slow_rank=None, no_groups_rank=None,
           per_rank=False):
    # a one-rank run is a pure function of (case, mode, lag, the library's / package's environment switches): computed once per
    # session (every launch is a fresh process: a second or two of start-up each on the GPU box)
    memo = None
    if world == 1 and slow_rank is None and no_groups_rank is None:
        memo = (case, mode, depth, per_rank, tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith(("MGRIT_", "PYMGRIT_")))))
        if memo in _ONE_RANK:
            return _ONE_RANK[memo]
    res = _launch(world, case, mode, backend, timeout, depth, slow_rank, no_groups_rank, per_rank)
    if memo is not None:
        _ONE_RANK[memo] = res
    return res


def _launch(world, case, mode, backend, timeout, depth, slow_rank, no_groups_rank, per_rank):
    out = tempfile.mkdtemp(prefix=f"mgrit_{case}_{world}_")
    port = free_port()
    env = dict(os.environ)
    env.pop("MGRIT_TEST_PIPELINE_DEPTH", None)
    env.pop("MGRIT_TEST_SLOW_RANK", None)
    if depth is not None:
        env["MGRIT_TEST_PIPELINE_DEPTH"] = str(depth)
    if slow_rank is not None:
        env["MGRIT_TEST_SLOW_RANK"] = str(slow_rank)
    env.pop("PYMGRIT_AMD_NO_EXTRA_GROUPS", None)
    if no_groups_rank is not None:
        env["PYMGRIT_AMD_NO_EXTRA_GROUPS"] = "1"
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(r), str(world), str(port), case,
                               mode, out, backend], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
             for r in range(world)]
    logs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=timeout)
            logs.append(o.decode(errors="replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    bad = [r for r, p in enumerate(procs) if p.returncode != 0]
    if bad:
        # a rank that only saw its peer disappear is an echo of the real failure: show the root cause first
        root = [r for r in bad if "Connection closed by peer" not in logs[r]] or bad
        raise AssertionError(f"ranks {bad} failed; rank {root[0]}:\n{logs[root[0]][-3000:]}")
    res = [np.load(os.path.join(out, f"rank{r}.npz")) for r in range(world)]
    if per_rank:     # local criteria: every rank has its own residual history
        return [(r["conv"], r["u"]) for r in res]
    conv = res[0]["conv"]
    for r in res[1:]:
        assert np.array_equal(r["conv"], conv), "every rank must hold the same residual history"
    t = np.concatenate([r["t"] for r in res])
    u = np.concatenate([r["u"] for r in res if r["u"].size], axis=0)
    assert np.all(np.diff(t) > 0), "owned blocks must tile the time grid in rank order"
    return conv, u


CASES = [
    ("dahlquist_config1", [2, 3]),              # nt=101, m=2: aligned and unaligned splits
    ("dahlquist_F", [2, 3]),                    # 4 levels, F-cycle
    ("dahlquist_procs_without_points", [5]),    # first factor 16: ranks that own no coarse point
    ("heat_nx33_V_nested", [2, 3]),             # nt=65/17/5 m=4: P=3 exercises comm_front/comm_back (op 1) and ops 2/3/7
    ("heat_nx33_F_nonested", [3]),
    ("heat_nx33_V_jump", [2]),                  # conv_crit=1
    ("heat_spatial_coarsening", [2]),
    ("heat_nx33_procs_without_points", [5]),
    ("bdf:bdf2_example_small", [3]),
]


@pytest.mark.parametrize("case,sizes", CASES, ids=[c for c, _ in CASES])
def test_multi_rank_equals_single_rank(case, sizes):
    conv1, u1 = launch(1, case)
    for world in sizes:
        conv, u = launch(world, case)
        assert np.array_equal(conv, conv1), (case, world, conv, conv1)
        assert np.array_equal(u, u1), (case, world, np.abs(u - u1).max())


PIPELINE_CASES = [   # (case, world, depth): stops by tolerance in the middle of the run, by max_iter, F-cycles, depth 0 = no lag
    ("dahlquist_config1", 3, 0), ("dahlquist_config1", 3, 1), ("dahlquist_config1", 2, 6),
    ("heat_nx33_V_nested", 3, 0), ("heat_nx33_V_nested", 3, 1), ("heat_nx33_F_nonested", 2, 2),
    ("heat_nx5_test_mgrit", 2, 4),   # max_iter = 2: the run ends before the first stopping value would be looked at
]


@pytest.mark.parametrize("case,world,depth", PIPELINE_CASES, ids=[f"{c}-P{w}-d{d}" for c, w, d in PIPELINE_CASES])
def test_pipelined_solve_is_bit_identical_for_every_depth(case, world, depth):
    """the stopping value may lag `depth` iterations (Mgrit._solve_pipelined): same residual history, same iteration count,
    same solution as one rank, whatever the depth -- including the rollback when tol was met `depth` iterations ago"""
    conv1, u1 = launch(1, case)
    conv, u = launch(world, case, depth=depth)
    assert np.array_equal(conv, conv1), (conv, conv1)
    assert np.array_equal(u, u1)


@pytest.mark.parametrize("slow_rank", [0, 1, 2])
def test_pipelined_solve_with_a_rank_far_behind(slow_rank):
    """one rank sleeps in every sweep: its neighbours run ahead by up to `depth` iterations, several messages per link
    are in flight, stopping values arrive late -- same results as one rank"""
    conv1, u1 = launch(1, "heat_nx33_V_nested")
    conv, u = launch(3, "heat_nx33_V_nested", depth=3, slow_rank=slow_rank)
    assert np.array_equal(conv, conv1) and np.array_equal(u, u1)


def test_fallback_to_the_main_group_alone():
    """no per-link / side groups (PYMGRIT_AMD_NO_EXTRA_GROUPS): every exchange on the main group, plain solve loop"""
    conv1, u1 = launch(1, "heat_nx33_V_nested")
    conv, u = launch(3, "heat_nx33_V_nested", no_groups_rank=1, timeout=120)
    assert np.array_equal(conv, conv1) and np.array_equal(u, u1)
