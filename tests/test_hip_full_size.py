"""BASELINE configs[2] at FULL size (heat_1d nx=16384, nt=65537, 3-level m=4; 16.6 GB of slabs) through a size-independent
property: the solve sharded over two and over four ranks (pipelined loop with its rollback, hand-over of the time-parallel
coarsest-level solve's mode amplitudes across the rank boundary) produces the residual history and the final time point of the
one-rank solve bit for bit, and converges like the bounded sample that is compared with the oracle elsewhere (three cycles to
1e-10); the timed paths of configs 3 and 5 at full size, and of config 4 at 256 x 256, cycle by cycle against the oracle."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run(world, out_dir):
    out_dir = os.path.join(str(out_dir), f"world{world}")
    os.makedirs(out_dir)
    cmd = [sys.executable, os.path.join(HERE, "full_size_worker.py"), out_dir]
    if world > 1:
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(HERE, "full_size_worker.py"), out_dir]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    res = []
    for r in range(world):      # one file per rank: nothing is parsed from the shared stdout
        path = os.path.join(out_dir, f"rank{r}.json")
        assert os.path.exists(path), (r, out.stdout[-2000:] + out.stderr[-2000:])
        with open(path) as fh:
            res.append(json.load(fh))
    assert [r["rank"] for r in res] == list(range(world))
    return res


def test_full_size_solve_sharded_equals_one_rank(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    one = run(1, tmp_path)[0]
    assert len(one["conv"]) == 3 and one["conv"][-1] < 1e-10 < one["conv"][-2], one["conv"]
    for world in (2, 4):
        parts = run(world, tmp_path)
        for r in parts:
            assert r["conv"] == one["conv"], (world, r["conv"], one["conv"])
        assert sum(r["n"] for r in parts) == one["n"] and all(r["first"] == 1 for r in parts[1:])   # local slot 0 = ghost point
        assert parts[-1]["u_last"] == one["u_last"], world      # the state at the final time, bit for bit


def test_full_size_planned_cycles_match_the_oracle(oracle, monkeypatch):
    """the path bench.py times -- config 3 at FULL size, the default cycle (program order, replayed as one hipGraph from its third
    execution on), whole-level passes, C-point storage, pre-relaxed C-points, the time-parallel coarsest-level solve (DESIGN.md
    3.8: 256 blocks of 16 steps, 50 sine modes) -- against the ORACLE running the same cycles at the same size: per-point
    residual norms of every cycle and sampled level-0 states, bit for bit. Needs ~20 GB of host memory for the oracle's slabs: a
    host without them FAILS the test (the only full-size comparison of the timed path must not vanish from a green suite)."""
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    monkeypatch.setenv("PYMGRIT_AMD_PLAN_GRAPH", "1")   # the one-block cycle replayed as a hipGraph (opt-in since round 4) must give the same bits
    import cases
    try:
        free_gb = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") / 2 ** 30
    except (ValueError, OSError):
        free_gb = 0.0
    if free_gb < 24:
        pytest.fail(f"host has {free_gb:.0f} GB free: the full-size oracle needs its own 17 GB of slabs")
    import bench
    from pymgrit_amd import Heat1D, Mgrit
    nx, nt0 = 16384, 65537
    t0 = np.linspace(0, 2.0 * (nt0 - 1) / 65536, nt0)
    grids = [t0, t0[::4], t0[::16]]
    prob = [Heat1D(x_start=0, x_end=1, nx=nx, a=1, init_cond=bench.init_cond, rhs_separable=[(bench.rhs_space, bench.rhs_time)],
                   t_interval=g) for g in grids]
    mg = Mgrit(prob, cf_iter=1, cycle_type='V', nested_iteration=False, max_iter=8, tol=0.0, logging_lvl=30)
    assert mg.backend.block_r[2] == 50 and mg._level_intervals(0) is not None
    op = oracle.OracleProblem([cases.heat_level_spec(nx, g) for g in grids], variant=1, cf_iter=1, nested_iteration=False, max_iter=8,
                              tol=0.0)
    assert op.set_threads(cases.usable_cpus(32)) >= 1      # (same bits for every thread count: tests/test_oracle_golden.py)
    sample = [1, 2, 3, 4, 16381, 16384, 32768, 40001, 65533, 65535, 65536]
    for it in range(4):        # iteration 0, then three steady cycles: the last one is the capture and its first replay
        mg.iteration(lvl=0, cycle_type='V', iteration=it, first_f=True)
        got = np.asarray(mg.compute_residual())
        op.iteration(0, 'V', it, True)
        want = op.residual_norms()
        assert np.array_equal(got, want), (it, float(np.abs(got - want).max()))
    assert any(p is not None and getattr(p, "_hip", {}).get("graph") is not None for p in mg._plans.values())
    ref = op.state("u", 0)
    for i in sample:
        assert np.array_equal(np.asarray(mg.u[0][i].get_values()), ref[i]), i


def test_full_size_both_forms_of_the_coarse_solve():
    """config 3 at FULL size with the coarsest level's forward_solve in BOTH forms: the default time-parallel blocks (DESIGN.md
    3.8) and the reference's step-by-step loop (mgrit.py:459-486; options.coarse_solve = 'sequential'), five iterations from the
    nested-iteration start. Per iteration the residual norms agree within 1e-10 relative + 2 eps ||u|| (tests/cases.py BLK_K_FORM);
    the deviations are recorded (gpurun_out/parity_config3_forms.json -> profiles/) and bench.py prints the same record as `parity`."""
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    import bench
    import cases
    from pymgrit_amd import Heat1D
    nx, nt0 = 16384, 65537
    t0 = np.linspace(0, 2.0 * (nt0 - 1) / 65536, nt0)
    prob = [Heat1D(x_start=0, x_end=1, nx=nx, a=1, init_cond=bench.init_cond, rhs_separable=[(bench.rhs_space, bench.rhs_time)],
                   t_interval=g) for g in (t0, t0[::4], t0[::16])]
    rec = bench.coarse_solve_parity(prob, iters=5, cf_iter=1, cycle_type='V', nested_iteration=True)
    print("parity of the two forms:", json.dumps(rec))
    out_dir = os.path.join(os.path.dirname(HERE), "gpurun_out")
    if os.path.isdir(out_dir) and os.access(out_dir, os.W_OK):
        with open(os.path.join(out_dir, "parity_config3_forms.json"), "w") as fh:
            json.dump(rec, fh, indent=1)
    assert rec["coarse_solve"].startswith("time-parallel") and len(rec["conv"]) == len(rec["conv_sequential"]) == 5
    assert cases.BLK_K_FORM == 2 and rec["within_bound"], rec


def test_full_size_config5_cycles_match_the_oracle(oracle, monkeypatch):
    """BASELINE configs[4] at FULL size -- advection_1d 8192 DOF, nt = 32769, 4 levels m = 2, periodic spatial coarsening on the first
    two level pairs, F-cycle -- as bench.py --workload advection times it: the general whole-level passes on all three level
    pairs, one graph per cycle, the time-parallel coarsest-level solves (all 2048 Fourier modes, 256 blocks) against the oracle at
    the same size: per-point residual norms of every cycle and sampled level-0 states, bit for bit (5 GB of oracle slabs)."""
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    monkeypatch.setenv("PYMGRIT_AMD_PLAN_GRAPH", "1")   # the one-block cycle replayed as a hipGraph (opt-in since round 4) must give the same bits
    import cases
    from pymgrit_amd import Advection1D, GridTransferAdvection, GridTransferCopy, Mgrit
    nt0, nxs = 32769, [8193, 4097, 2049, 2049]
    t0 = np.linspace(0, 2, nt0)
    grids = [t0[::2 ** k] for k in range(4)]
    prob = [Advection1D(c=1, x_start=-1, x_end=1, nx=n, t_interval=g) for n, g in zip(nxs, grids)]
    mg = Mgrit(prob, transfer=[GridTransferAdvection(), GridTransferAdvection(), GridTransferCopy()], cf_iter=1, cycle_type='F',
               nested_iteration=False, max_iter=8, tol=0.0, logging_lvl=30)
    assert mg.backend.block_r[3] == 2048 and all(mg._gen_intervals(lvl) is not None for lvl in range(3))
    op = oracle.OracleProblem([cases.advection_level_spec(n, g) for n, g in zip(nxs, grids)], transfer=[2, 2, 0], variant=1, cf_iter=1,
                              cycle_type='F', nested_iteration=False, max_iter=8, tol=0.0)
    op.set_threads(cases.usable_cpus(32))      # (takes effect on hierarchies of copy transfers only: 1 here)
    for it in range(4):        # iteration 0, then three steady cycles: the last one is the capture and its first replay
        mg.iteration(lvl=0, cycle_type='F', iteration=it, first_f=True)
        got = np.asarray(mg.compute_residual())
        op.iteration(0, 'F', it, True)
        want = op.residual_norms()
        assert np.array_equal(got, want), (it, float(np.abs(got - want).max()))
    assert any(p is not None and getattr(p, "_hip", {}).get("graph") is not None for p in mg._plans.values())
    ref = op.state("u", 0)
    for i in [1, 2, 3, 8191, 8192, 16384, 20001, 32766, 32767, 32768]:
        assert np.array_equal(np.asarray(mg.u[0][i].get_values()), ref[i]), i


def test_config4_timed_path_matches_the_oracle(oracle):
    """BASELINE configs[3]'s timed path at a size the oracle finishes in minutes: Heat2D 256 x 256, backward Euler, 2 levels m = 8,
    nt = 513 (64 coarsest steps = 4 blocks): batched MFMA sweeps in program order and the time-parallel coarsest-level solve on the
    full sine spectrum against the oracle: per-point residual norms and states bit for bit over two cycles. (At 512 x 512 one Phi
    of the oracle's plain loops takes half a second; tests/test_hip_heat2d.py holds the 512 x 512 property checks.)"""
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    import cases
    from pymgrit_amd import Mgrit
    oracle.set_h2d_threads(cases.usable_cpus(16))
    ts = cases.h2d_grids([513, 65])
    prob = [cases.h2d_app(256, 256, t, "BE", True) for t in ts]
    mg = Mgrit(prob, nested_iteration=False, max_iter=4, tol=0.0, logging_lvl=30)
    assert mg.backend.block_r[1] == 254 * 254
    op = oracle.OracleProblem([cases.h2d_level_spec(a) for a in prob], nested_iteration=False, max_iter=4, tol=0.0)
    for it in range(2):
        mg.iteration(lvl=0, cycle_type='V', iteration=it, first_f=True)
        got = np.asarray(mg.compute_residual())
        op.iteration(0, 'V', it, True)
        want = op.residual_norms()
        assert np.array_equal(got, want), (it, float(np.abs(got - want).max()))
    for lvl in (0, 1):
        assert np.array_equal(mg.backend.natural("u", lvl), op.state("u", lvl)), lvl
