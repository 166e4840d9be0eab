"""bench.py end to end on small instances: one JSON line with the fields the harness reads (metric, value, roofline,
cpu_baseline, iters_to_tol) on one GPU, and through torch.distributed.run on two ranks (gloo, both on the one GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_line():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--nx", "1024", "--nt", "1025", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    b = _last_json(r.stdout)
    assert b["metric"].startswith("time-point-DOF updates/sec") and b["n_gpus"] == 1 and b["steps"] == 3 and b["dtype"] == "f64"
    assert b["value"] > 0 and b["higher_is_better"] is True and b["vs_baseline"] is None and b["data"] == "synthetic"
    assert set(b["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"} and b["roofline"]["bound"] == "hbm"
    assert b["sweeps"] and all(r["ms_per_cycle"] > 0 for r in b["sweeps"].values()) and b["cycle"]["algorithmic_bytes"] > 0
    assert set(b["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and b["cpu_baseline"]["kind"] == "port"
    assert b["iters_to_tol"]["sample_nt1025"]["gpu_iters"] == b["iters_to_tol"]["sample_nt1025"]["cpu_iters"]
    assert "workload" in b["config"]


def test_bench_two_ranks_line():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    # no launcher around it: `python bench.py --gpus 2` starts its own ranks (torch.distributed.run as a child process, before
    # anything touches the GPU) and relays rank 0's line
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--nx", "1024",
                        "--nt", "2049", "--steps", "3", "--warmup", "1", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    b = _last_json(r.stdout)
    assert b["n_gpus"] == 2 and b["config"]["pipeline_depth"] == 4 and b["value"] > 0 and b["scaling"] == "strong"
    assert b["fcf_relax_level0"]["ms"] > 0
    assert b["exchange"]["messages_per_cycle"] > 0 and b["exchange"]["backend"] == "gloo" and b["nccl_ranks"] == 0


def test_bench_four_ranks_line():
    """four real processes (gloo) on the one GPU, config 3's layout shrunk: every share ends on a block border of the coarsest level
    (257 points: 4 blocks of 16 steps per rank), so the time-parallel forward solve hands its amplitudes over across three rank
    boundaries; rank 0's line. (Six processes may use the card at once: the test runner, the launcher's four ranks.)"""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--nx", "1024",
                        "--nt", "4097", "--steps", "3", "--warmup", "1", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    b = _last_json(r.stdout)
    assert b["n_gpus"] == 4 and b["value"] > 0 and b["scaling"] == "strong" and b["exchange"]["messages_per_cycle"] > 0
    assert b["exchange"]["backend"] == "gloo" and b["nccl_ranks"] == 0


def test_bench_emulated_solve_line():
    """`--emulate-solve 8`: Mgrit.solve() end to end on eight loopback ranks (config 3's layout at nx = 1024, nt = 8193: aligned
    ranks, block-solve hand-over across seven boundaries), the residual history of the one-rank run bit for bit"""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--emulate-solve", "8", "--nx", "1024", "--nt", "8193"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    b = _last_json(r.stdout)
    assert b["emulated_ranks"] == 8 and b["every_rank_reports_the_same_history"] and b["equals_one_rank_history_bit_for_bit"]
    assert all(row["aligned"] and row["block_solve_sharded"] for row in b["ranks"]) and b["iterations"] >= 2


@pytest.mark.parametrize("extra,levels", [(["--nx", "1024", "--nt", "2049"], 3), (["--workload", "advection", "--nt-adv", "1025"], 4)])
def test_bench_emulated_ranks_line(extra, levels):
    """`--emulate-rank all/2`: both ranks of a two-rank run rehearsed on the one GPU (loopback exchange), configs[2] and configs[4]
    shrunk: aligned ranks, the cycle of a rank replayed as a graph, rows sent through the device exchange"""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--emulate-rank", "all/2", "--steps", "4", "--warmup", "1"] + extra,
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    b = _last_json(r.stdout)
    assert b["config"]["emulated_ranks"] == 2 and len(b["ranks"]) == 2 and b["all_ranks_on_one_gpu_ms_per_cycle"] > 0
    assert b["exchange"]["messages_per_cycle"] > 0
    for row in b["ranks"]:
        assert row["aligned"] and row["ms_per_cycle"] > 0 and len(row["local_points_by_level"]) == levels
    if levels == 4:      # spatial coarsening + Advection1D: the general whole-level passes on every level pair of every rank
        assert all(row["general_passes_on_levels"] == [0, 1, 2] for row in b["ranks"])


def test_bench_refuses_a_mismatched_launcher():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120,
                       cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stdout + r.stderr)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: the RCCL transport cannot share one device")
def test_bench_two_gpus_over_rccl():
    """the N > 1 line the driver's scaling run produces: ghost rows travel device to device over RCCL (nccl backend)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--nx", "4096", "--nt", "4097", "--steps", "3",
                        "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    b = _last_json(r.stdout)
    assert b["n_gpus"] == 2 and b["nccl_ranks"] == 2 and b["exchange"]["device_resident_messages"] == b["exchange"]["messages_total"]


def test_rccl_stack_comes_up_on_one_gpu(tmp_path):
    """what CAN be exercised of the RCCL transport on a one-GPU box: a one-rank nccl process group with the device bound at
    initialisation (eager communicator), a sub-group split off it, a collective on device memory, the time communicator on top
    of it -- the calls every rank of a sharded run makes before its first exchange (the exchanges themselves need two devices)"""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    script = tmp_path / "one_rank_nccl.py"
    script.write_text('''
import os, sys, socket
sys.path.insert(0, %r)
import torch, torch.distributed as dist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.ones(1024, dtype=torch.float64, device="cuda")
dist.all_reduce(x)
dist.barrier()
g = dist.new_group([0], backend="nccl")
dist.all_reduce(x, group=g)
side = dist.new_group([0], backend="gloo")
from pymgrit_amd.core.comm import resolve_comm
comm = resolve_comm(None)
comm.prepare()
assert comm.size == 1 and comm.backend == "nccl" and comm.allgather_object(3) == [3]
torch.cuda.synchronize()
assert float(x[0]) == 1.0
dist.destroy_process_group()
print("RCCL_OK")
''' % ROOT)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]
