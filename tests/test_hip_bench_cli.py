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
    assert set(b["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and b["roofline"]["bound"] == "hbm"
    assert set(b["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and b["cpu_baseline"]["kind"] == "port"
    assert b["iters_to_tol"]["sample_nt1025"]["gpu_iters"] == b["iters_to_tol"]["sample_nt1025"]["cpu_iters"]
    assert "workload" in b["config"]


def test_bench_two_ranks_line():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--nx", "1024",
                        "--nt", "2049", "--steps", "3", "--warmup", "1", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    b = _last_json(r.stdout)
    assert b["n_gpus"] == 2 and b["config"]["pipeline_depth"] == 4 and b["value"] > 0 and b["scaling"] == "strong"
    assert b["fcf_relax_level0"]["ms"] > 0
