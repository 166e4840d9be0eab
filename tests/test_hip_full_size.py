"""BASELINE configs[2] at FULL size (heat_1d nx=16384, nt=65537, 3-level m=4; 16.6 GB of slabs) through a size-independent
property: the solve sharded over two and over four ranks (pipelined loop with its rollback, hand-over of the overlapped coarsest-level
chain across the rank boundary) produces the residual history and the final time point of the one-rank solve bit for bit,
and converges like the bounded sample that is compared with the oracle elsewhere (three cycles to 1e-10)."""
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
