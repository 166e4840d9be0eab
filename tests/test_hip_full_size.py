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


def run(world):
    cmd = [sys.executable, os.path.join(HERE, "full_size_worker.py")]
    if world > 1:
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(HERE, "full_size_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    res = [json.loads(line.split("RESULT", 1)[1]) for line in out.stdout.splitlines() if "RESULT" in line]
    assert len(res) == world, out.stdout[-2000:] + out.stderr[-2000:]
    return sorted(res, key=lambda r: r["rank"])


def test_full_size_solve_sharded_equals_one_rank():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    one = run(1)[0]
    assert len(one["conv"]) == 3 and one["conv"][-1] < 1e-10 < one["conv"][-2], one["conv"]
    for world in (2, 4):
        parts = run(world)
        for r in parts:
            assert r["conv"] == one["conv"], (world, r["conv"], one["conv"])
        assert sum(r["n"] for r in parts) == one["n"] and all(r["first"] == 1 for r in parts[1:])   # local slot 0 = ghost point
        assert parts[-1]["u_last"] == one["u_last"], world      # the state at the final time, bit for bit
