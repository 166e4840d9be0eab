"""Exchange fuzz against the LIVE reference (this container only: imports /root/reference and the thread-backed MPI stand-in):
python tests/golden/fuzz_live.py LO HI runs tests/fuzz_cases.py seeds LO..HI-1 on the reference and on pymgrit_amd at the same
rank count and reports every difference (solution values bit for bit, stopping values to 1e-14). Cases in which the reference
trips over one of its two defects are skipped and counted. Round 1: seeds 7200..9300, 0 differences,
24 skipped. Nothing here is imported by the product or by the tests."""
import sys, os
sys.path.insert(0, '/root/repo/tests/golden/_mpi_stub'); sys.path.insert(0, '/root/reference/src')
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import numpy as np, logging, warnings
warnings.filterwarnings("ignore")
logging.disable(logging.WARNING)
from mpi4py import MPI
from pymgrit.core.mgrit import Mgrit as RefMgrit
from pymgrit.dahlquist.dahlquist import Dahlquist as RefD
from fuzz_cases import random_case
from mock_comm import run_ranks
from pymgrit_amd import Mgrit, Dahlquist
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = skipped = 0
class Lab:
    def __init__(s, d, l): s.data, s.lvl = d, l
crossed = []
rs, rr = RefMgrit.send, RefMgrit.receive
def send(self, data, dest, lvl, op_id): rs(self, Lab(data, lvl), dest, lvl, op_id)
def receive(self, source, lvl, op_id):
    g = rr(self, source, lvl, op_id)
    if g.lvl != lvl: crossed.append(1)
    return g.data
RefMgrit.send, RefMgrit.receive = send, receive
for seed in range(lo, hi):
    grids, opts, size, depth = random_case(seed)
    size = min(size, len(grids[0]))
    del crossed[:]
    def one(rank):
        m = RefMgrit(problem=[RefD(t_interval=np.asarray(g)) for g in grids], logging_lvl=50, **opts)
        info = m.solve()
        scr = False
        for order in m.index_local_f:
            seen = set(); mem = set(int(i) for i in order)
            for i in (int(i) for i in order):
                scr |= (i - 1 in mem and i - 1 not in seen); seen.add(i)
        return info["conv"], [float(m.u[0][int(i)].get_values()) for i in m.index_local[0]], scr
    try:
        ref = MPI.run_world(size, one, timeout=60)
    except Exception as e:
        print("seed", seed, "reference failed:", repr(e)[:100]); skipped += 1; continue
    if crossed or any(r[2] for r in ref):
        skipped += 1; continue
    def target(comm):
        mg = Mgrit([Dahlquist(t_interval=np.asarray(g)) for g in grids], comm_time=comm, logging_lvl=50, pipeline_depth=depth, **opts)
        conv = mg.solve()["conv"]
        return conv, [float(mg.u[0][int(i)].get_values()) for i in mg.index_local[0]]
    try:
        res = run_ranks(size, target, timeout=60)
    except BaseException as e:
        print("seed", seed, "MINE FAILED", repr(e)[:200], [len(g) for g in grids], opts, size, depth); bad += 1; continue
    for r, ((c, u), (cr, ur, _)) in enumerate(zip(res, ref)):
        if len(c) != len(cr) or not np.allclose(c, cr, rtol=1e-14, atol=0) or not np.array_equal(u, ur):
            print("seed", seed, "MISMATCH rank", r, c, cr, [len(g) for g in grids], opts, size, depth); bad += 1; break
print("checked", hi - lo, "bad", bad, "skipped (reference defects)", skipped)
