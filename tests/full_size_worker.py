"""Worker of tests/test_hip_full_size.py: Mgrit.solve() on BASELINE configs[2] at full size (heat_1d nx=16384, nt=65537, 3-level
m=4) on one rank or sharded over the ranks of torch.distributed.run (gloo transport, all ranks on GPU 0). Each rank writes
its residual history and hashes of sampled level-0 states to its OWN file <out_dir>/rank<r>.json (the ranks of
torch.distributed.run share one stdout pipe: lines of two ranks can interleave there)."""
import os, sys, hashlib, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
torch.cuda.set_device(0)
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("gloo")
from pymgrit_amd import Heat1D, Mgrit
nts = (65537, 16385, 4097)
prob = [Heat1D(x_start=0, x_end=1, nx=16384, a=1, init_cond=bench.init_cond, rhs_separable=[(bench.rhs_space, bench.rhs_time)],
               t_start=0, t_stop=2, nt=nt) for nt in nts]
mg = Mgrit(prob, cf_iter=1, cycle_type='V', nested_iteration=True, max_iter=4, tol=1e-10, logging_lvl=30)
conv = mg.solve()["conv"]
h = hashlib.sha256()
own = [int(i) for i in mg.index_local[0]]
u = mg.backend.U[0]
for i in own[:: max(1, len(own) // 64)]:
    h.update(u[i].cpu().numpy().tobytes())
part = {"rank": rank, "conv": [float(c) for c in conv], "first": own[0], "n": len(own), "u_last": hashlib.sha256(u[own[-1]].cpu().numpy().tobytes()).hexdigest()}
out_dir = sys.argv[1]
tmp = os.path.join(out_dir, f"rank{rank}.json.tmp")
with open(tmp, "w") as fh:
    json.dump(part, fh)
os.replace(tmp, os.path.join(out_dir, f"rank{rank}.json"))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
