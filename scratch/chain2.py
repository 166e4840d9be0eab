import sys, os, shutil
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from pymgrit_amd.core import hip_lib
if len(sys.argv) > 1:
    hip_lib.LIB_PATH = os.path.join(root, "scratch", sys.argv[1])
import numpy as np, torch
from scratch.chain import one_level
from scratch.gap import timeit
for nx in (16384, 4096):
    mg = one_level(4097, nx, True)
    ms = timeit(lambda: mg.forward_solve(0), 3)
    print(f"{sys.argv[1:]} chain no-g nx={nx}: {ms:.2f} ms = {ms/4096*1e3:.2f} us/step")
