"""Build a uniform time-level hierarchy from the finest problem (reference
src/pymgrit/core/simple_setup_problem.py:14-43): level k is a deep copy of the finest Application whose grid is
every ``coarsening``-th point of level k-1."""
import copy
import warnings
from typing import List

from pymgrit_amd.core.application import Application


def simple_setup_problem(problem: Application, level: int, coarsening: int) -> List[Application]:
    if len(problem.t[::coarsening * level]) == 1:
        warnings.warn(
            "This choice leads to a coarsest grid with only one time point, which is the initial point. "
            "It is recommended to choose a structure with at least two points on the coarsest grid.")
    hierarchy = [problem]
    for _ in range(1, level):
        grid = hierarchy[-1].t[::coarsening]
        coarse = copy.deepcopy(problem)
        coarse.t, coarse.nt = grid, len(grid)
        coarse.t_start, coarse.t_end = grid[0], grid[-1]
        hierarchy.append(coarse)
    return hierarchy
