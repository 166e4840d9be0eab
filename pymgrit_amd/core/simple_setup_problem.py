"""Build a uniform time-level hierarchy from the finest problem (reference
src/pymgrit/core/simple_setup_problem.py:14-43): level k is a deep copy of the finest Application whose grid is
every ``coarsening``-th point of level k-1."""
import copy
import warnings
from typing import List

from pymgrit_amd.core.application import Application


def _with_grid(problem: Application, grid) -> Application:
    """independent copy of ``problem`` living on the time grid ``grid``"""
    clone = copy.deepcopy(problem)
    clone.t, clone.nt = grid, len(grid)
    clone.t_start, clone.t_end = grid[0], grid[-1]
    return clone


def simple_setup_problem(problem: Application, level: int, coarsening: int) -> List[Application]:
    """[problem, problem on t[::c], problem on t[::c][::c], ...] with ``level`` entries."""
    if len(problem.t[::coarsening * level]) == 1:  # same (odd) test and message as the reference
        warnings.warn("This choice leads to a coarsest grid with only one time point, which is the initial point. "
                      "It is recommended to choose a structure with at least two points on the coarsest grid.")
    grids = [problem.t]
    while len(grids) < level:
        grids.append(grids[-1][::coarsening])
    return [problem] + [_with_grid(problem, g) for g in grids[1:]]
