"""AT-MGRIT (FAS form): MGRIT whose coarsest-level solve is truncated to local coarse grids of ``k`` points.

Drop-in for ``pymgrit.core.at_mgrit.AtMgrit`` (reference src/pymgrit/core/at_mgrit.py:16-249): same constructor
(``k`` first, global stopping criteria only), same attributes (``k``, ``local_coarse_grid``, ``comm_coarsest_level``,
``c_points_per_proc``). Instead of the sequential forward solve, every point of the coarsest level is recomputed from the
old value ``k-1`` points back by ``k-1`` steps; the points do not depend on each other, so on the MI355X engine the whole
level is one launch of independent workgroups (``mgrit_hip_at_solve``) -- no serial coarsest-level chain.
"""
import logging
import time

import numpy as np

from pymgrit_amd.core.mgrit import Mgrit


class AtMgrit(Mgrit):
    def __init__(self, k, conv_crit=0, *args, **kwargs):
        """:param k: distance (number of points) of the local coarse grids; the rest as for Mgrit"""
        self.k = k
        self.local_coarse_grid = None
        self.c_points_per_proc = None
        self.comm_coarsest_level = None
        if conv_crit not in [0, 1]:
            raise Exception('Local convergence criteria are not implemented for AT-MGRIT. Please select a global criterion.')
        super().__init__(conv_crit=conv_crit, *args, **kwargs)

    def pipeline_depth(self) -> int:
        return 0   # the coarsest solve gathers rows collectively: keep the check-every-iteration loop

    def forward_solve(self, lvl: int) -> None:
        """Local coarse-grid solves on the coarsest level (at_mgrit.py:37-87); a one-level hierarchy solves nothing there."""
        t0 = time.time()
        if self.lvl_max != 1:
            self.backend.at_forward_solve(lvl, self.k)
        logging.debug(f"Forward solve on {self.comm_time_rank} took {time.time() - t0} s")

    def setup_points_and_comm_info(self, lvl: int) -> None:
        """Mgrit's layout plus, on the coarsest level, which rank holds which coarse point and this rank's local coarse
        grid (at_mgrit.py:186-214)."""
        super().setup_points_and_comm_info(lvl=lvl)
        if lvl != self.lvl_max - 1:
            return
        ends = np.cumsum(self.split_into(number_points=len(self.global_t[0]), number_processes=self.comm_time_size)) - 1
        split = self.global_t[0][ends]
        self.comm_coarsest_level = np.array([np.min(np.where(item <= split)) for item in self.global_t[-1]])
        _, self.c_points_per_proc = np.unique(self.comm_coarsest_level, return_counts=True)
        holders = self.comm_coarsest_level[1:] if self.c_points_per_proc[0] == 2 else self.comm_coarsest_level
        if self.comm_time_rank in holders and self.cpts[lvl].size > 0:
            first = self.cpts[lvl][0]
            last = self.cpts[lvl][1] if (self.comm_time_rank == 0 and self.c_points_per_proc[0] != 1 and
                                         self.cpts[lvl].size > 1) else first
            self.local_coarse_grid = np.array(self.global_t[lvl][max(0, first - self.k + 1):last + 1])

    def ouput_run_information(self) -> None:
        rows = [('time interval', '[' + str(self.problem[0].t[0]) + ', ' + str(self.problem[0].t[-1]) + ']'),
                ('number of time points ', str(len(self.problem[0].t))),
                ('max dt ', str(np.max(self.problem[0].t[1:] - self.problem[0].t[:-1]))),
                ('number of levels', str(self.lvl_max)), ('coarsening factors', str(self.m[:-1])),
                ('relaxation weight', str(self.weight_c)), ('cf_iter', str(self.cf_iter)),
                ('nested iteration', str(self.nes_it)), ('cycle type', str(self.cycle_type)),
                ('stopping tolerance', str(self.tol)), ('time communicator size', str(self.comm_time_size)),
                ('space communicator size', str(self.comm_space_size)), ('distance', str(self.k))]
        self.log_info('\n'.join(['Run parameter overview'] + ['  ' + '{0: <25}'.format(k) + ' : ' + v for k, v in rows]))
