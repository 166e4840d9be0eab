"""MGRIT solver driver for MI355X: same constructor, attributes and ``solve()`` surface as the reference's
``pymgrit.core.mgrit.Mgrit`` (reference src/pymgrit/core/mgrit.py:33-37,590-646), with the relaxation / residual /
grid-transfer sweeps delegated to a sweep backend:

  * ``HipBackend``    -- every level's states in one HBM slab, sweeps as gfx950 kernels (libmgrit_hip.so);
                         chosen when ALL level Applications provide ``device_stepper()``. No CPU fallback.
  * ``PluginBackend`` -- the user's own Python ``Application.step`` per time point; chosen when NO level provides
                         a device description (user plugins, Dahlquist).

What stays on the host is control flow only: the cycle recursion (mgrit.py:261-290), the exchange schedule at rank
boundaries (mgrit.py:304-331,346-352,397-403,467-484,502-517 -- mpi4py pickled isend/recv replaced by
torch.distributed point-to-point, RCCL over xGMI for device rows) and the stopping test.

Local convergence criteria (conv_crit 2/3) run on one or several ranks (the farewell messages of a rank that leaves the
solve loop early: Mgrit._leave_local).
"""
import logging
import os
import sys
import time
from typing import List, Tuple

import numpy as np

from pymgrit_amd.core.options import options
from pymgrit_amd.core.application import Application
from pymgrit_amd.core.comm import resolve_comm
from pymgrit_amd.core.grid_transfer import GridTransfer
from pymgrit_amd.core.grid_transfer_copy import GridTransferCopy
from pymgrit_amd.core.pipelined import PipelinedLoop
from pymgrit_amd.core.rank_schedules import RankSchedules
from pymgrit_amd.core.layout import IndexArray, as_index_array, compute_layout, consecutive_runs, member_mask, \
    split_into as _split_into, split_points as _split_points


def time_norm(values: np.ndarray, ord) -> float:
    """np.linalg.norm(values, ord) for ord in {1, None, inf} (mgrit.py:182,430) without the BLAS ddot of the 2-norm:
    on a many-core host the threaded BLAS wake-up (tens of ms after an idle stretch) would sit on the critical path of
    every MGRIT iteration. The 2-norm is sqrt(sum(v*v)) with numpy's pairwise summation (<= 1 ulp from ddot)."""
    if values.size == 0:
        return 0.0
    if ord is None:
        return float(np.sqrt(np.sum(values * values)))
    if ord == 1:
        return float(np.sum(np.abs(values)))
    return float(np.max(np.abs(values)))


class IndexList(list):
    """A list of runs / points / pairs that can carry backend handles as attributes (plain lists cannot)."""


class Mgrit(RankSchedules, PipelinedLoop):
    """Multigrid-reduction-in-time (FAS) solver. The solved space-time stencil is [-Phi I] on every level."""

    def __init__(self, problem: List[Application], transfer: List[GridTransfer] = None, weight_c: float = 1.0,
                 max_iter: int = 100, tol: float = 1e-7, nested_iteration: bool = True, cf_iter: int = 1,
                 cycle_type: str = 'V', comm_time=None, comm_space=None, logging_lvl: int = logging.INFO,
                 output_fcn=None, output_lvl=1, t_norm=2, random_init_guess: bool = False, conv_crit: int = 0,
                 pipeline_depth: int = None, plan_blocks: int = None) -> None:
        logging.basicConfig(format='%(levelname)s - %(asctime)s - %(message)s', datefmt='%d-%m-%y %H:%M:%S',
                            level=logging_lvl, stream=sys.stdout)
        if transfer is None:
            transfer = [GridTransferCopy() for _ in range(len(problem) - 1)]
        cf_iter = self._validate(problem, transfer, cycle_type, output_lvl, t_norm, conv_crit, cf_iter)

        self.comm_time = resolve_comm(comm_time)
        self.comm_space = comm_space
        self.comm_time_rank = self.comm_time.Get_rank()
        self.comm_time_size = self.comm_time.Get_size()
        if self.comm_time_size > len(problem[0].t):
            raise Exception('More processors than time points. Not useful and not implemented yet')
        if conv_crit in (2, 3) and self.comm_time_size > 1 and any(not isinstance(t, GridTransferCopy) for t in transfer):
            # the farewell message of op 4 (reference mgrit.py:663-665) carries a row of the FINER level into a buffer of the
            # coarser one: with a transfer that changes the number of values that is a size-mismatched point-to-point message
            raise Exception('Local convergence criteria on several ranks need the identity transfer (GridTransferCopy) on '
                            'every level')
        self.spatial_parallel = comm_space is not None
        self.comm_space_rank = comm_space.Get_rank() if self.spatial_parallel else -99
        self.comm_space_size = comm_space.Get_size() if self.spatial_parallel else 1

        getattr(self.comm_time, 'prepare', lambda: None)()   # optional hook of the communicator
        self.comm_time.barrier()
        setup_start = time.time()
        self.log_info("Start setup")

        self.problem = problem
        self.transfer_objects = transfer
        self.weight_c = weight_c
        self.lvl_max = len(problem)
        self.step, self.u, self.v, self.g, self.t, self.m = [], [], [], [], [], []
        self.restriction, self.interpolation = [], []
        self.tol = tol
        self.conv = np.zeros(max_iter + 1)
        self.cf_iter = cf_iter
        self.cycle_type = cycle_type
        self.random_init_guess = random_init_guess
        self.iter_max = max_iter
        self.solve_iter = 0
        self.nes_it = nested_iteration
        self.runtime_solve = self.runtime_setup = 0
        self.int_start = self.int_stop = 0
        self.cpts, self.index_local, self.index_local_c, self.index_local_f = [], [], [], []
        self.comm_front, self.comm_back = [], []
        self.first_is_f_point, self.first_is_c_point, self.last_is_f_point, self.last_is_c_point = [], [], [], []
        self.send_to, self.get_from, self.global_t = [], [], []
        self.t_norm = 1 if t_norm == 1 else None if t_norm == 2 else np.inf
        self.conv_crit = conv_crit
        self.global_conv_crit = conv_crit in (0, 1)
        self.finished = [False, None]
        self.pre_finished = [True, 0] if self.comm_time_rank == 0 else [False, None]   # mgrit.py:224-228
        self._dry = None            # 'send' / 'recv': farewell traffic of the local criteria (_drain_out / _drain_in)
        self._gone = {}             # rank -> 'draining' / 'done': lower ranks that have left the solve loop (local criteria)
        self._gone_count = 0        # ranks 0 .. _gone_count-1 have left
        self._drain_seen = set()
        self._announced_at = None   # criterion call in which this rank told its successor that it has finished
        self._announce_seen = set()
        self._real_backend = None
        self.save_values_last_iter = None
        self._pipeline_request = pipeline_depth
        self._pl = None
        self._plan_request, self._plans, self._plan_recording = plan_blocks, {}, False
        self.output_lvl = output_lvl
        self.output_fcn = output_fcn if (output_fcn is not None and callable(output_fcn)) else None
        self._ghost, self._is_c_local = [], []

        self._transfer_for_selection = transfer
        self.backend = self._select_backend(problem)

        for lvl in range(self.lvl_max):
            self.t.append(np.copy(problem[lvl].t))
            if lvl != self.lvl_max - 1:
                self.restriction.append(transfer[lvl].restriction)
                self.interpolation.append(transfer[lvl].interpolation)
                cp = np.where(member_mask(problem[lvl].t, problem[lvl + 1].t))[0]
                gaps = np.diff(cp)
                self.m.append(int(gaps[0]))
                if not np.all(np.isclose(gaps, gaps[0])) and self.comm_time_rank == 0:
                    logging.warning('Non-uniform coarsening between level ' + str(lvl) + ' and ' + str(lvl + 1) +
                                    '. Poorly tested.')
            else:
                self.m.append(1)
            self.setup_points_and_comm_info(lvl=lvl)
            self.step.append(problem[lvl].step)
            self.create_u_v_g(lvl=lvl)
        self.backend.finalize()
        self._aligned = self._detect_aligned()
        # logging.DEBUG: the reference reports the time of every sweep (mgrit.py:333,370,486,549); on the device path the lines
        # also carry the DEVICE time of the sweep's kernels (HIP events around every entry point, mgrit_hip_set_timing)
        self._sweep_timing = logging_lvl <= logging.DEBUG and hasattr(self.backend, "timing_drain")
        if self._sweep_timing:
            self.backend.set_timing(True)

        if nested_iteration:
            self.nested_iteration()
        if self.conv_crit in (1, 3):
            self.backend.save_last()

        if self.iter_max == 0:
            self.comm_time.barrier()
        self.runtime_setup = time.time() - setup_start
        if self.output_fcn is not None and self.output_lvl == 2:
            self.output_fcn(self)
        self.log_info(f"Setup took {self.runtime_setup} s")

    # ------------------------------------------------------------------------------------------------
    # construction helpers
    # ------------------------------------------------------------------------------------------------
    @staticmethod
    def _validate(problem, transfer, cycle_type, output_lvl, t_norm, conv_crit, cf_iter):
        """Constructor checks and messages of the reference (mgrit.py:78-120)."""
        if len(problem) != (len(transfer) + 1):
            raise Exception('There should be exactly one transfer operator for each level except the coarsest grid')
        for i in range(len(problem) - 1):
            if len(problem[i].t) < len(problem[i + 1].t):
                raise Exception('The time grid on level ' + str(i + 1) + ' contains more time points than level ' + str(i))
        if cycle_type not in ('V', 'F'):
            raise Exception("Cycle-type " + str(cycle_type) + " is not implemented. Choose 'V' or 'F'")
        if output_lvl not in [0, 1, 2]:
            raise Exception("Unknown output level. Choose 0, 1 or 2.")
        for lvl in range(1, len(problem)):
            t_l = np.asarray(problem[lvl].t)
            if t_l.size > 1 and not bool((t_l[1:] > t_l[:-1]).all()):
                t_l = np.unique(t_l)        # (an ascending grid is its own set of unique values)
            if np.count_nonzero(member_mask(t_l, problem[lvl - 1].t)) != len(problem[lvl].t):
                raise Exception('Some points from level ' + str(lvl - 1) + ' are not points of level ' + str(lvl))
        if t_norm not in [1, 2, 3]:
            raise Exception('Unknown norm. Please choose 1 (one norm), 2 (two-norm) or 3 (inf-norm)')
        if conv_crit not in [0, 1, 2, 3]:
            raise Exception('Unknown convergence criterion. Please choose: 0 (global space-time residual), '
                            '1 (global jump)2 (local space-time residual)3 (local jump)')
        if isinstance(cf_iter, int):
            return [cf_iter for _ in range(len(problem))]
        if isinstance(cf_iter, list):
            if len(cf_iter) < len(problem) - 1:
                raise Exception('Too few cf_iter. Specify a list of values for all but the coarsest level or an integer '
                                '(used for all levels).')
            return cf_iter
        raise Exception('Incorrect datatype cf_iter. Specify a list of values for all but the coarsest level or an '
                        'integer ( used for all levels).')

    @staticmethod
    def _library_method(obj, name):
        """True when obj.<name> is the library's own implementation: the class that provides it lives in pymgrit_amd and
        the instance does not shadow it. A user subclass that overrides ``step`` (or a transfer's ``restriction`` /
        ``interpolation``) wants ITS code to run -- the device kernels would silently ignore it."""
        if name in getattr(obj, "__dict__", {}):
            return False
        for klass in type(obj).__mro__:
            if name in klass.__dict__:
                return klass.__module__.split(".")[0] == "pymgrit_amd"
        return False

    def _select_backend(self, problem):
        """Backend by application TYPE (never by hardware availability): the HIP engine for hierarchies whose levels all
        describe their time stepper declaratively (``device_stepper()``) AND whose ``step`` / transfer methods are the
        library's own; everything else -- user plugins, Dahlquist, user subclasses that override ``step`` or a transfer --
        runs through the plugin path, i.e. through the user's Python code, like in the reference."""
        has_desc = [hasattr(p, "device_stepper") and p.device_stepper() is not None for p in problem]
        custom = [type(p).__name__ for p, d in zip(problem, has_desc) if d and not self._library_method(p, "step")]
        if all(has_desc) and custom:
            logging.warning('pymgrit_amd: %s override step(): the hierarchy runs through these Python methods (plugin path), '
                            'not through the MI355X kernels', sorted(set(custom)))
            has_desc = [False] * len(problem)
        # (a user's GridTransfer, or a library transfer with overridden restriction / interpolation, does NOT leave the device
        # path: the 1-D engine applies such a transfer through its Python methods between the kernels, backend_hip.fas_rhs)
        if all(has_desc):
            from pymgrit_amd.core.backend_hip import HipBackend
            return HipBackend(self)
        if any(has_desc) and not custom:
            # the reference takes ANY Application per level (core/mgrit.py:79-99). A hierarchy that mixes library applications with
            # a user's Python application runs where that application's code runs: every level through its step() on the plugin path
            # (the library's applications carry host steppers for exactly this), not through the kernels
            host = sorted({type(p).__name__ for p, d in zip(problem, has_desc) if not d})
            logging.warning('pymgrit_amd: %s have no device description: the whole hierarchy runs through the applications\' step() '
                            'methods (plugin path), not through the MI355X kernels', host)
        from pymgrit_amd.core.backend_plugin import PluginBackend
        return PluginBackend(self)

    def _log_sweep(self, what: str, t0: float) -> None:
        """the reference's per-sweep debug line (same wording), plus the device time of the sweep's kernels"""
        if not logging.getLogger().isEnabledFor(logging.DEBUG):
            return
        line = f"{what} on {self.comm_time_rank} took {time.time() - t0} s"
        if getattr(self, "_sweep_timing", False) and not self._plan_recording:
            recs = self.backend.timing_drain()
            if recs:
                line += " | device: " + ", ".join(f"{k} L{lv} {ms:.3f} ms" for k, lv, ms in recs)
        logging.debug(line)

    def log_info(self, message: str) -> None:
        """Only the last time rank (and space rank 0) logs (mgrit.py:247-259)."""
        if self.comm_time_rank == self.comm_time_size - 1 and (not self.spatial_parallel or self.comm_space_rank == 0):
            logging.info(message)

    def split_into(self, number_points: int, number_processes: int) -> np.ndarray:
        return _split_into(number_points, number_processes)

    def split_points(self, length: int, size: int, rank: int) -> Tuple[int, int]:
        return _split_points(length, size, rank)

    def setup_points_and_comm_info(self, lvl: int) -> None:
        """Local index sets and exchange flags of level ``lvl`` for (comm_time_rank, comm_time_size); appends to the
        per-level attribute lists exactly like the reference (mgrit.py:742-827) so the rank-overwrite trick of its
        tests (tests/core/test_mgrit.py:109-129) works on this class too."""
        self.global_t.append(np.copy(self.problem[lvl].t))
        lay = compute_layout([p.t for p in self.problem], lvl, self.comm_time_rank, self.comm_time_size)
        if lvl == 0:
            self.int_start, self.int_stop = lay.int_start, lay.int_stop
        self.t[lvl] = lay.t_local
        self.cpts.append(lay.cpts)
        self.comm_front.append(lay.comm_front)
        self.comm_back.append(lay.comm_back)
        self.index_local.append(lay.index_local)
        self.index_local_c.append(lay.index_local_c)
        self.index_local_f.append(lay.index_local_f)
        self.first_is_c_point.append(lay.first_is_c_point)
        self.first_is_f_point.append(lay.first_is_f_point)
        self.last_is_c_point.append(lay.last_is_c_point)
        self.last_is_f_point.append(lay.last_is_f_point)
        self.send_to.append(lay.send_to)
        self.get_from.append(lay.get_from)
        self._ghost.append(lay.ghost)
        self._is_c_local.append(lay.is_c_local)

    def create_u_v_g(self, lvl: int) -> None:
        self.backend.create_u_v_g(lvl)

    # ------------------------------------------------------------------------------------------------
    # run / pair lists derived from the index sets
    # ------------------------------------------------------------------------------------------------
    def _cached(self, key, build):
        """index-derived lists are built once per (kind, level) and reused by every sweep (they also carry the
        backend's device-side handle, see IndexList)"""
        store = self.__dict__.setdefault('_index_lists', {})
        if key not in store:
            got = build()
            store[key] = got if isinstance(got, IndexArray) else IndexList(got)
        return store[key]

    def _f_runs(self, lvl):
        return self._cached(('f', lvl), lambda: consecutive_runs(
            np.sort(np.asarray(self.index_local_f[lvl], dtype=np.int64))))

    def _c_points(self, lvl):
        """local C-point slots that are relaxed: all except global index 0 (mgrit.py:357,408,525)."""
        def build():
            pts = np.asarray(self.index_local_c[lvl], dtype=np.int64)
            if self.comm_time_rank == 0 and pts.size and pts[0] == 0:
                pts = pts[1:]
            return IndexArray(pts)
        return self._cached(('c', lvl), build)

    def _c_runs(self, lvl):
        return self._cached(('crun', lvl), lambda: consecutive_runs(self._c_points(lvl)))

    def _pairs(self, lvl, skip_first):
        """(fine slot of the i-th local C-point, coarse slot index_local[lvl+1][i]) (mgrit.py:498-500,528,722-726)."""
        def build():
            fine = np.asarray(self.index_local_c[lvl], dtype=np.int64)
            coarse = np.asarray(self.index_local[lvl + 1], dtype=np.int64)[:fine.size]
            out = np.stack((fine[:coarse.size], coarse), axis=1)
            if skip_first and self.comm_time_rank == 0:
                out = out[1:]
            return IndexArray(out, width=2)
        return self._cached(('pair', lvl, bool(skip_first)), build)

    def _exchange(self, lvl, send_idx=None, recv_idx=None, dest=None, src=None, op=None):
        """One exchange point of operation `op` (the reference's op ids 0-5, 7; mgrit.py:693-713). With a local stopping
        criterion ranks leave the solve loop one after the other (see _drain_out): a receive from a predecessor that has
        left is served once per (level, op) from its farewell messages and skipped afterwards."""
        if self._dry == 'send':        # farewell of a finished rank: first occurrence of every send, frozen values, no receives
            recv_idx = None
            if send_idx is not None:
                if (lvl, op) in self._drain_seen:
                    return
                self._drain_seen.add((lvl, op))
                if op == 4:            # mgrit.py:663-665: the last C-point of the FINER level (identity transfer)
                    send = (self._real_backend.payload(lvl - 1, int(self.index_local_c[lvl - 1][-1])), dest)
                    self.comm_time.exchange(send=send)
                    return
        if self._announced_at is not None and send_idx is not None and self._dry is None:
            # finished by the point-wise test but still iterating (the norm of the local values is not below tol yet,
            # mgrit.py:627-635): the successor takes the first message of every (level, op) of the NEXT iteration and
            # nothing after that (sender_finished, mgrit.py:693-701)
            if self.solve_iter != self._announced_at + 1 or (lvl, op) in self._announce_seen:
                send_idx = None
            else:
                self._announce_seen.add((lvl, op))
        if self._dry != 'send' and recv_idx is not None and not self.global_conv_crit:
            state = self._gone.get(src)     # the sender of this level may be any lower rank (ranks without points are skipped)
            if state == 'done' or (self._dry == 'recv' and state != 'draining'):
                recv_idx = None
            elif state == 'draining':
                if (lvl, op) in self._drain_seen:
                    recv_idx = None
                else:
                    self._drain_seen.add((lvl, op))
        if self._dry == 'recv':
            send_idx = None
        backend = self._real_backend if self._dry else self.backend
        if self._dry is None and self.global_conv_crit and getattr(backend, "device_links", False):
            if send_idx is not None or recv_idx is not None:    # rows as stream operations of the engine (mgrit_hip_exchange)
                backend.exchange(lvl, op, send_idx=send_idx, dest=dest, recv_idx=recv_idx, src=src)
            return
        kw = {'op': op} if op == 5 else {}     # op 5 (forward-solve hand-over): a backend may send more than the point itself
        send = (backend.payload(lvl, send_idx, **kw), dest) if send_idx is not None else None
        recv = (backend.recv_buffer(lvl, recv_idx, **kw), src) if recv_idx is not None else None
        if send is None and recv is None:
            return
        got = self.comm_time.exchange(send=send, recv=recv)
        if recv is not None:
            backend.commit(lvl, recv_idx, got, **kw)

    def _last_slot(self, lvl):
        return len(self.t[lvl]) - 1

    # ------------------------------------------------------------------------------------------------
    # the MGRIT cycle (mgrit.py:261-290)
    # ------------------------------------------------------------------------------------------------
    def plan_blocks(self, probe_usable: bool = False) -> int:
        """Blocks of time points of a planned cycle (core/cycle_plan.py); 1 = the cycle runs in program order. The plan
        reorders the launches of ONE rank's cycle, so it needs a cycle without exchange points (one rank) whose sweeps are
        the library's own (a subclass that overrides a sweep keeps the program order)."""
        if self._dry is not None or not self._one_rank_like():
            return 0 if probe_usable else 1
        if self._plan_request is not None:
            want = int(self._plan_request)
        elif options.plan_blocks is not None:
            want = int(options.plan_blocks)
        else:
            want = int(getattr(self.backend, "plan_blocks", lambda: 1)() or 1)
        own = all(getattr(type(self), name) is getattr(Mgrit, name) for name in
                  ("iteration", "f_relax", "c_relax", "fas_residual", "error_correction", "forward_solve", "_exchange",
                   "_ec_f_relax", "_fas_residual_fused", "_relax_f"))
        usable = (self._one_rank_like() and self.lvl_max > 1 and own and self._dry is None and
                  not getattr(self, "_sweep_timing", False) and     # per-sweep debug timing reports the sweeps in program order
                  getattr(self.backend, "plan_allowed", lambda: True)())
        if probe_usable:
            return int(usable)
        return max(want, 1) if usable else 1

    def _planned(self, cycle_type, iteration, first_f):
        """the cycle plan for this cycle shape (recorded on first use), or None"""
        blocks = self.plan_blocks()
        if blocks <= 1:
            # one block = program order; the device backend still replays it as one graph launch (small hierarchies)
            single = (self._plan_request is None and options.plan_blocks is None and self._one_rank_like()
                      and self.plan_blocks(probe_usable=True) and getattr(self.backend, "plan_single_block", lambda: False)())
            if not single:
                return None
            blocks = 1
        from pymgrit_amd.core.cycle_plan import PlanUnsupported, record_cycle
        key = (cycle_type, iteration == 0, bool(first_f), blocks, tuple(self.cf_iter), float(self.weight_c),
               bool(getattr(self.backend, "_cycle_pre", False)),   # (the down pass's launch argument, baked into a captured graph)
               bool(getattr(self.backend, "_mirror_on", False)),   # (likewise: the C-point mirror of the way up)
               getattr(self.backend, "write_generation", lambda: None)(),
               frozenset(self.__dict__.get('_head_done', ())))          # which injections of the first time point the cycle still holds
        if key not in self._plans:
            self._plan_recording = True
            held = set(self.__dict__.get('_head_done', ()))
            try:
                self._plans[key] = record_cycle(self, self.backend, blocks, lambda: self.iteration(
                    lvl=0, cycle_type=cycle_type, iteration=iteration, first_f=first_f))
            except PlanUnsupported:
                self._plans[key] = None
                self._head_done = held      # nothing of the abandoned recording has run
            finally:
                self._plan_recording = False
        return self._plans[key]

    def iteration(self, lvl: int, cycle_type: str, iteration: int, first_f: bool) -> None:
        if lvl == 0 and not self._plan_recording:
            if first_f and iteration == 0:
                # the cycle opens with a plain F-relaxation of level 0 (mgrit.py:273-275): whatever the cycle before left for the
                # whole-level down pass alone (pre-relaxed C-points, backend_hip._f_stale == 2) is put back in place first, so
                # that no cycle ever starts in that state with a level-0 sweep in front of the pass
                getattr(self.backend, "materialise", lambda: None)()
            getattr(self.backend, "begin_cycle", lambda: None)()
            plan = self._planned(cycle_type, iteration, first_f)
            if plan is not None:
                plan.run(self.backend)
                return
        if lvl == self.lvl_max - 1:
            self.forward_solve(lvl=lvl)
            return
        fresh, self._fresh_level = getattr(self, "_fresh_level", None) == lvl, None
        coarse = self._coarse_down(lvl) if (fresh and first_f and lvl > 0) else None
        crank = self._coarse_down_rank(lvl) if (coarse is None and fresh and first_f and lvl > 0 and self.comm_time_size > 1) else None
        if crank is not None:
            # several ranks: the two coarse-level passes on the rank's complete intervals; the first local C-point, the partial
            # intervals at the two ends and every exchange point as in the sweep-by-sweep form
            fc_runs, c0_run, edge_runs = crank
            self.f_relax(lvl=lvl, runs=edge_runs)                        # first F-relaxation: ops 0 / 1, partial intervals
            self._exchange(lvl, send_idx=self._last_slot(lvl) if self.last_is_f_point[lvl] else None,
                           recv_idx=0 if self.first_is_c_point[lvl] else None, dest=self.send_to[lvl], src=self.get_from[lvl], op=2)
            self.backend.relax(lvl, c0_run, 'C')
            self.backend.relax(lvl, fc_runs, 'FC')                       # ... and C-relaxation, complete intervals
            self.f_relax(lvl=lvl, runs=edge_runs)                        # second F-relaxation: ops 0 / 1, partial intervals
            self._fas_residual_fused(lvl, with_f_relax=True)             # ... folded into the FAS pass for the complete intervals
            self._fresh_level = lvl + 1
            self.iteration(lvl=lvl + 1, cycle_type=cycle_type, iteration=iteration, first_f=True)
            self._up(lvl, None)
            if cycle_type == 'F':
                self.iteration(lvl=lvl, cycle_type='V', iteration=iteration, first_f=False)
            return
        if coarse is not None:
            # a level the finer level's FAS sweep has just filled (u == v): F-relaxation + C-relaxation in one pass, then the
            # F-relaxation folded into the FAS sweep (the F-points of the way down are stored by neither)
            fc_runs, triples, head, skip_u = coarse
            ranks = self.comm_time_size > 1     # aligned ranks: the exchange points of the sweeps this pass stands for
            # (op 0 of the first F-relaxation, mgrit.py:271, would carry the last point of the rank before as the finer level's FAS
            # sweep has just injected it -- the very row op 4 of that sweep has put into the ghost slot already (Mgrit._x4; the
            # level is `fresh`): not sent a second time)
            if self.cf_iter[lvl] == 1:
                self.backend.relax(lvl, fc_runs, 'FC')
                if ranks:
                    self._x0(lvl)               # f_relax (mgrit.py:275), behind the C-relaxation
            self._head(lvl, head, 'u')
            self.backend.fas_fused(lvl, triples, with_f_relax=True, skip_coarse_u=skip_u)
            self._head(lvl, head, 'v')
            if ranks:
                if skip_u and self.send_to[lvl + 1] >= 0:   # the one row of u^{l+1} that op 4 sends
                    self.backend.restrict_u(lvl, self._cached(('pair_last', lvl), lambda: [self._xpairs(lvl)[-1]]))
                self._x4(lvl)                   # fas_residual (mgrit.py:511-520)
            self._fresh_level = lvl + 1
            self.iteration(lvl=lvl + 1, cycle_type=cycle_type, iteration=iteration, first_f=True)
            self._up(lvl, None)
            if cycle_type == 'F':
                self.iteration(lvl=lvl, cycle_type='V', iteration=iteration, first_f=False)
            return
        if first_f and (lvl > 0 or iteration == 0):
            # (tried in round 3: this F-relaxation as part of the general way-down pass on a level the finer level's FAS sweep has
            # just filled -- per interval the F-points from v, not stored. Bit-identical and slower: config 5's level-1 pass 0.44 ->
            # 0.76 ms for the 0.17 ms launch it replaces; the extra Phi per interval costs more than the rows it saves.)
            self.f_relax(lvl=lvl)
        fused = self._level_intervals(lvl)     # whole-level sweeps in one pass (device backend, one rank), or None
        shard = self._rank_intervals(lvl) if (fused is None and self.comm_time_size > 1 and self.cf_iter[lvl] >= 1) else None
        if shard is not None:
            # several ranks: the same down pass on the rank's COMPLETE intervals (both C-points local); the first local C-point,
            # the partial intervals at the two ends and every exchange point as in the sweep-by-sweep form
            intervals, c0_run, edge_runs = shard
            for _ in range(self.cf_iter[lvl] - 1):
                self.backend.f_relax_follows = True
                try:
                    self.c_relax(lvl=lvl)
                finally:
                    self.backend.f_relax_follows = False
                self.f_relax(lvl=lvl)
            self._exchange(lvl, send_idx=self._last_slot(lvl) if self.last_is_f_point[lvl] else None,
                           recv_idx=0 if self.first_is_c_point[lvl] else None, dest=self.send_to[lvl], src=self.get_from[lvl], op=2)
            self.backend.relax(lvl, c0_run, 'C')                    # the first local C-point (its interval began on the rank before)
            self.backend.cf_fas(lvl, intervals)                     # C-relaxation, F-relaxation, FAS residual of the complete intervals
            self.f_relax(lvl=lvl, runs=edge_runs)                   # ops 0 / 1 and the partial intervals
            self._fas_residual_fused(lvl, skip_triples=True)        # ops 3 / 4 and the first local C-point
            self._fresh_level = lvl + 1
            self.iteration(lvl=lvl + 1, cycle_type=cycle_type, iteration=iteration, first_f=True)
            self._up(lvl, None)
            return
        down = fused is not None and self.cf_iter[lvl] >= 1
        gen = self._gen_intervals(lvl) if (fused is None and self.cf_iter[lvl] >= 1) else None
        for _ in range(self.cf_iter[lvl] - (1 if (down or gen is not None) else 0)):
            self.backend.f_relax_follows = True     # (the F-relaxation below rewrites every F-point: nothing to put in place first)
            try:
                self.c_relax(lvl=lvl)
            finally:
                self.backend.f_relax_follows = False
            self.f_relax(lvl=lvl)
        if down:      # the last C-relaxation + F-relaxation + the FAS residual: one pass
            head = self._cached(('pair_head_x', lvl), lambda: self._pairs(lvl, skip_first=False)[:1] if self.comm_time_rank == 0 else [])
            if self.comm_time_size > 1:
                # aligned ranks: op 0 of the F-relaxation inside the pass (mgrit.py:306-310) wants the RELAXED last C-point, which
                # the pass itself computes. With pre-relaxed C-points (backend_hip._f_stale == 2) the cycle before has left exactly
                # that value in the row of the last F-point; otherwise it is computed up front by a C-relaxation of that one
                # point (the pass recomputes the same bits from the same F-point)
                last = self._last_slot(lvl)
                if getattr(self.backend, "_cycle_pre", False) and lvl == 0:
                    self._x0(lvl, send_row=last - 1)
                else:
                    if self.last_is_c_point[lvl]:
                        self.backend.relax(lvl, self._cached(('c_last', lvl), lambda: [(last, 1)]), 'C')
                    self._x0(lvl)
            self._head(lvl, head, 'u')
            self.backend.cf_fas(lvl, fused)
            self._head(lvl, head, 'v')
            if self.comm_time_size > 1:
                self._x4(lvl)
        elif gen is not None:      # any 1-D stepper pair, any of the library's transfers: the same sweeps as three launches
            head = self._cached(('pair_head_x', lvl), lambda: self._pairs(lvl, skip_first=False)[:1] if self.comm_time_rank == 0 else [])
            if self.comm_time_size > 1:
                # aligned ranks: op 0 of the F-relaxation inside the pass (mgrit.py:306-310) carries the RELAXED last C-point: computed
                # up front by a C-relaxation of that one point (the pass recomputes the same bits from the same F-point)
                last = self._last_slot(lvl)
                if self.last_is_c_point[lvl]:
                    self.backend.relax(lvl, self._cached(('c_last', lvl), lambda: [(last, 1)]), 'C')
                self._x0(lvl)
            self._head(lvl, head, 'u')
            self._head(lvl, head, 'v')      # (the coarse half of the first interval starts from v^{l+1}_0)
            if self.comm_time_size > 1:     # op 4 (mgrit.py:511-520) between the restriction and the coarse half, which reads v of the ghost
                self.backend.gen_down(lvl, gen, 1)
                self._x4(lvl)
                self.backend.gen_down(lvl, gen, 2)
            else:
                self.backend.gen_down(lvl, gen)
        else:
            self.fas_residual(lvl=lvl)
        self._fresh_level = lvl + 1      # the next level starts from what the FAS sweep has just written (u == v there)
        self.iteration(lvl=lvl + 1, cycle_type=cycle_type, iteration=iteration, first_f=True)
        self._up(lvl, fused, gen)
        if lvl != 0 and cycle_type == 'F':
            self.iteration(lvl=lvl, cycle_type='V', iteration=iteration, first_f=False)

    def f_relax(self, lvl: int, ec: bool = False, runs=None) -> None:
        """F-relaxation (mgrit.py:292-333): every F-interval is propagated from its preceding point. Exchange:
        op 0 = last local C-point to the next owner's ghost; op 1 = hand-off inside an F-interval that straddles a
        rank boundary (comm_front / comm_back).
        ec=True (internal, Mgrit._ec_f_relax): the launches also apply the error correction to the C-point in front of
        each interval; the sweep order and the exchange points are those of the plain relaxation."""
        t0 = time.time()
        self._exchange(lvl, send_idx=self._last_slot(lvl) if self.last_is_c_point[lvl] else None,
                       recv_idx=0 if self.first_is_f_point[lvl] else None, dest=self.send_to[lvl], src=self.get_from[lvl], op=0)
        runs = self._f_runs(lvl) if runs is None else runs     # (runs: the boundary runs of a rank whose complete intervals
        if runs:                                                # a whole-level pass has already relaxed, _rank_intervals)
            front, back = self.comm_front[lvl], self.comm_back[lvl]
            if front and back and len(runs) == 1:
                self._exchange(lvl, recv_idx=0, src=self.get_from[lvl], op=1)
                self._relax_f(lvl, 'f_all', runs, ec)
                self._exchange(lvl, send_idx=self._last_slot(lvl), dest=self.send_to[lvl], op=1)
            elif not front and not back:
                self._relax_f(lvl, 'f_all', runs, ec)
            else:
                lo, hi = (1 if front else 0), (len(runs) - 1 if back else len(runs))
                tag = '' if runs is self._f_runs(lvl) else '_b'
                if back:  # the interval feeding the next rank goes first
                    self._relax_f(lvl, 'f_last' + tag, self._cached(('f_last' + tag, lvl), lambda: runs[-1:]), ec)
                    self._exchange(lvl, send_idx=self._last_slot(lvl), dest=self.send_to[lvl], op=1)
                self._relax_f(lvl, 'f_mid' + tag, self._cached(('f_mid' + tag, lvl), lambda: runs[lo:hi]), ec)
                if front:
                    self._exchange(lvl, recv_idx=0, src=self.get_from[lvl], op=1)
                    self._relax_f(lvl, 'f_first' + tag, self._cached(('f_first' + tag, lvl), lambda: runs[:1]), ec)
        self._log_sweep("F-relax", t0)

    def _relax_f(self, lvl, tag, runs, ec):
        if not ec:
            self.backend.relax(lvl, runs, 'F')
            return
        def build():   # (start, length, coarse slot of the corrected C-point in front of the run, or -1)
            P, R = as_index_array(self._pairs(lvl, skip_first=True), 2), as_index_array(runs, 2)
            co = np.full(R.shape[0], -1, dtype=np.int64)
            if P.shape[0]:      # (the fine slots of the pairs ascend)
                at = np.minimum(np.searchsorted(P[:, 0], R[:, 0] - 1), P.shape[0] - 1)
                hit = P[at, 0] == R[:, 0] - 1
                co[hit] = P[at[hit], 1]
            return IndexArray(np.column_stack((R, co)), width=3)
        self.backend.ec_relax(lvl, self._cached(('ec_' + tag, lvl), build))

    def _ec_f_relax(self, lvl: int) -> None:
        """error_correction(lvl) then f_relax(lvl) (mgrit.py:283-284) with the correction of every C-point that is followed
        by a local F-interval folded into that interval's launch; the remaining C-points (the last local point, C-points
        followed by another C-point) are corrected up front -- in particular before op 0 sends the last local C-point."""
        def leftover():
            P, R = as_index_array(self._pairs(lvl, skip_first=True), 2), as_index_array(self._f_runs(lvl), 2)
            return IndexArray(P[~np.isin(P[:, 0], R[:, 0] - 1)], width=2)
        self.backend.error_correction(lvl, self._cached(('ec_left', lvl), leftover))
        self.f_relax(lvl, ec=True)

    def c_relax(self, lvl: int) -> None:
        """C-relaxation (mgrit.py:335-370); op 2 = last local F-point to the next owner's ghost."""
        t0 = time.time()
        self._exchange(lvl, send_idx=self._last_slot(lvl) if self.last_is_f_point[lvl] else None,
                       recv_idx=0 if self.first_is_c_point[lvl] else None, dest=self.send_to[lvl], src=self.get_from[lvl], op=2)
        self.backend.relax(lvl, self._c_runs(lvl), 'C')
        self._log_sweep("C-relax", t0)

    def compute_jump(self) -> list:
        """||u_i - u_i(previous iteration)|| at the local C-points (mgrit.py:372-385)."""
        return self.backend.jump_norms(self._c_points(0))

    def compute_residual(self) -> list:
        """Per-C-point norms of r_i = Phi(u_{i-1}) - u_i on level 0 (mgrit.py:387-413); op 7 ghost refresh."""
        self._exchange(0, send_idx=self._last_slot(0) if self.last_is_f_point[0] else None,
                       recv_idx=0 if self.first_is_c_point[0] else None, dest=self.send_to[0], src=self.get_from[0], op=7)
        return self.backend.residual_norms(self._c_points(0))

    def convergence_criterion(self, iteration: int) -> None:
        """Global stopping value (mgrit.py:415-432): the per-point norms of all ranks, in time order, reduced with
        np.linalg.norm(ord=t_norm). gather+bcast of the reference becomes one all-gather of a few floats."""
        t0 = time.time()
        val = self.compute_residual() if self.conv_crit in (0, 2) else self.compute_jump()
        if self.global_conv_crit:
            if self.comm_time_size == 1:     # no Python loop over the (tens of thousands of) per-point values on the critical path
                allv = np.asarray(val, dtype=np.float64).ravel()
            else:
                parts = self.comm_time.allgather_object(np.asarray(val, dtype=np.float64).ravel().tolist())
                allv = np.array([x for part in parts for x in part])
            self.conv[iteration] = time_norm(allv, self.t_norm)
        else:   # local criterion (mgrit.py:434-455): every local point below the tolerance AND the previous rank has finished
            # op 6 carries HOW MANY leading ranks have left (the reference piggybacks a flag on every message instead): a
            # coarse level's sender may be any lower rank, and its farewell messages must be recognised as such
            rank = self.comm_time_rank
            for q, st in list(self._gone.items()):
                if st == 'draining':
                    self._gone[q] = 'done'      # one iteration has consumed their farewell messages
            if rank > 0 and not self.pre_finished[0]:
                known = int(self.comm_time.exchange(recv=(None, rank - 1)))
                for q in range(self._gone_count, known):
                    self._gone[q] = 'draining'
                if known > self._gone_count:
                    self._gone_count, self._drain_seen = known, set()
                if known >= rank:
                    self.pre_finished = [True, iteration]
            self.finished = [bool(self.pre_finished[0] and (all(v < self.tol for v in val) or iteration == self.iter_max)),
                             iteration]
            if rank < self.comm_time_size - 1 and self._announced_at is None:   # the successor stops listening afterwards
                self.comm_time.exchange(send=(rank + 1 if self.finished[0] else self._gone_count, rank + 1))
                if self.finished[0]:
                    self._announced_at = iteration
            self.conv[iteration] = time_norm(np.asarray(val, dtype=np.float64).ravel(), self.t_norm)
        self._log_sweep("Convergence criterion", t0)

    def _coarsest_u_unread(self) -> bool:
        """the step-by-step forward solve (mgrit.py:459-486) overwrites every point of the coarsest level but the first before
        anything reads it, so the sweeps above need not store u there; the time-parallel form works on the defect
        of the level's CURRENT values and reads them all"""
        return (type(self).forward_solve is Mgrit.forward_solve and
                not getattr(self.backend, "block_r", {}).get(self.lvl_max - 1))

    def forward_solve(self, lvl: int) -> None:
        """Sequential time stepping on level ``lvl`` (mgrit.py:459-486); op 5 = pipeline hand-off between owners."""
        t0 = time.time()
        if self._dry is None and getattr(self.backend, "block_sharded", {}).get(lvl):
            # time-parallel form on a sharded level: the first pass needs nothing from the rank before; what the
            # hand-over carries is the point and the amplitudes of a few sine modes, and the next rank waits for no more than the
            # recurrence over this rank's blocks (one tiny launch) -- not for a pipeline stage of hundreds of steps
            be = self.backend
            be.block_solve(lvl, 1)
            if self.get_from[lvl] != -99:
                self._exchange(lvl, recv_idx=0, src=self.get_from[lvl], op=5)
            be.block_solve(lvl, 2)
            if self.send_to[lvl] != -99:
                self._exchange(lvl, send_idx=int(self.index_local[lvl][-1]), dest=self.send_to[lvl], op=5)
            be.block_solve(lvl, 4)
            self._log_sweep("Forward solve", t0)
            return
        if self.get_from[lvl] != -99:
            self._exchange(lvl, recv_idx=0, src=self.get_from[lvl], op=5)
        n = len(self.t[lvl])
        if n > 1:
            self.backend.relax(lvl, self._cached(('chain', lvl), lambda: [(1, n - 1)]), 'CHAIN')
        if self.send_to[lvl] != -99:
            self._exchange(lvl, send_idx=int(self.index_local[lvl][-1]), dest=self.send_to[lvl], op=5)
        self._log_sweep("Forward solve", t0)

    def fas_residual(self, lvl: int) -> None:
        """Inject the C-points and the FAS right-hand side into level lvl+1 (mgrit.py:488-549). op 3 = ghost refresh on
        lvl, op 4 = last local point of lvl+1 to the next owner's ghost."""
        t0 = time.time()
        held = self.__dict__.get('_head_done')
        if held:      # this sweep injects the first time point itself: what the levels below hold of it (_head) is out of date
            for key in [k for k in held if k[0] >= lvl]:
                held.discard(key)
        if getattr(self.backend, "can_fuse_fas", None) is not None and self.backend.can_fuse_fas(lvl):
            self._fas_residual_fused(lvl)
            self._log_sweep("Fas residual", t0)
            return
        self.backend.restrict_u(lvl, self._pairs(lvl, skip_first=False))
        self._exchange(lvl, send_idx=self._last_slot(lvl) if self.last_is_f_point[lvl] else None,
                       recv_idx=0 if self.first_is_c_point[lvl] else None, dest=self.send_to[lvl], src=self.get_from[lvl], op=3)
        self._exchange(lvl + 1,
                       send_idx=int(self.index_local[lvl + 1][-1]) if self.send_to[lvl + 1] >= 0 else None,
                       recv_idx=0 if self.get_from[lvl + 1] >= 0 else None,
                       dest=self.send_to[lvl + 1], src=self.get_from[lvl + 1], op=4)
        self.backend.copy_u_to_v(lvl + 1)
        self.backend.fas_rhs(lvl, self._pairs(lvl, skip_first=True))
        self._log_sweep("Fas residual", t0)

    def _fas_residual_fused(self, lvl: int, skip_triples: bool = False, with_f_relax: bool = False) -> None:
        """Same sweep as fas_residual with the device backend's fused kernel: every local C-point whose previous
        C-point is local too is handled in one pass (restriction, clone into v, FAS right-hand side); the first local
        C-point goes through the separate kernels because its v_{j-1} is the ghost that arrives with op 4."""
        be = self.backend
        all_pairs = self._pairs(lvl, skip_first=False)
        head = self._cached(('pair_head', lvl), lambda: all_pairs[:1])

        def build_triples():   # (fine slot, fine slot of the C-point before it, coarse slot)
            P = as_index_array(all_pairs, 2)
            return IndexArray(np.column_stack((P[1:, 0], P[:-1, 0], P[1:, 1])), width=3)
        triples = self._cached(('triples', lvl), build_triples)
        self._exchange(lvl, send_idx=self._last_slot(lvl) if self.last_is_f_point[lvl] else None,
                       recv_idx=0 if self.first_is_c_point[lvl] else None, dest=self.send_to[lvl], src=self.get_from[lvl], op=3)
        be.restrict_u(lvl, head)
        if not skip_triples:      # (skip: the whole-level pass of the rank's complete intervals has done exactly these)
            if with_f_relax:      # the F-relaxation of the triples' intervals is part of the pass (their F-points are not stored)
                be.fas_fused(lvl, triples, with_f_relax=True)
            else:
                be.fas_fused(lvl, triples)
        self._exchange(lvl + 1,
                       send_idx=int(self.index_local[lvl + 1][-1]) if self.send_to[lvl + 1] >= 0 else None,
                       recv_idx=0 if self.get_from[lvl + 1] >= 0 else None,
                       dest=self.send_to[lvl + 1], src=self.get_from[lvl + 1], op=4)
        be.copy_pairs_u_to_v(lvl, head)
        if self._ghost[lvl + 1]:  # v ghost = clone of the received u ghost; then the first local pair, unfused
            be.copy_pairs_u_to_v(lvl, self._cached(('pair_ghost', lvl), lambda: [(all_pairs[0][0], 0)] if all_pairs else []))
            be.fas_rhs(lvl, head)

    def error_correction(self, lvl: int) -> None:
        """u^l at C-points += P(u^{l+1} - v^{l+1}) (mgrit.py:715-726)."""
        self.backend.error_correction(lvl, self._pairs(lvl, skip_first=True))

    def nested_iteration(self) -> None:
        """Initial guess from the coarsest level upwards (mgrit.py:551-566)."""
        self.forward_solve(self.lvl_max - 1)
        for lvl in range(self.lvl_max - 2, -1, -1):
            self.backend.interpolate(lvl, self._pairs(lvl, skip_first=True))
            if lvl > 0:
                self.iteration(lvl=lvl, cycle_type='V', iteration=0, first_f=True)

    # ------------------------------------------------------------------------------------------------
    # solve (mgrit.py:590-646) -- log lines keep the reference's wire format (tests/mpi/mpi.py parses "conv:")
    # ------------------------------------------------------------------------------------------------
    def ouput_run_information(self) -> None:
        rows = [('time interval', '[' + str(self.problem[0].t[0]) + ', ' + str(self.problem[0].t[-1]) + ']'),
                ('number of time points ', str(len(self.problem[0].t))),
                ('max dt ', str(np.max(self.problem[0].t[1:] - self.problem[0].t[:-1]))),
                ('number of levels', str(self.lvl_max)),
                ('coarsening factors', str(self.m[:-1])),
                ('relaxation weight', str(self.weight_c)),
                ('cf_iter', str(self.cf_iter[:self.lvl_max - 1])),
                ('nested iteration', str(self.nes_it)),
                ('cycle type', str(self.cycle_type)),
                ('stopping tolerance', str(self.tol)),
                ('time communicator size', str(self.comm_time_size)),
                ('space communicator size', str(self.comm_space_size)),
                ('convergence criterion', str(self.conv_crit))]
        self.log_info('\n'.join(['Run parameter overview'] + ['  ' + '{0: <25}'.format(k) + ' : ' + v for k, v in rows]))

    def _exchange_stats_begin(self):
        """the time communicator's counters are per process (several solvers may share it): a solve reports its own share"""
        st = getattr(self.comm_time, "stats", None)
        self._stats0 = dict(st) if st is not None else None
        self.exchange_stats = None

    def _exchange_stats_end(self):
        st = getattr(self.comm_time, "stats", None)
        if st is not None and getattr(self, "_stats0", None) is not None:
            self.exchange_stats = {k: v - self._stats0.get(k, 0) for k, v in st.items()}

    def solve(self) -> dict:
        self._exchange_stats_begin()
        if self.pipeline_depth() > 0:
            return self._solve_pipelined()
        self.log_info("Start solve")
        solve_start = time.time()
        # where the host's time goes (milliseconds, summed over the iterations): enqueue = Mgrit.iteration() issuing the cycle's
        # launches, wait = the device finishing them, criterion = the stopping value (its launch unless the way up has left the
        # sums, the read-back, the norm), log = the reference's per-iteration line, final = the F-relaxation that puts every
        # F-point in place + the last wait, tail = drain / barrier / run information
        bd = self.solve_breakdown = dict(enqueue=0.0, wait=0.0, criterion=0.0, log=0.0, final=0.0, tail=0.0, iterations=0)
        clock = time.perf_counter
        for iteration in range(self.iter_max):
            self.solve_iter = iteration + 1
            it_start = time.time()
            c0 = clock()
            self.iteration(lvl=0, cycle_type=self.cycle_type, iteration=iteration, first_f=True)
            c1 = clock()
            self.backend.sync()
            c2 = clock()
            it_stop = time.time()
            self.convergence_criterion(iteration=iteration + 1)
            c3 = clock()
            now, before = self.conv[iteration + 1], self.conv[iteration]
            if logging.getLogger().isEnabledFor(logging.INFO):
                with np.errstate(divide='ignore', invalid='ignore'):     # a rank whose residuals are exactly zero
                    factor = '-' if iteration == 0 else str(now / before)
                label = f" | conv: {now}" if self.global_conv_crit else f" | conv on process {self.comm_time_size - 1}: {now}"
                self.log_info('{0: <7}'.format(f"iter {iteration + 1}") + '{0: <32}'.format(label) +
                              '{0: <37}'.format(f" | conv factor: {factor}") +
                              '{0: <35}'.format(f" | runtime: {it_stop - it_start} s"))
            if self.output_fcn is not None and self.output_lvl == 2:
                self.output_fcn(self)
            c4 = clock()
            bd["enqueue"] += 1e3 * (c1 - c0); bd["wait"] += 1e3 * (c2 - c1); bd["criterion"] += 1e3 * (c3 - c2); bd["log"] += 1e3 * (c4 - c3)
            bd["iterations"] += 1
            if now < self.tol or iteration == self.iter_max - 1:
                if self.global_conv_crit or (self.finished[0] and self.pre_finished[0]) or iteration == self.iter_max - 1:
                    if not self.global_conv_crit and self.comm_time_size > 1:
                        self._leave_local(iteration)
                    break
        c0 = clock()
        getattr(self.backend, 'materialise', lambda: None)()   # C-point storage on the way up: every F-point in place again
        self.backend.sync()
        c1 = clock()
        getattr(self.comm_time, 'drain', lambda: None)()
        self.comm_time.barrier()
        self.runtime_solve = time.time() - solve_start
        self._exchange_stats_end()
        self.log_info(f"Solve took {self.runtime_solve} s")
        if self.output_fcn is not None and self.output_lvl == 1:
            self.output_fcn(self)
        self.ouput_run_information()
        bd["final"], bd["tail"] = 1e3 * (c1 - c0), 1e3 * (clock() - c1)
        return {'conv': self.conv[np.where(self.conv != 0)], 'time_setup': self.runtime_setup,
                'time_solve': self.runtime_solve}
