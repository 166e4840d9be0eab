"""The solve loop on several ranks: stopping values looked at ``pipeline_depth`` iterations late (so that a rank does not idle
until the last rank's residual is known), and the farewell walk of a rank that leaves a LOCAL stopping criterion early
(reference mgrit.py:434-455,627-635,648-691). Host logic only -- a mixin of ``Mgrit`` (core/mgrit.py), split out of it in round 4."""
import time

import numpy as np

from pymgrit_amd.core.options import options


def _library():
    from pymgrit_amd.core.mgrit import Mgrit     # (late: mgrit.py imports this module)
    return Mgrit


def time_norm(values, ord):
    from pymgrit_amd.core.mgrit import time_norm as f
    return f(values, ord)


class PipelinedLoop:
    # ------------------------------------------------------------------------------------------------
    # Pipelined solve on several ranks. The coarsest-level solve is a pipeline over the ranks (op 5), so inside ONE
    # iteration rank r idles while the ranks before it step through their part of the coarsest grid, and again afterwards
    # until the last rank is done and the global stopping value is known. Nothing but that stopping value keeps rank r
    # from starting the next iteration early: its sweeps only need its own points and ghost points of rank r-1, which is
    # ahead of it. So the stopping values are collected asynchronously and looked at `depth` iterations late:
    #   * before iteration it (0-based) starts, conv[it - depth] must be known; if it is below tol nobody starts it.
    #     The decision depends only on global values, so every rank executes the same set of iterations and every
    #     posted message is matched;
    #   * the result must be the state of the FIRST iteration whose conv is below tol (the reference stops there): each
    #     iteration ends with F-relax(0), so the level-0 C-points of an iteration determine its whole end state; they
    #     are snapshotted (a copy of 1/m of u[0]) and on a late stop the solver restores the C-points of the iteration
    #     before, rebuilds the F-points with one F-relax and repeats that one iteration -- bit-identical to the
    #     unpipelined run, internal levels included.
    # ------------------------------------------------------------------------------------------------
    def pipeline_depth(self) -> int:
        if self._pipeline_request is not None:
            want = int(self._pipeline_request)
        else:   # default 4 (the last of 8 ranks lags about two cycles, its values are posted one trip late); PYMGRIT_AMD_PIPELINE_DEPTH overrides it without touching the script
            want = int(options.pipeline_depth)
        usable = (self.comm_time_size > 1 and self.lvl_max > 1 and self.conv_crit == 0 and
                  hasattr(self.comm_time, "iallgather_floats") and getattr(self.comm_time, "async_gather", True) and
                  not (self.output_fcn is not None and self.output_lvl == 2) and
                  type(self).convergence_criterion is _library().convergence_criterion and
                  type(self).iteration is _library().iteration)
        return max(want, 0) if usable else 0

    def _pl_state(self):
        if self._pl is None:
            depth = self.pipeline_depth()
            counts = self.comm_time.allgather_object(len(self._c_points(0)))
            self._pl = {"depth": depth, "executed": 0, "resolved": 0, "pending": {}, "open": None, "slots": depth + 2,
                        "max_count": max(max(counts), 1), "snap_points": [int(i) for i in self.index_local_c[0]]}
            self.backend.snapshot_cpoints(0, self._pl["snap_points"])
        return self._pl

    def _pl_resolve(self, c):
        """conv[c] (1-based iteration count) from the gather posted after iteration c"""
        pl = self._pl
        while pl["resolved"] < c:
            k = pl["resolved"] + 1
            parts = pl["pending"].pop(k).result()
            self.conv[k] = time_norm(np.array([x for part in parts for x in part]), self.t_norm)
            factor = '-' if k == 1 else str(self.conv[k] / self.conv[k - 1])
            self.log_info('{0: <7}'.format(f"iter {k}") + '{0: <32}'.format(f" | conv: {self.conv[k]}") +
                          '{0: <37}'.format(f" | conv factor: {factor}") + '{0: <35}'.format(" | runtime: pipelined"))
            pl["resolved"] = k
        return self.conv[c]

    def _pl_advance(self, n, stop_on_tol=True):
        """run up to n more iterations; returns the 1-based index of the iteration that met tol (None if none did)"""
        pl = self._pl_state()
        depth = pl["depth"]
        for _ in range(n):
            it = pl["executed"]
            if it >= self.iter_max:
                break
            gate = it - depth
            if gate >= 1:
                if pl.get("open") is not None and pl["open"][0] <= gate:
                    self._pl_post()   # depth 0: the value looked at is the one of the trip just finished
                if self._pl_resolve(gate) < self.tol and stop_on_tol:
                    return gate
            self.solve_iter = it + 1
            # (the C-point snapshot of this iteration, written by the cycle's own way up where that is the whole-level pass)
            getattr(self.backend, "mirror_cpoints", lambda *a: None)((it + 1) % pl["slots"], pl["snap_points"])
            self.iteration(lvl=0, cycle_type=self.cycle_type, iteration=it, first_f=True)
            # residual of this iteration: launched now, read one trip later, so the host never waits for the device to
            # finish the iteration it has just queued (it stays one iteration ahead of it)
            self._exchange(0, send_idx=self._last_slot(0) if self.last_is_f_point[0] else None,
                           recv_idx=0 if self.first_is_c_point[0] else None, dest=self.send_to[0], src=self.get_from[0], op=7)
            handle = self.backend.residual_begin(self._c_points(0))
            pl["executed"] = it + 1
            self.backend.snapshot_cpoints((it + 1) % pl["slots"], pl["snap_points"])
            self._pl_post()
            pl["open"] = (it + 1, handle)
        return None

    def _pl_post(self):
        """hand the residual values of the previous trip to the asynchronous gather"""
        pl = self._pl
        if pl.get("open") is not None:
            c, handle = pl["open"]
            pl["pending"][c] = self.comm_time.iallgather_floats(self.backend.residual_end(handle), pl["max_count"])
            pl["open"] = None

    def _pl_finish(self, stop_on_tol=True):
        """resolve everything posted; roll back to the first iteration below tol. Returns the iteration count kept."""
        pl = self._pl_state()
        self._pl_post()
        stop = None
        for c in range(1, pl["executed"] + 1):    # the first iteration below tol, resolving values only as far as needed
            if self._pl_resolve(c) < self.tol and stop_on_tol:
                stop = c
                break
        keep = pl["executed"] if stop is None else stop
        for k in list(pl["pending"]):       # gathers of discarded iterations: complete them (every rank posted them)
            pl["pending"].pop(k).result()
        if keep < pl["executed"]:
            self.backend.restore_cpoints((keep - 1) % pl["slots"], pl["snap_points"])
            if keep - 1 >= 1:
                self.f_relax(lvl=0)
            self.iteration(lvl=0, cycle_type=self.cycle_type, iteration=keep - 1, first_f=True)
            pl["executed"] = pl["resolved"] = keep
        getattr(self.backend, "mirror_cpoints", lambda *a: None)(None, None)
        self.conv[keep + 1:] = 0.0
        self.solve_iter = keep
        return keep

    def _solve_pipelined(self) -> dict:
        self.log_info("Start solve")
        solve_start = time.time()
        self._pl = None
        self._pl_advance(self.iter_max)
        self._pl_finish()
        getattr(self.backend, 'materialise', lambda: None)()   # C-point storage on the way up: every F-point in place again
        self.backend.sync()
        getattr(self.comm_time, 'drain', lambda: None)()
        self.comm_time.barrier()
        self.runtime_solve = time.time() - solve_start
        self._exchange_stats_end()
        self.log_info(f"Solve took {self.runtime_solve} s")
        if self.output_fcn is not None and self.output_lvl == 1:
            self.output_fcn(self)
        self.ouput_run_information()
        return {'conv': self.conv[np.where(self.conv != 0)], 'time_setup': self.runtime_setup,
                'time_solve': self.runtime_solve}

    # ------------------------------------------------------------------------------------------------
    # Local stopping criteria on several ranks (mgrit.py:434-455,627-635,648-691). Ranks leave the solve loop one after
    # the other, in rank order. The reference lets a leaving rank post one last message per (level, op) with its final
    # boundary values (clean_up); its successor, still iterating, picks each of them up the first time it reaches that
    # receive and skips the receive from then on -- the ghost values stay what they are. The messages of the reference are
    # matched by tag; here a channel is first-in first-out, so the leaving rank posts them in the order in which its
    # successor's next iteration reaches the receives: it walks once through its own iteration with every sweep switched
    # off, sending at the first occurrence of each send point (_dry = 'send'). A rank that leaves in the same iteration as
    # its predecessor has nothing left to iterate and takes the messages off the channel the same way (_dry = 'recv').
    # ------------------------------------------------------------------------------------------------
    class _NoSweeps:
        """stands in for the backend during a dry walk: no sweep does anything"""

        def __getattr__(self, name):
            if name.startswith("can_"):
                raise AttributeError(name)
            return lambda *a, **k: None

    def _dry_walk(self, mode, iteration):
        self._real_backend, self.backend = self.backend, _library()._NoSweeps()
        self._dry, self._drain_seen = mode, set()
        try:
            self.iteration(lvl=0, cycle_type=self.cycle_type, iteration=iteration, first_f=True)
            if self.conv_crit == 2:     # op 7 belongs to the residual criterion; the jump criterion exchanges nothing
                self._exchange(0, send_idx=self._last_slot(0) if self.last_is_f_point[0] else None,
                               recv_idx=0 if self.first_is_c_point[0] else None, dest=self.send_to[0], src=self.get_from[0], op=7)
        finally:
            self.backend, self._real_backend, self._dry = self._real_backend, None, None

    def _leave_local(self, iteration):
        """clean_up (mgrit.py:648-691) of a rank that leaves the solve loop under a local criterion"""
        if self.comm_time_rank < self.comm_time_size - 1 and self._announced_at == iteration + 1:
            self._dry_walk('send', iteration + 1)   # announced earlier: the successor has already been served
        if any(st == 'draining' for st in self._gone.values()):   # ranks that left in this very iteration: take their messages
            self._dry_walk('recv', iteration + 1)
            for q in self._gone:
                self._gone[q] = 'done'

