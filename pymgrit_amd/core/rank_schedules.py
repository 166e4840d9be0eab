"""Index lists of the whole-level passes and of a rank's share of them: which intervals of a level one launch may take (one rank,
aligned ranks), what the closing C-point of every interval needs on the coarser level, the boundary rows that travel between the
exchange points of the reference (mgrit.py:693-713). Host logic only -- a mixin of ``Mgrit`` (core/mgrit.py), split out of it
in round 4; every list is built once per (kind, level) and cached (``Mgrit._cached``)."""
import numpy as np

from pymgrit_amd.core.layout import IndexArray, as_index_array
from pymgrit_amd.core.options import options


def _library():
    from pymgrit_amd.core.mgrit import Mgrit     # (late: mgrit.py imports this module)
    return Mgrit


class RankSchedules:
    def _detect_aligned(self) -> bool:
        """Several ranks whose shares of the time grid all END ON A C-POINT of every level (BASELINE configs 2-5 on 2 / 4 / 8
        ranks: nt - 1 a multiple of the rank count times every coarsening factor): a rank's local grid then looks
        like a one-rank grid -- slot 0 (the ghost point, a C-point of every level owned by the rank before) takes the place of
        the first time point, which nobody relaxes or corrects, and whole intervals follow. Such a rank runs the one-rank
        machinery (whole-level passes, C-point storage, pre-relaxed C-points, the planned cycle) with the exchange points of
        the reference refreshing slot 0 (ops 0 / 4 / 5; ops 1 / 2 / 3 / 7 never fire). Decided by all ranks together."""
        size, rank = self.comm_time_size, self.comm_time_rank
        if size == 1 or not getattr(self.backend, "device_links", False) or not self.global_conv_crit or \
                options.no_aligned:
            if size > 1 and getattr(self.backend, "device_links", False):
                self.comm_time.allgather_object(False)
            return False
        ok = self.lvl_max >= 2
        for lvl in range(self.lvl_max):
            if not ok:
                break
            n_own = len(self.index_local[lvl])
            ok = n_own >= 1 and (rank == 0 or (bool(self._ghost[lvl]) and self.get_from[lvl] == rank - 1)) and \
                (rank == size - 1 or self.send_to[lvl] == rank + 1) and (rank != 0 or self.get_from[lvl] < 0) and \
                (rank != size - 1 or self.send_to[lvl] < 0)
            if ok and lvl < self.lvl_max - 1:
                ok = (not self.comm_front[lvl] and not self.comm_back[lvl] and not self.first_is_c_point[lvl] and
                      not self.last_is_f_point[lvl] and len(self.index_local_c[lvl]) >= 1 and
                      bool(self._is_c_local[lvl][-1]) and bool(self._is_c_local[lvl][0]) and
                      (rank == 0 or bool(self.first_is_f_point[lvl])) and (rank == size - 1 or bool(self.last_is_c_point[lvl])))
        return all(self.comm_time.allgather_object(bool(ok)))

    def _one_rank_like(self) -> bool:
        """no exchange point falls INSIDE a whole-level pass: one rank, or ranks whose shares end on C-points (_detect_aligned)"""
        return self.comm_time_size == 1 or getattr(self, "_aligned", False)

    def _xpairs(self, lvl):
        """(fine slot, coarse slot) of the level's local C-points including the frozen one in front: the first point of the time
        grid on rank 0 (it is an owned C-point there), the ghost point (slot 0 on both levels) on an aligned rank > 0"""
        def build():
            own = self._pairs(lvl, skip_first=False)
            if self.comm_time_rank > 0 and getattr(self, "_aligned", False):
                return IndexArray(np.concatenate((np.zeros((1, 2), dtype=np.int64), as_index_array(own, 2))), width=2)
            return own
        return self._cached(('xpair', lvl), build)

    def _x0(self, lvl, send_row=None, staged=None):
        """op 0, the exchange point at the head of an F-relaxation (mgrit.py:304-311): the last local C-point to the next
        owner's ghost slot. send_row: the row that HOLDS that value when it is not the point's own row (pre-relaxed C-points);
        staged: the (fine, coarse) pair whose corrected value is sent before the pass that corrects it in place has run"""
        send = self._last_slot(lvl) if self.last_is_c_point[lvl] else None
        recv = 0 if self.first_is_f_point[lvl] else None
        if send is None and recv is None:
            return
        if staged is not None and send is not None:
            self.backend.exchange_staged(lvl, 0, staged, dest=self.send_to[lvl], recv_idx=recv, src=self.get_from[lvl])
        else:
            self.backend.exchange(lvl, 0, send_idx=(send_row if (send_row is not None and send is not None) else send),
                                  dest=self.send_to[lvl], recv_idx=recv, src=self.get_from[lvl], raw=True)

    def _x4(self, lvl):
        """op 4 of fas_residual(lvl) (mgrit.py:511-517): the last local point of lvl+1 to the next owner's ghost slot, then the
        clone of the received ghost into v (mgrit.py:520)"""
        up = lvl + 1
        send = int(self.index_local[up][-1]) if self.send_to[up] >= 0 else None
        recv = 0 if self.get_from[up] >= 0 else None
        if send is None and recv is None:
            return
        self.backend.exchange(up, 4, send_idx=send, dest=self.send_to[up], recv_idx=recv, src=self.get_from[up], raw=True)
        if recv is not None:
            self.backend.copy_pairs_u_to_v(lvl, self._cached(('pair_ghost_x', lvl), lambda: [(0, 0)]))

    def _head(self, lvl, head, which):
        """one rank: the first point of the time grid is relaxed and corrected by nobody, so its injection into level lvl+1
        (u, then the clone into v: mgrit.py:498-500, 520) writes the same row in every cycle -- two launches of a few microseconds
        that a small hierarchy notices. Done once, and again after anything has been written into the slabs from outside."""
        done = self.__dict__.setdefault('_head_done', set())
        gen = getattr(self.backend, "write_generation", lambda: None)()
        if gen is None:
            done.clear()
        elif gen != self.__dict__.get('_head_gen'):
            done.clear()
            self._head_gen = gen
        if (lvl, which) in done:
            return
        if which == 'u':
            self.backend.restrict_u(lvl, head)
            for key in [k for k in done if k[0] > lvl or k == (lvl, 'v')]:   # the levels below copy from the row just written (nested
                done.discard(key)                                           # iteration reaches them before level 0 has injected)
        else:
            self.backend.copy_pairs_u_to_v(lvl, head)
        if gen is not None:      # (while a cycle is being recorded: the recorded cycle runs right away and does it; the cycles
            done.add((lvl, which))   # recorded after it leave it out; _planned keys its plans by the write generation)

    def _up(self, lvl, fused, gen=None):
        """error correction + F-relaxation of level lvl on the way up (mgrit.py:283-284), in the most fused form available"""
        if gen is not None:     # (the way down of this cycle was mgrit_hip_gen_down over the same intervals)
            res = lvl == 0 and self.conv_crit in (0, 2)
            if self.comm_time_size > 1:     # aligned ranks: op 0 of the F-relaxation carries the CORRECTED last C-point (see below)
                self._x0(lvl, staged=self._cached(('pair_last', lvl), lambda: [self._xpairs(lvl)[-1]]))
            self.backend.gen_up(lvl, gen, residual=res)
            if res:
                self.backend.residual_ready(self._c_points(0))
            return
        shard = self._rank_intervals_up(lvl) if (fused is None and lvl == 0 and self.comm_time_size > 1) else None
        if shard is not None:
            # several ranks: correction + F-relaxation + residual sums of the rank's complete intervals in one pass; the first
            # local C-point is corrected up front (the interval it closes belongs to the rank before), the partial intervals at
            # the two ends follow as plain F-relaxations with their exchange points; the residual of the first local C-point is
            # the residual kernel's (it needs the ghost point of op 7)
            intervals, c0_pair, edge_runs, n_head = shard
            buf = self.backend.residual_reserve(self._c_points(0), n_head)
            self.backend.error_correction(lvl, c0_pair)
            self.backend.ec_relax_res_to(lvl, intervals, buf)
            self.f_relax(lvl=lvl, runs=edge_runs)
            return
        if fused is not None and lvl == 0 and self.conv_crit in (0, 2):   # correction + F-relaxation + the residual check's sums
            if self.comm_time_size > 1:     # aligned ranks: op 0 of the F-relaxation (mgrit.py:306-310) carries the CORRECTED last
                # C-point, which the pass corrects in place only later
                self._x0(lvl, staged=self._cached(('pair_last', lvl), lambda: [self._xpairs(lvl)[-1]]))
            self.backend.ec_relax_res(lvl, fused)
            self.backend.residual_ready(self._c_points(0))
        elif lvl > 0 and self._level_intervals(lvl, up=True) is not None:   # coarser level: the same pass with g, no residual
            if self.comm_time_size > 1:
                self._x0(lvl, staged=self._cached(('pair_last', lvl), lambda: [self._xpairs(lvl)[-1]]))
            self.backend.ec_relax_res(lvl, self._level_intervals(lvl, up=True))
        elif self._can_fuse_ec(lvl):
            self._ec_f_relax(lvl)
        else:
            self.error_correction(lvl=lvl)
            self.f_relax(lvl=lvl)

    def _rank_intervals_up(self, lvl):
        """several ranks, level 0, residual criterion: (intervals, c0_pair, edge_runs, n_head) for the way up on the rank's complete
        intervals (mgrit_hip_ec_relax_res_to), else None. res_pos of an interval = position of its closing C-point among the
        rank's relaxed C-points (the order of the residual values); n_head = how many of those lie in front of the first closing
        point (1: the first local C-point on ranks > 0; 0 on rank 0)."""
        def build():
            down = self._rank_intervals(lvl)
            if (down is None or self.conv_crit != 0 or
                    not self._can_fuse_ec(lvl) or getattr(self.backend, "residual_reserve", None) is None):
                return [None]
            ivals, c0_run, edge = (list(x) for x in down)
            cpts = list(self._c_points(lvl))
            pos = {c: i for i, c in enumerate(cpts)}
            if any(iv[1] not in pos for iv in ivals):
                return [None]
            n_head = pos[ivals[0][1]]
            if n_head != len(c0_run) or [pos[iv[1]] for iv in ivals] != list(range(n_head, n_head + len(ivals))) or \
                    n_head + len(ivals) != len(cpts):
                return [None]
            up = [(cs, ce, jcs, jce, pos[ce], keep) for (cs, ce, jcs, jce, _, keep) in ivals]
            corrected = dict(self._pairs(lvl, skip_first=True))
            c0_pair = [(c0_run[0][0], corrected[c0_run[0][0]])] if c0_run else []
            return [(up, c0_pair, edge, n_head)]
        got = self._cached(('rank_intervals_up', lvl), build)[0]
        if got is None:
            return None
        return (self._cached(('riu_list', lvl), lambda: got[0]), self._cached(('riu_c0', lvl), lambda: got[1]),
                self._cached(('riu_edge', lvl), lambda: got[2]), got[3])

    def _coarse_down_rank(self, lvl):
        """several ranks, a level > 0 the finer level's FAS sweep has just filled: (fc_runs, c0_run, edge_runs) when the rank's
        complete intervals can take the two coarse-level passes (relax mode FC, fas_fused with_f_relax), else None"""
        def build():
            be = self.backend
            own = all(getattr(type(self), name) is getattr(_library(), name) for name in
                      ("iteration", "f_relax", "c_relax", "fas_residual", "error_correction", "_exchange", "_ec_f_relax",
                       "_fas_residual_fused", "_relax_f"))
            can = getattr(be, "can_fuse_coarse_down", None)
            if not (not options.no_rank_fusion and own and 0 < lvl < self.lvl_max - 1 and
                    self.weight_c == 1.0 and self.cf_iter[lvl] == 1 and self.global_conv_crit and
                    not getattr(self, "_sweep_timing", False) and can is not None and can(lvl)):
                return [None]
            pairs = list(self._pairs(lvl, skip_first=False))
            if len(pairs) < 3:
                return [None]
            c0, ck = pairs[0][0], pairs[-1][0]
            runs = [tuple(r) for r in self._f_runs(lvl)]
            inner = [(pairs[k][0] + 1, pairs[k + 1][0] - pairs[k][0] - 1) for k in range(len(pairs) - 1)]
            edge = [r for r in runs if r[0] < c0 or r[0] > ck]
            relaxed = set(self._c_points(lvl))
            if (any(ln < 1 for _, ln in inner) or sorted(inner + edge) != sorted(runs) or len(edge) > 2 or
                    any(p[0] not in relaxed for p in pairs[1:]) or (c0 in relaxed and c0 < 1)):
                return [None]
            return [([(st, ln + 1) for st, ln in inner], [(c0, 1)] if c0 in relaxed else [], edge)]
        got = self._cached(('coarse_down_rank', lvl), build)[0]
        if got is None:
            return None
        return (self._cached(('cdr_fc', lvl), lambda: got[0]), self._cached(('cdr_c0', lvl), lambda: got[1]),
                self._cached(('cdr_edge', lvl), lambda: got[2]))

    def _rank_intervals(self, lvl):
        """several ranks, level 0: (intervals, c0_run, edge_runs) when the rank's complete intervals -- both C-points local -- can
        take the down pass mgrit_hip_cf_fas, else None. intervals: as _level_intervals (every coarse row kept: the exchanges and
        the generic sweeps of the coarser level read them); c0_run: the first local C-point as a run list for the generic
        C-relaxation (empty on rank 0, whose first point is never relaxed); edge_runs: the F-runs in front of the first and behind
        the last local C-point (the partial intervals this rank shares with its neighbours)."""
        def build():
            be = self.backend
            own = all(getattr(type(self), name) is getattr(_library(), name) for name in
                      ("iteration", "f_relax", "c_relax", "fas_residual", "error_correction", "_exchange", "_ec_f_relax",
                       "_fas_residual_fused", "_relax_f"))
            can = getattr(be, "can_fuse_level", None)
            if not (not options.no_rank_fusion and own and lvl == 0 and self.lvl_max > 1 and
                    self.weight_c == 1.0 and self.global_conv_crit and not getattr(self, "_sweep_timing", False) and
                    can is not None and can(lvl) and getattr(be, "can_fuse_fas", lambda l: False)(lvl)):
                return [None]
            pairs = list(self._pairs(lvl, skip_first=False))         # (fine slot, coarse slot) of every local C-point
            if len(pairs) < 3:
                return [None]
            c0, ck = pairs[0][0], pairs[-1][0]
            runs = [tuple(r) for r in self._f_runs(lvl)]
            inner = [(pairs[k][0] + 1, pairs[k + 1][0] - pairs[k][0] - 1) for k in range(len(pairs) - 1)]
            edge = [r for r in runs if r[0] < c0 or r[0] > ck]
            if any(ln < 1 for _, ln in inner) or sorted(inner + edge) != sorted(runs) or len(edge) > 2:
                return [None]
            relaxed = set(self._c_points(lvl))
            if any(p[0] not in relaxed for p in pairs[1:]):
                return [None]
            first_relaxed = c0 in relaxed                            # False on rank 0 (global point 0)
            if first_relaxed and c0 < 1:
                return [None]
            ivals = [(pairs[k][0], pairs[k + 1][0], pairs[k][1] if (k >= 1 or first_relaxed) else -1, pairs[k + 1][1], k, 3)
                     for k in range(len(pairs) - 1)]
            return [(ivals, [(c0, 1)] if first_relaxed else [], edge)]
        got = self._cached(('rank_intervals', lvl), build)[0]
        if got is None:
            return None
        return (self._cached(('rank_intervals_list', lvl), lambda: got[0]), self._cached(('rank_c0', lvl), lambda: got[1]),
                self._cached(('rank_edge', lvl), lambda: got[2]))

    def _coarse_down(self, lvl):
        """(fc_runs, triples, head, skip_coarse_u) when the way down of level lvl > 0 can run as two passes (relax mode FC,
        fas_fused with_f_relax; backend_hip.can_fuse_coarse_down), else None: one rank, the library's own sweeps, weight 1,
        cf_iter 0 or 1, every F-point between two local C-points."""
        def build():
            be = self.backend
            own = all(getattr(type(self), name) is getattr(_library(), name) for name in
                      ("iteration", "f_relax", "c_relax", "fas_residual", "error_correction", "_exchange", "_ec_f_relax",
                       "_fas_residual_fused"))
            can = getattr(be, "can_fuse_coarse_down", None)
            if not (own and self._one_rank_like() and self.weight_c == 1.0 and 0 < lvl < self.lvl_max - 1 and
                    self.cf_iter[lvl] in (0, 1) and not getattr(self, "_sweep_timing", False) and can is not None and can(lvl)):
                return [None]
            pairs = self._xpairs(lvl)
            P, R = as_index_array(pairs, 2), as_index_array(self._f_runs(lvl), 2)
            if P.shape[0] < 2 or P[0, 0] != 0:
                return [None]
            want = np.stack((P[:-1, 0] + 1, P[1:, 0] - P[:-1, 0] - 1), axis=1)   # one run of F-points between two C-points
            if R.shape != want.shape or not np.array_equal(R, want) or (want[:, 1] < 1).any():
                return [None]
            fc_runs = IndexArray(want + np.array([0, 1]), width=2)             # the F-points and the C-point closing them
            triples = IndexArray(np.column_stack((P[1:, 0], P[:-1, 0], P[1:, 1])), width=3)
            skip_u = lvl + 1 == self.lvl_max - 1 and self._coarsest_u_unread()
            return [(fc_runs, triples, pairs[:1] if self.comm_time_rank == 0 else [], skip_u)]
        return self._cached(('coarse_down', lvl), build)[0]

    def _level_intervals(self, lvl, up=False):
        """[(cstart, cend, cstart_coarse, cend_coarse, res_pos, keep)] of level lvl when its sweeps can run as whole-level passes
        (mgrit_hip_cf_fas / mgrit_hip_ec_relax_res), else None: one rank (no exchange point inside the pass), the library's own
        sweeps, weight 1, and a level whose F-points all lie between two local C-points."""
        def build():
            be = self.backend
            own = all(getattr(type(self), name) is getattr(_library(), name) for name in
                      ("iteration", "f_relax", "c_relax", "fas_residual", "error_correction", "compute_residual", "_exchange",
                       "_ec_f_relax"))
            # up: the pass of the way up alone (error correction + F-relaxation), which exists for every level pair
            can = getattr(be, "can_fuse_level_up" if up else "can_fuse_level", None)
            if not (own and self._one_rank_like() and self.weight_c == 1.0 and lvl < self.lvl_max - 1 and
                    not getattr(self, "_sweep_timing", False) and     # per-sweep debug lines: sweep by sweep
                    can is not None and can(lvl) and (not up or self._can_fuse_ec(lvl))):
                return [None]
            pairs = self._xpairs(lvl)
            P = as_index_array(pairs, 2)
            if len(pairs) < 2 or P[0, 0] != 0 or not np.array_equal(as_index_array(self._c_points(lvl), 1)[:, 0], P[1:, 0]):
                return [None]
            # every interval between two C-points is one run of F-points, at least one (compared as arrays: 16384 of them at config 3)
            R = as_index_array(self._f_runs(lvl), 2)
            want_len = P[1:, 0] - P[:-1, 0] - 1
            if R.shape[0] != P.shape[0] - 1 or not np.array_equal(R[:, 0], P[:-1, 0] + 1) or not np.array_equal(R[:, 1], want_len) or \
                    (want_len < 1).any():
                return [None]
            # rows of lvl+1 that the down pass must really store for the closing C-point (include/mgrit_hip.h, keep): u only
            # where the coarse level reads it before writing it -- its C-points when it starts with an F-relaxation (always,
            # mgrit.py:270-271), nothing on a coarsest level that forward_solve overwrites from its first point on --, v only
            # when the correction on the way up is not the pass that takes v from the fine C-point
            coarsest = lvl + 1 == self.lvl_max - 1
            jce = P[1:, 1]                                   # coarse slot of the C-point an interval ends on
            c_next = np.asarray(self.index_local_c[lvl + 1], dtype=np.int64)
            if coarsest:
                need_u = np.zeros(jce.size, dtype=bool) if self._coarsest_u_unread() else np.ones(jce.size, dtype=bool)
            else:
                need_u = np.isin(jce, c_next)
            need_v = 0 if (lvl == 0 and self.conv_crit in (0, 2)) else 2
            # ... and where the coarse level's first pass (relax mode FC, _coarse_down) starts its runs from v: its C-points
            v_start = np.isin(jce, c_next) if (not coarsest and self._coarse_down(lvl + 1) is not None) else np.zeros(jce.size, dtype=bool)
            keep = need_u.astype(np.int64) | need_v | (2 * v_start.astype(np.int64))
            # aligned ranks: the last local point travels to the next owner (op 4: u^{l+1}) and its corrected value is staged for
            # op 0 of the way up from u^{l+1} and v^{l+1} (Mgrit._x0): both rows are kept there
            if self.comm_time_size > 1 and self.send_to[lvl + 1] >= 0:
                keep[-1] |= 3
            jcs = P[:-1, 1].copy()
            jcs[0] = -1
            return [IndexArray(np.column_stack((P[:-1, 0], P[1:, 0], jcs, jce, np.arange(len(pairs) - 1), keep)), width=6)]
        got = self._cached(('intervals', lvl, up), build)[0]
        if got is None:
            return None
        return self._cached(('intervals_list', lvl, up), lambda: got)

    def _gen_intervals(self, lvl):
        """[(cstart, cend, cstart_coarse, cend_coarse, res_pos, 3)] of level lvl when its sweeps can run as the general
        whole-level passes (mgrit_hip_gen_down / mgrit_hip_gen_up: any 1-D stepper pair, any of the library's transfers), else
        None: one rank, the library's own sweeps, weight 1, every F-point between two local C-points."""
        def build():
            be = self.backend
            own = all(getattr(type(self), name) is getattr(_library(), name) for name in
                      ("iteration", "f_relax", "c_relax", "fas_residual", "error_correction", "compute_residual", "_exchange",
                       "_ec_f_relax", "_fas_residual_fused", "_relax_f"))
            can = getattr(be, "can_gen_level", None)
            if not (own and self._one_rank_like() and self.weight_c == 1.0 and lvl < self.lvl_max - 1 and self._dry is None and
                    not getattr(self, "_sweep_timing", False) and can is not None and can(lvl) and
                    True):
                return [None]
            P, R = as_index_array(self._xpairs(lvl), 2), as_index_array(self._f_runs(lvl), 2)
            if P.shape[0] < 2 or P[0, 0] != 0 or not np.array_equal(as_index_array(self._c_points(lvl), 1)[:, 0], P[1:, 0]):
                return [None]
            want = np.stack((P[:-1, 0] + 1, P[1:, 0] - P[:-1, 0] - 1), axis=1)
            if R.shape != want.shape or not np.array_equal(R, want) or (want[:, 1] < 1).any():
                return [None]
            jcs = P[:-1, 1].copy()
            jcs[0] = -1
            n_iv = P.shape[0] - 1
            return [IndexArray(np.column_stack((P[:-1, 0], P[1:, 0], jcs, P[1:, 1], np.arange(n_iv), np.full(n_iv, 3))), width=6)]
        got = self._cached(('gen_intervals', lvl), build)[0]
        if got is None:
            return None
        return self._cached(('gen_intervals_list', lvl), lambda: got)

    def _can_fuse_ec(self, lvl):
        return (getattr(self.backend, "can_fuse_ec", None) is not None and self.backend.can_fuse_ec(lvl) and
                type(self).error_correction is _library().error_correction and type(self).f_relax is _library().f_relax and
                type(self).fas_residual is _library().fas_residual)   # the kernel takes v_j from u_c: only the library's FAS sweep guarantees it

