"""Plugin sweep backend: runs the MGRIT sweeps through the user's own Python ``Application.step`` /
``Vector`` / ``GridTransfer`` objects, one call per time point.

This is the path for Applications that carry NO device description (user-defined plugins, Dahlquist): their Phi is
arbitrary Python and cannot run on the GPU. It is selected by application type, never as a fallback for a missing
GPU or library. Semantics follow the reference operation for operation (reference src/pymgrit/core/mgrit.py,
line numbers on each method) -- new Vector objects are bound on every update, operand order is preserved.
"""
import numpy as np


class PluginBackend:
    name = "plugin"

    def __init__(self, mg):
        self.mg = mg

    # -- state (mgrit.py:840-858) -------------------------------------------------------------------
    def create_u_v_g(self, lvl):
        mg = self.mg
        tmpl = mg.problem[lvl].vector_template
        n = len(mg.t[lvl])
        if lvl == 0:
            make = tmpl.clone_rand if mg.random_init_guess else tmpl.clone_zero
            mg.u.append([make() for _ in range(n)])
            mg.v.append(None)
            mg.g.append(None)
        else:
            mg.u.append([tmpl.clone_zero() for _ in range(n)])
            mg.v.append([item.clone_zero() for item in mg.u[lvl]])
            mg.g.append([item.clone_zero() for item in mg.u[lvl]])
        if mg.comm_time_rank == 0:
            mg.u[lvl][0] = mg.problem[lvl].vector_t_start.clone()

    def finalize(self):
        pass

    # -- exchange payloads (mgrit.py:693-713: pack / unpack) ------------------------------------------
    def payload(self, lvl, idx, op=None):
        return self.mg.u[lvl][idx].pack()

    def recv_buffer(self, lvl, idx, op=None):
        return None

    def commit(self, lvl, idx, got, op=None):
        self.mg.u[lvl][idx].unpack(got)

    # -- relaxation (mgrit.py:319-327, 358-368, 472-481) ----------------------------------------------
    def _phi(self, lvl, src, i):
        mg = self.mg
        return mg.step[lvl](u_start=src[i - 1], t_start=mg.t[lvl][i - 1], t_stop=mg.t[lvl][i])

    def relax(self, lvl, runs, mode):
        """mode 'F' (F-intervals), 'C' (weighted C-points) or 'CHAIN' (sequential coarsest-level solve = 'F' arithmetic)"""
        mg = self.mg
        u = mg.u[lvl]
        for start, length in runs:
            for i in range(start, start + length):
                new = self._phi(lvl, u, i) if lvl == 0 else mg.g[lvl][i] + self._phi(lvl, u, i)
                if mode == 'C':
                    new = new * mg.weight_c + u[i] * (1.0 - mg.weight_c)
                u[i] = new

    # -- convergence (mgrit.py:372-413) ---------------------------------------------------------------
    def residual_norms(self, points):
        u = self.mg.u[0]
        return [(self._phi(0, u, i) - u[i]).norm() for i in points]

    # -- AT-MGRIT (core/at_mgrit.py:37-87) ---------------------------------------------------------------
    def at_forward_solve(self, lvl, k):
        """every coarsest-level point from the OLD value k-1 points back, k-1 steps with the FAS right-hand side. Several
        ranks: the (u, g) rows of the level are gathered first (the reference gathers the same rows, one per rank, over its
        black/green communicators and therefore allows one coarsest point per rank only; any distribution works here)."""
        mg = self.mg
        t, tmpl = mg.global_t[lvl], mg.problem[lvl].vector_template
        n = len(t)
        if mg.comm_time_size == 1:
            old, g, owned = [v.clone() for v in mg.u[lvl]], mg.g[lvl], list(range(n))
            local_of = {p: p for p in owned}
        else:
            mine = {int(gi): (mg.u[lvl][int(li)].pack(), mg.g[lvl][int(li)].pack())
                    for gi, li in zip(mg.cpts[lvl], mg.index_local[lvl])}
            rows = {}
            for part in mg.comm_time.allgather_object(mine):
                rows.update(part)

            def vec(payload):
                v = tmpl.clone_zero()
                v.unpack(payload)
                return v
            old = [vec(rows[p][0]) for p in range(n)]
            g = [vec(rows[p][1]) for p in range(n)]
            owned = [int(gi) for gi in mg.cpts[lvl]]
            local_of = {int(gi): int(li) for gi, li in zip(mg.cpts[lvl], mg.index_local[lvl])}
        for p in owned:
            cur = old[max(0, p - k + 1)]
            for i in range(max(1, p - k + 2), p + 1):
                cur = g[i] + mg.step[lvl](u_start=cur, t_start=t[i - 1], t_stop=t[i])
            mg.u[lvl][local_of[p]] = cur

    def residual_begin(self, points):
        return self.residual_norms(points)

    def residual_end(self, handle):
        return handle

    def save_last(self):
        self.mg.save_values_last_iter = [item.clone() for item in self.mg.u[0]]

    # -- C-point snapshots of level 0 (pipelined solve, Mgrit._solve_pipelined) -----------------------
    def snapshot_cpoints(self, slot, points):
        if not hasattr(self, "_snap"):
            self._snap = {}
        self._snap[slot] = [self.mg.u[0][i].clone() for i in points]

    def restore_cpoints(self, slot, points):
        for i, vec in zip(points, self._snap[slot]):
            self.mg.u[0][i] = vec.clone()

    def jump_norms(self, points):
        mg = self.mg
        out = [(mg.u[0][i] - mg.save_values_last_iter[i]).norm() for i in points]
        self.save_last()
        return out

    # -- grid transfer sweeps (mgrit.py:498-500, 520, 524-547, 715-726, 559-563) ----------------------
    def restrict_u(self, lvl, pairs):
        mg = self.mg
        for i, j in pairs:
            mg.u[lvl + 1][j] = mg.restriction[lvl](mg.u[lvl][i])

    def copy_u_to_v(self, lvl):
        self.mg.v[lvl] = [item.clone() for item in self.mg.u[lvl]]

    def fas_rhs(self, lvl, pairs):
        mg = self.mg
        for i, j in pairs:
            fine = self._phi(lvl, mg.u[lvl], i)
            if lvl == 0:
                defect = fine - mg.u[lvl][i]
            else:
                defect = mg.g[lvl][i] - mg.u[lvl][i] + fine
            mg.g[lvl + 1][j] = mg.restriction[lvl](defect) + mg.v[lvl + 1][j] - self._phi(lvl + 1, mg.v[lvl + 1], j)

    def error_correction(self, lvl, pairs):
        mg = self.mg
        for i, j in pairs:
            mg.u[lvl][i] = mg.u[lvl][i] + mg.interpolation[lvl](mg.u[lvl + 1][j] - mg.v[lvl + 1][j])

    def interpolate(self, lvl, pairs):
        mg = self.mg
        for i, j in pairs:
            mg.u[lvl][i] = mg.interpolation[lvl](u=mg.u[lvl + 1][j])

    def sync(self):
        pass
