"""HIP sweep backend: the MGRIT sweeps as hand-written gfx950 kernels behind the C ABI of include/mgrit_hip.h.

State layout: per level one float64 slab ``[n_local_points][ld]`` per array (u, v, g) in HBM (rows in the engine's
lane-blocked order, ``hip_lib.row_permutation``), allocated as torch CUDA tensors (PyTorch is plumbing here: device memory, the stream, torch.distributed); the library only
sees raw device pointers. ``mgrit.u[lvl]`` stays an indexable sequence of Vector objects (lazy host views), which
is what output_fcn callbacks and subclasses of the reference read.

No CPU fallback: constructing this backend without libmgrit_hip.so or without a GPU raises MgritHipError.
"""
import ctypes as C
import os

import numpy as np
import torch

from pymgrit_amd.core import hip_lib
from pymgrit_amd.core.options import options
from pymgrit_amd.core.hip_lib import MgritHipError, check


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _cols(items, k):
    """the k columns of a list of equal-length int tuples as contiguous int32 arrays (one conversion, not k list walks)"""
    if hasattr(items, "columns"):       # layout.IndexArray: the columns without a detour through Python objects
        cols = items.columns()
        if len(cols) == k:
            return cols
    if not len(items):
        return [np.zeros(0, dtype=np.int32) for _ in range(k)]
    arr = np.asarray(items, dtype=np.int32).reshape(len(items), -1)
    return [np.ascontiguousarray(arr[:, c]) for c in range(k)]


def _time_factor(f, t):
    """tau(t_i) of a separable forcing term at every point of a level. The user's callable is written for ONE time (heat_1d.py's
    rhs is called per step); evaluated point by point a level of BASELINE config 3 costs 65537 Python calls (70 ms of a 115 ms
    setup). options.time_factor: 'pointwise' = always the loop (the reference's calls, no question asked); 'auto' (default) =
    the callable is tried on the whole array and its result taken only if
      * it has the right shape,
      * later calls on t[1:] and on the first half of t return the same bits for the same times (no state between calls; an
        elementwise function must not depend on a value's position in the array -- SIMD lane or tail --, on the array's length
        or on its other entries),
      * it agrees bit for bit with the point-by-point calls at EVERY point of a level of <= 1024 points, and at both ends + 64
        sampled points of a longer one;
    anything else -- an exception, a scalar, a different bit anywhere -- falls back to the loop. A callable carrying the
    attribute `elementwise = True` (pymgrit_amd.elementwise(f)) declares the property itself and skips the shifted calls.
    INTEGRATION.md states the contract."""
    t = np.asarray(t, dtype=np.float64)
    mode = options.time_factor
    if t.size > 256 and mode != "pointwise":
        try:
            v = np.asarray(f(t), dtype=np.float64)
            ok = v.shape == t.shape
            if ok and not getattr(f, "elementwise", False):
                h = t.size // 2
                ok = (np.asarray(f(t[1:]), dtype=np.float64).tobytes() == v[1:].tobytes()
                      and np.asarray(f(t[:h]), dtype=np.float64).tobytes() == v[:h].tobytes())
            if ok:
                if t.size <= 1024:
                    idx = range(t.size)
                else:
                    idx = np.unique(np.concatenate(([0, 1, t.size - 2, t.size - 1], np.random.default_rng(t.size).integers(0, t.size, 64))))
                if all(np.float64(f(t[i])).tobytes() == v[i].tobytes() for i in idx):
                    return v
        except Exception:       # noqa: BLE001 -- whatever the callable does with an array is its business: ask it point by point
            pass
    return np.asarray([f(tt) for tt in t], dtype=np.float64)


def _ptr(a):
    return C.c_void_p(a.ctypes.data) if a.size else C.c_void_p(0)


class SlabVectorList:
    """``mgrit.u[lvl]``-compatible view of a device slab: indexing copies ONE row to the host (undoing the engine's
    lane-blocked storage order, include/mgrit_hip.h) and wraps it in the application's Vector type; assignment
    uploads a Vector."""

    def __init__(self, slab, n, template, perm, on_write=None, on_read=None):
        self.slab, self.n, self.template, self.perm, self.on_write, self.on_read = slab, n, template, perm, on_write, on_read

    def __len__(self):
        return self.slab.shape[0]

    def _row(self, i):
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return i

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if self.on_read is not None:
            self.on_read()
        vec = self.template.clone_zero()
        host = self.slab[self._row(int(i))][self.perm].cpu().numpy()
        vec.unpack(host.reshape(np.shape(vec.pack())).copy())   # pack()/unpack(): the Vector's own flat payload form
        return vec

    def __setitem__(self, i, vec):
        if self.on_write is not None:
            self.on_write()
        vals = np.ascontiguousarray(np.asarray(vec.pack(), dtype=np.float64)).ravel()
        # through pinned memory, enqueued behind whatever the stream still holds (the slab's zero fill, in the constructor): a
        # pageable copy would make the host wait for all of it
        host = torch.empty(vals.size, dtype=torch.float64, pin_memory=True)
        host.numpy()[:] = vals
        self.slab[self._row(int(i))][self.perm] = host.to(self.slab.device, non_blocking=True)

    def __iter__(self):
        return (self[k] for k in range(len(self)))


class HipBackend:
    name = "hip"

    def __init__(self, mg):
        self.mg = mg
        self.lib = hip_lib.load()
        if not torch.cuda.is_available() or self.lib.mgrit_hip_device_count() < 1:
            raise MgritHipError("no MI355X/HIP device visible: device applications have no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.stream = torch.cuda.current_stream(self.device)
        self.h = C.c_void_p()
        # one spare level index: the work level of AT-MGRIT on several ranks (at_forward_solve), described on first use
        check(self.lib.mgrit_hip_create(C.byref(self.h), mg.lvl_max + 1, C.c_void_p(self.stream.cuda_stream)))
        self.desc = [p.device_stepper() for p in mg.problem]
        self.n = [int(d["n"]) for d in self.desc]
        # 1-D steppers: lane-blocked rows; Heat2D: the nx x ny grid in natural row-major order
        # two-point states: the row is [first | second], each half lane-blocked like a 1-D row of n values
        self.ld = [((n + 15) // 16) * 16 if d["kind"] == "heat2d" else
                   (2 if d["kind"] == "heat1d_2pts" else 1) * hip_lib.row_stride(n) for n, d in zip(self.n, self.desc)]
        self.perm, made = [], {}
        for n, d in zip(self.n, self.desc):     # (levels of one spatial size share the tensor: one upload, in front of the slabs' zero fills)
            key = (n, d["kind"])
            if key not in made:
                if d["kind"] == "heat2d":
                    perm = np.arange(n)
                elif d["kind"] == "heat1d_2pts":
                    half = hip_lib.row_permutation(n)
                    perm = np.concatenate((half, half + hip_lib.row_stride(n)))
                else:
                    perm = hip_lib.row_permutation(n)
                made[key] = torch.from_numpy(np.asarray(perm, dtype=np.int64)).to(self.device)
            self.perm.append(made[key])
        self._U, self.V, self.G, self.FB = [], [], [], {}
        self._f_stale = 0         # level-0 F-points: 0 all in place; 1 all but the last of every interval await materialise();
                                  # 2 as 1, and the last one's row holds Phi of it (the next C-relaxation's value, cf_fas pre)
        self._cycle_pre = False   # the state at the start of the running cycle was 2 (begin_cycle)
        self._runs, self._pairs = {}, {}
        self._described = [False] * mg.lvl_max
        self.chain_state, self.chain_handover, self._handover = {}, {}, {}
        self.prev = None
        self._sumsq = None
        self._cur_stream, self._chain_stream = self.stream, None

    def __del__(self):
        try:
            if self.h:
                self.lib.mgrit_hip_destroy(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass

    # -- state (mgrit.py:840-858) -------------------------------------------------------------------
    def _describe_heat1d(self, engine_lvl, d, t_local, n, ld):
        n_pts = t_local.size
        s = np.ascontiguousarray(np.asarray(d.get("forcing_space", np.zeros((0, n))), dtype=np.float64).reshape(-1, n))
        K = s.shape[0]
        tau = np.zeros((K, n_pts))
        for k in range(K):
            tau[k] = _time_factor(d["forcing_time"][k], t_local)
        tau = np.ascontiguousarray(tau)
        check(self.lib.mgrit_hip_level_heat1d(self.h, engine_lvl, n_pts, _ptr(t_local), n, ld, float(d["fac"]), K,
                                              _ptr(s), _ptr(tau)))
        if d.get("forcing_rows") is not None:
            # general forcing: rows rhs(x, t_i)*dt_i for the local points, uploaded in blocks (a level of BASELINE config 3
            # is 8.6 GB of them), in the engine's row storage order
            rows = torch.zeros((max(n_pts, 1), ld), dtype=torch.float64, device=self.device)
            perm = hip_lib.row_permutation(n)
            for a in range(1, n_pts, 1024):
                z = min(n_pts, a + 1024)
                host = np.zeros((z - a, ld))
                for i in range(a, z):
                    host[i - a, perm] = d["forcing_rows"](float(t_local[i - 1]), float(t_local[i]))
                rows[a:z].copy_(torch.from_numpy(host))
            self.FB[engine_lvl] = rows
            check(self.lib.mgrit_hip_level_forcing_rows(self.h, engine_lvl, C.c_void_p(rows.data_ptr())))

    def create_u_v_g(self, lvl):
        mg = self.mg
        d, n, ld = self.desc[lvl], self.n[lvl], self.ld[lvl]
        t_local = np.ascontiguousarray(np.asarray(mg.t[lvl], dtype=np.float64))
        n_pts = t_local.size
        if d["kind"] == "heat1d":
            self._describe_heat1d(lvl, d, t_local, n, ld)
        elif d["kind"] == "heat1d_2pts":
            s = np.ascontiguousarray(np.asarray(d.get("forcing_space", np.zeros((0, n))), dtype=np.float64).reshape(-1, n))
            K = s.shape[0]
            tau, tau2 = np.zeros((K, n_pts)), np.zeros((K, n_pts))
            for k in range(K):
                tau[k] = _time_factor(d["forcing_time"][k], t_local)
                tau2[k] = _time_factor(d["forcing_time"][k], t_local + d["dtau"])      # (t + dtau elementwise: the same sums)
            check(self.lib.mgrit_hip_level_heat1d_2pts(self.h, lvl, n_pts, _ptr(t_local), n, ld, float(d["fac"]),
                                                       float(d["dtau"]), int(d["order"]), K, _ptr(s),
                                                       _ptr(np.ascontiguousarray(tau)), _ptr(np.ascontiguousarray(tau2))))
        elif d["kind"] == "advection1d":
            check(self.lib.mgrit_hip_level_advection1d(self.h, lvl, n_pts, _ptr(t_local), n, ld, float(d["fac"])))
        elif d["kind"] == "heat2d":
            nx, ny = int(d["nx"]), int(d["ny"])
            S = np.ascontiguousarray(np.asarray(d["forcing_space"], dtype=np.float64).reshape(-1, (nx - 2) * (ny - 2)))
            K = S.shape[0]
            tau = np.zeros((K, n_pts))
            for k in range(K):
                tau[k] = _time_factor(d["forcing_time"][k], t_local)
            tau = np.ascontiguousarray(tau)
            bc = np.ascontiguousarray(np.asarray(d["bc"], dtype=np.float64).ravel())
            check(self.lib.mgrit_hip_level_heat2d(self.h, lvl, n_pts, _ptr(t_local), nx, ny, ld, float(d["fx"]), float(d["fy"]),
                                                  float(d["theta"]), _ptr(bc), K, _ptr(S), _ptr(tau)))
            if d.get("forcing_rows") is not None:
                # general forcing: rhs(x, y, t_i) on the padded interior for every local point (the theta-scheme weighs the two
                # ends of a step itself), uploaded in blocks
                Mi, Mj = C.c_int(0), C.c_int(0)
                check(self.lib.mgrit_hip_heat2d_padded(self.h, lvl, C.byref(Mi), C.byref(Mj)))
                Mi, Mj = Mi.value, Mj.value
                rows = torch.zeros((max(n_pts, 1), Mi * Mj), dtype=torch.float64, device=self.device)
                for a in range(0, n_pts, 256):
                    z = min(n_pts, a + 256)
                    host = np.zeros((z - a, Mi, Mj))
                    for i in range(a, z):
                        host[i - a, :nx - 2, :ny - 2] = d["forcing_rows"](float(t_local[i]))
                    rows[a:z].copy_(torch.from_numpy(host.reshape(z - a, Mi * Mj)))
                self.FB[lvl] = rows
                check(self.lib.mgrit_hip_level_heat2d_forcing_rows(self.h, lvl, C.c_void_p(rows.data_ptr())))
        else:
            raise MgritHipError(f"unknown device stepper kind {d['kind']!r}")
        u = torch.zeros((n_pts, ld), dtype=torch.float64, device=self.device)
        tmpl = mg.problem[lvl].vector_template
        if lvl == 0 and mg.random_init_guess and n_pts:
            host = np.zeros((n_pts, ld))
            perm = self.perm[lvl].cpu().numpy()
            for i in range(n_pts):  # clone_rand per time point, in time order (heat_1d.py:88-96)
                host[i, perm] = np.asarray(tmpl.clone_rand().pack(), dtype=np.float64).ravel()
            u.copy_(torch.from_numpy(host))
        v = g = None
        if lvl > 0:
            v = torch.zeros_like(u)
            g = torch.zeros_like(u)
        self._U.append(u), self.V.append(v), self.G.append(g)
        check(self.lib.mgrit_hip_level_bind(self.h, lvl, C.c_void_p(u.data_ptr()),
                                            C.c_void_p(v.data_ptr() if v is not None else 0),
                                            C.c_void_p(g.data_ptr() if g is not None else 0)))
        # Overlapped chain: every owner of the level must take the same form of the forward solve and exchange
        # the same hand-over (op 5: the last point + the chain's running state), so the decision comes from the level's GLOBAL
        # time grid -- the engine only sees this rank's points
        gt = np.asarray(mg.global_t[lvl], dtype=np.float64)
        dts = np.diff(gt)
        n_terms = np.asarray(d.get("forcing_space", np.zeros((0, n)))).reshape(-1, n).shape[0] if d["kind"] == "heat1d" else 0
        wide = (lvl > 0 and d["kind"] == "heat1d" and 1024 < n <= hip_lib.MAX_N and n_terms <= 1 and d.get("forcing_rows") is None and dts.size > 0
                and bool(np.all(dts.view(np.int64) == dts.view(np.int64)[0])) and not options.chain_plain)
        check(self.lib.mgrit_hip_chain_enable(self.h, lvl, int(wide)))
        slen = C.c_int(0)
        check(self.lib.mgrit_hip_chain_state_len(self.h, lvl, C.byref(slen)))
        self.chain_state[lvl] = None
        self.chain_handover[lvl] = ld + 64 if wide else 0     # doubles behind the point in an op-5 message (0: the point alone)
        if slen.value:
            assert slen.value == self.chain_handover[lvl]
            self.chain_state[lvl] = torch.zeros(slen.value, dtype=torch.float64, device=self.device)
            check(self.lib.mgrit_hip_chain_bind(self.h, lvl, C.c_void_p(self.chain_state[lvl].data_ptr())))
        self._config_block_solve(lvl, d, n, gt)
        mg.u.append(SlabVectorList(u, n, tmpl, self.perm[lvl], on_write=self._before_write if lvl == 0 else self._forget_residual,
                                   on_read=self.materialise if lvl == 0 else None))
        mg.v.append(SlabVectorList(v, n, tmpl, self.perm[lvl]) if v is not None else None)
        mg.g.append(SlabVectorList(g, n, tmpl, self.perm[lvl]) if g is not None else None)
        if mg.comm_time_rank == 0 and n_pts:
            mg.u[lvl][0] = mg.problem[lvl].vector_t_start
        self._described[lvl] = True

    def _config_block_solve(self, lvl, d, n, gt):
        """Time-parallel forward solve (csrc/mgrit_hip_blk.inc): the rule is applied to the level's GLOBAL time
        grid, so every owner of a sharded level takes the same form. block_r[lvl] = sine modes in use (0: step by step).
        Several ranks: every rank's share of the level has to be whole blocks (its ghost point a multiple of BLOCK_K steps from
        the start, every rank at least one block) -- decided from the global layout, the same on every rank; the op-5 message
        then carries the BLOCK_RMAX mode amplitudes behind the point instead of a chain state."""
        from pymgrit_amd.core.layout import compute_layout
        mg = self.mg
        self.block_r = getattr(self, "block_r", {})
        self.block_sharded = getattr(self, "block_sharded", {})
        self.block_uh = getattr(self, "block_uh", {})
        r = C.c_int(0)
        if d["kind"] == "heat2d":
            # Heat2D (backward Euler or Crank-Nicolson, one rank): the engine's own rule (level > 0, theta = 1 or 1/2, >= 64 steps), full sine spectrum
            want = (lvl > 0 and lvl == mg.lvl_max - 1 and options.coarse_solve != "sequential" and mg.comm_time_size == 1 and
                    getattr(type(mg).forward_solve, "__qualname__", "") == "Mgrit.forward_solve")
            check(self.lib.mgrit_hip_block_solve_config(self.h, lvl, -1 if want else 0, 1, 0, None, None))
            check(self.lib.mgrit_hip_block_solve_state(self.h, lvl, C.byref(r)))
            self.block_r[lvl], self.block_sharded[lvl] = r.value, False
            return
        kinds = {"heat1d": hip_lib.STEPPER_HEAT1D, "advection1d": hip_lib.STEPPER_ADVECTION1D}
        if (lvl > 0 and lvl == mg.lvl_max - 1 and d["kind"] in kinds and n <= hip_lib.MAX_N and options.coarse_solve != "sequential"
                and getattr(type(mg).forward_solve, "__qualname__", "") == "Mgrit.forward_solve" and mg.global_conv_crit):
            check(self.lib.mgrit_hip_block_solve_rank(kinds[d["kind"]], n, float(d["fac"]), gt.size, _ptr(np.ascontiguousarray(gt)),
                                                      C.byref(r)))
        size, K = mg.comm_time_size, hip_lib.BLOCK_K
        first_real, successor = 1, 0
        uh_in = uh_out = None
        if r.value and size > 1:
            ok = True
            for p in range(size):
                lay = compute_layout(mg.global_t, lvl, p, size)
                n_owned = len(lay.index_local)
                a = lay.first_owned
                z = a + n_owned - 1
                # first rank: owns point 0; the others: ghost a-1 on a block border; every rank ends on a block border or on
                # the last point, and has at least one block of its own
                ok = ok and n_owned > 0 and (a == 0 if p == 0 else (a - 1) % K == 0) and (z == gt.size - 1 or z % K == 0) and \
                    (z - max(a - 1, 0)) >= K
            if not ok:
                r = C.c_int(0)
            else:
                first_real = int(mg.get_from[lvl] == -99)
                successor = int(mg.send_to[lvl] != -99)
                hlen = hip_lib.BLOCK_RMAX if d["kind"] == "heat1d" else 2 * n     # sine-mode amplitudes / n complex Fourier amplitudes
                uh_in = torch.zeros(hlen, dtype=torch.float64, device=self.device)
                uh_out = torch.zeros(hlen, dtype=torch.float64, device=self.device)
                self.block_uh[lvl] = (uh_in, uh_out)
        check(self.lib.mgrit_hip_block_solve_config(self.h, lvl, r.value, first_real, successor,
                                                    C.c_void_p(uh_in.data_ptr() if uh_in is not None else 0),
                                                    C.c_void_p(uh_out.data_ptr() if uh_out is not None else 0)))
        self.block_r[lvl] = r.value
        self.block_sharded[lvl] = bool(r.value and size > 1)
        if self.block_sharded[lvl]:
            self.chain_handover[lvl] = int(uh_in.numel())      # doubles behind the point in the op-5 message
            # no chain on this level any more: the engine must not keep the address of a state tensor that goes back to the allocator
            check(self.lib.mgrit_hip_chain_enable(self.h, lvl, 0))
            check(self.lib.mgrit_hip_chain_bind(self.h, lvl, None))
            self.chain_state[lvl] = None

    def block_solve(self, lvl, phases):
        """phases of the time-parallel forward solve on a rank of a sharded level (mgrit_hip_block_solve): 1 = first pass,
        2 = recurrence over the blocks (+ the corrected last point when a successor waits for it), 4 = corrections + second pass"""
        self._settle(lvl)
        check(self.lib.mgrit_hip_block_solve(self.h, lvl, int(phases)))

    def block_solve_form(self, lvl):
        """how the level's forward solve runs (mgrit_hip_block_solve_form): 0 step by step, 1 a launch or more per phase of the
        time-parallel form, 2 the whole time-parallel solve in one launch (small Heat1D levels)"""
        form = C.c_int(0)
        check(self.lib.mgrit_hip_block_solve_form(self.h, lvl, C.byref(form)))
        return form.value

    def finalize(self):
        """after every level is described: register the spatial transfers"""
        mg = self.mg
        for lvl in range(mg.lvl_max - 1):
            tr = mg.transfer_objects[lvl]
            kind = int(tr.device_transfer()) if self._device_transfer(lvl) else hip_lib.TRANSFER_CALLER
            if kind == hip_lib.TRANSFER_CALLER and self.desc[lvl]["kind"] not in ("heat1d", "advection1d", "heat2d"):
                raise MgritHipError(f"transfer {type(tr).__name__} is applied through its Python methods, which the "
                                    f"{self.desc[lvl]['kind']} levels do not support (they take GridTransferCopy)")
            check(self.lib.mgrit_hip_level_transfer(self.h, lvl, kind))
        # ghost rows as stream operations of the engine (mgrit_hip_exchange) when the time communicator offers links
        # (RcclTimeComm, LoopbackComm): opening them is collective -- every rank is in its constructor at this point
        self.device_links = False
        comm = mg.comm_time
        if mg.comm_time_size > 1 and getattr(comm, "device_exchange", False) and \
                options.exchange != "torch":
            from pymgrit_amd.core.comm import links_needed
            agreed = getattr(comm, "open_links_agreed", None)
            self.link_error = agreed(self, links_needed(mg)) if agreed is not None else comm.open_links(self, links_needed(mg))
            if self.link_error is None:
                self.device_links = True
            else:   # every rank has the same word: the ghost rows travel through torch.distributed instead (core/comm.py)
                import warnings
                warnings.warn(f"RCCL links could not be opened, exchange through torch.distributed: {self.link_error}")

    def _xlink(self, peer, direction, ch, ordinal):
        """((link handle, slot), done): a single message shakes hands with its peer now (LoopbackComm: the receiver enqueues
        behind the sender; RCCL links: nothing to do); the ordinal-th message of a planned cycle has its slot fixed, the
        hand-shake was the cycle's (plan_run)"""
        comm = self.mg.comm_time
        if ordinal is not None:
            return comm.cycle_slot(self, peer, direction, ch, ordinal), None
        if direction == 'send':
            return comm.send_begin(self, peer, ch), lambda: comm.send_end(self, peer, ch)
        return comm.recv_begin(self, peer, ch), lambda: comm.recv_end(self, peer, ch)

    def exchange(self, lvl, op, send_idx=None, dest=None, recv_idx=None, src=None, raw=False, ordinals=(None, None)):
        """one exchange point (reference mgrit.py:693-713) as stream operations of the engine: mgrit_hip_exchange sends row
        send_idx of u^lvl to rank dest and receives row recv_idx from rank src (ncclSend / ncclRecv on an RCCL link); the
        hand-over of forward_solve (op 5) takes the chain's running state along. raw: the caller knows what
        the row holds (C-point storage: no materialise() in front of the send)"""
        from pymgrit_amd.core.comm import CH_CHAIN, CH_SWEEP
        ch = CH_CHAIN if op == 5 else CH_SWEEP
        if lvl == 0 and send_idx is not None and not raw:
            self.materialise()
        sl = ss = rl = rs = -1
        done = []
        if send_idx is not None:
            (sl, ss), fin = self._xlink(dest, 'send', ch, ordinals[0])
            done.append(fin)
        if recv_idx is not None:
            (rl, rs), fin = self._xlink(src, 'recv', ch, ordinals[1])
            done.append(fin)
            self._residual_cache = None
        check(self.lib.mgrit_hip_exchange(self.h, lvl, op, sl, -1 if send_idx is None else int(send_idx), ss, rl,
                                          -1 if recv_idx is None else int(recv_idx), rs,
                                          int(self.chain_handover.get(lvl, 0) or 0) if op == 5 else 0))
        for fin in done:
            if fin is not None:
                fin()

    def exchange_staged(self, lvl, op, pair, dest=None, recv_idx=None, src=None, ordinals=(None, None)):
        """the same exchange point with the CORRECTED value of the C-point `pair` = [(fine slot, coarse slot)] as the row sent
        (mgrit_hip_error_correction_to into a staging row): an aligned rank sends its last C-point before the whole-level pass
        that corrects it in place has run (Mgrit._x0)"""
        from pymgrit_amd.core.comm import CH_SWEEP
        ld = self.ld[lvl]
        if not hasattr(self, "_stage"):
            self._stage = {}
        if lvl not in self._stage:
            self._stage[lvl] = torch.zeros(ld, dtype=torch.float64, device=self.device)
        stage = self._stage[lvl]
        check(self.lib.mgrit_hip_error_correction_to(self.h, lvl, self._pair_id(lvl, pair), C.c_void_p(stage.data_ptr()), ld))
        (sl, ss), fin = self._xlink(dest, 'send', CH_SWEEP, ordinals[0])
        check(self.lib.mgrit_hip_send(self.h, sl, ss, C.c_void_p(stage.data_ptr()), ld))
        if fin is not None:
            fin()
        if recv_idx is not None:
            (rl, rs), fin = self._xlink(src, 'recv', CH_SWEEP, ordinals[1])
            self._residual_cache = None
            check(self.lib.mgrit_hip_recv(self.h, rl, rs, C.c_void_p(self._U[lvl][int(recv_idx)].data_ptr()), ld))
            if fin is not None:
                fin()

    def _host_transfers(self):
        return any(not self._device_transfer(lvl) for lvl in range(self.mg.lvl_max - 1))

    def _device_transfer(self, lvl):
        """True when the transfer between lvl and lvl+1 is one of the library's own (its kernels apply it); False for a user's
        GridTransfer (reference core/grid_transfer.py:31-55) or a subclass that overrides restriction / interpolation: those
        run through their Python methods, row by row on the host, while every Phi stays on the device"""
        tr = self.mg.transfer_objects[lvl]
        lib_method = type(self.mg)._library_method
        return hasattr(tr, "device_transfer") and lib_method(tr, "restriction") and lib_method(tr, "interpolation")

    def _before_write(self):
        """mgrit.u[0][i] = vec under C-point storage: the F-points the last cycle left out are put in place FIRST -- the write
        replaces one row of a complete solution, as it does in the reference; rebuilt afterwards they would overwrite a written
        F-point, or follow a written C-point"""
        self.materialise()
        self._forget_residual()

    def _forget_residual(self):
        self._residual_cache = None
        self._write_gen = getattr(self, "_write_gen", 0) + 1      # a Vector has been written into a slab from outside

    def write_generation(self):
        """changes whenever slab contents have been written from outside the sweeps (mgrit.u[lvl][i] = vec, set_natural):
        Mgrit._head re-injects the first time point then"""
        return getattr(self, "_write_gen", 0)

    @property
    def U(self):
        """the state slabs, level 0 with every F-point in place (tests, post-processing)"""
        self.materialise()
        return self._U

    def materialise(self):
        """C-point storage (include/mgrit_hip.h, mgrit_hip_ec_relax_res): in the steady state of a solve the way up stores, of
        every level-0 interval's F-points, only the last one. Whoever wants to SEE the solution -- the end of Mgrit.solve(),
        output_fcn, mgrit.u[0][i], natural(), the U slabs -- gets the others rebuilt here by one F-relaxation from the
        C-points: the same Phi on the same values, so bit for bit what an every-point store would have left."""
        self._cycle_pre = False     # whatever the cycle's down pass was promised (begin_cycle), the rows are plain again
        if self._f_stale:
            self._f_stale = 0
            cache = self._residual_cache          # an F-relaxation that rewrites identical values leaves the residual valid
            self.relax(0, self.mg._f_runs(0), 'F')
            self._residual_cache = cache

    def _settle(self, lvl):
        """before any sweep other than the two whole-level passes reads level 0: state 2 (the last F-point's row holds Phi of
        it) is something only cf_fas understands"""
        if lvl == 0 and self._f_stale == 2:
            self.materialise()

    def begin_cycle(self):
        """start of Mgrit.iteration(lvl=0): the down pass of THIS cycle reads the rows the cycle before left, whatever the
        cycle's own up pass does to later blocks in the meantime (planned cycle)"""
        self._cycle_pre = self._f_stale == 2
        self._res_open = None       # a pre-filled residual slot nobody asked for belongs to the cycle before

    def natural(self, which, lvl):
        """host copy of a whole slab in natural x order, shape [n_local_points][n] (tests / post-processing)"""
        if which == "u" and lvl == 0:
            self.materialise()
        slab = {"u": self._U, "v": self.V, "g": self.G}[which][lvl]
        return slab[:, self.perm[lvl]].cpu().numpy()

    def set_natural(self, which, lvl, values):
        """upload a [n_local_points][n] host array given in natural x order"""
        self._residual_cache = None
        self._write_gen = getattr(self, "_write_gen", 0) + 1
        if which == "u" and lvl == 0:
            self._f_stale = 0      # every row replaced: nothing of the last cycle's C-point storage is left to rebuild
        slab = {"u": self._U, "v": self.V, "g": self.G}[which][lvl]
        slab.zero_()
        slab[:, self.perm[lvl]] = torch.from_numpy(np.ascontiguousarray(values, dtype=np.float64)).to(slab.device)

    # -- exchange payloads: slab rows travel in place over RCCL ---------------------------------------
    def payload(self, lvl, idx, op=None):
        if lvl == 0:
            self.materialise()
        if op == 5 and self.block_sharded.get(lvl):    # time-parallel solve: the point and the mode amplitudes behind it
            return torch.cat((self._U[lvl][idx], self.block_uh[lvl][1]))
        if op == 5 and self.chain_handover.get(lvl):   # forward-solve hand-over: the point and the chain's running state
            state = self.chain_state[lvl]
            if state is None:   # this rank has no step of its own on the level (it owns the first point only): a fresh start
                state = torch.cat((self._U[lvl][idx], torch.zeros(64, dtype=torch.float64, device=self.device)))
            return torch.cat((self._U[lvl][idx], state))
        return self._U[lvl][idx]

    def recv_buffer(self, lvl, idx, op=None):
        if op == 5 and self.chain_handover.get(lvl):
            if lvl not in self._handover:
                self._handover[lvl] = torch.empty(self._U[lvl].shape[1] + self.chain_handover[lvl], dtype=torch.float64,
                                                  device=self.device)
            return self._handover[lvl]
        return self._U[lvl][idx]

    def commit(self, lvl, idx, got, op=None):
        self._residual_cache = None
        if op == 5 and self.block_sharded.get(lvl):
            ld = self._U[lvl].shape[1]
            self._U[lvl][idx].copy_(got[:ld])
            self.block_uh[lvl][0].copy_(got[ld:])
        elif op == 5 and self.chain_handover.get(lvl):
            ld = self._U[lvl].shape[1]
            self._U[lvl][idx].copy_(got[:ld])
            if self.chain_state[lvl] is not None:
                self.chain_state[lvl].copy_(got[ld:])
                check(self.lib.mgrit_hip_chain_resume(self.h, lvl, 1))   # the next forward solve continues the sender's chain

    # -- helpers ---------------------------------------------------------------------------------------
    def _handle(self, store, lvl, items, tag, create):
        """device-side list handle: cached on the list object itself when it can carry attributes (Mgrit's IndexList),
        else in a dict keyed by content"""
        attr = f"_hip_{tag}_{lvl}_{id(self)}"
        hid = getattr(items, attr, None)
        if hid is None:
            key = None
            if not hasattr(items, "__dict__"):
                key = (lvl, tuple(items))
                hid = store.get(key)
            if hid is None:
                hid = create()
                if key is None:
                    setattr(items, attr, hid)
                else:
                    store[key] = hid
        return hid

    def _run_id(self, lvl, runs):
        def create():
            rid = C.c_int(-1)
            st, ln = _cols(runs, 2)
            check(self.lib.mgrit_hip_runs_create(self.h, lvl, len(runs), _ptr(st), _ptr(ln), C.byref(rid)))
            return rid.value
        return self._handle(self._runs, lvl, runs, "runs", create)

    def _point_run_id(self, lvl, points):
        def create():
            rid = C.c_int(-1)
            st, ln = _cols(points, 1)[0], np.ones(len(points), dtype=np.int32)
            check(self.lib.mgrit_hip_runs_create(self.h, lvl, len(points), _ptr(st), _ptr(ln), C.byref(rid)))
            return rid.value
        return self._handle(self._runs, lvl, points, "pts", create)

    def _pair_id(self, lvl, pairs):
        def create():
            pid = C.c_int(-1)
            fi, co = _cols(pairs, 2)
            check(self.lib.mgrit_hip_pairs_create(self.h, lvl, len(pairs), _ptr(fi), _ptr(co), C.byref(pid)))
            return pid.value
        return self._handle(self._pairs, lvl, pairs, "pairs", create)

    def _sumsq_buf(self, n):
        if self._sumsq is None or self._sumsq.numel() < n:
            self._sumsq = torch.zeros(max(n, 1), dtype=torch.float64, device=self.device)
        return self._sumsq

    # -- sweeps ----------------------------------------------------------------------------------------
    def _before_c_write(self, lvl):
        """a sweep that WRITES level-0 C-points on its own (a C-relaxation, an error correction, an interpolation called by
        hand) while the last cycle's F-points still await materialise(): they are a function of the C-points as they ARE, so
        they are put in place first -- rebuilt afterwards they would follow the new C-points, which is not what the reference's
        state holds (found by tests/test_hip_state_fuzz.py). Mgrit.iteration's own C-relaxations are followed by an F-relaxation
        that rewrites every F-point anyway (f_relax_follows)."""
        if lvl == 0 and self._f_stale and not getattr(self, "f_relax_follows", False):
            self.materialise()

    def relax(self, lvl, runs, mode):
        self._settle(lvl)
        if mode == 'C':
            self._before_c_write(lvl)
        if lvl == 0:
            self._residual_cache = None
        if not runs:
            return
        code = {'F': hip_lib.RELAX_F, 'C': hip_lib.RELAX_C, 'CHAIN': hip_lib.RELAX_CHAIN, 'FC': hip_lib.RELAX_FC}[mode]
        check(self.lib.mgrit_hip_relax(self.h, lvl, self._run_id(lvl, runs), code, float(self.mg.weight_c)))

    def relax_chain_part(self, lvl, runs, resume):
        """one block of the coarsest-level solve (cycle_plan.py); resume: it continues the chain of the block before it from
        the running state that chain left in the hand-over buffer, exactly as a chain that continues on the
        next rank does"""
        if resume:
            check(self.lib.mgrit_hip_chain_resume(self.h, lvl, 1))
        self.relax(lvl, runs, 'CHAIN')

    # -- planned cycle (cycle_plan.py): sweeps on the engine's stream, chain parts on a second stream -------------------
    def plan_blocks(self):
        """how many blocks of time points a planned cycle uses by default (1 = program order): the overlap pays when the
        coarsest-level solve is long; Heat2D / two-point levels keep the program order"""
        if any(d["kind"] not in ("heat1d", "advection1d", "heat2d") for d in self.desc) or self.mg.lvl_max < 2 or self._host_transfers():
            return 1
        n_c = len(self.mg.t[-1])
        if any(d["kind"] == "heat2d" for d in self.desc) and self.block_r.get(self.mg.lvl_max - 1):
            return 1     # time-parallel coarsest-level solve: batches of its own, nothing sequential to overlap
        if any(d["kind"] == "heat2d" for d in self.desc):
            # the coarsest-level solve (six launches per step) beside the batched sweeps of other blocks, each on CUs of its own
            # (_masked_streams: one XCD of 32 CUs for the solve). Measured on config 4 (2049 coarsest points), ms per cycle: one
            # block 619; without the CU partition two 593, four 578, eight 574; with it eight 533, twelve 523, sixteen 520
            # (16 CUs for the solve: 676-696, 24: 620, 40: 539-550, 48: 540-548, 64: 552-569)
            # ... and since the sweeps apply their own arithmetic inside the last transform (h2d_inv_kernel<true>): sixteen 506,
            # twenty-four 501, thirty-two 501 (40 CUs for the solve: 517-520, 64: 528-530)
            return int(options.plan_blocks_heat2d if options.plan_blocks_heat2d is not None else max(1, min(24, n_c // 85)))
        # measured on config 3 (4097 coarsest points, round 2): 4 blocks 12.2 ms, 5 11.3, 6 10.8, 7 11.9, 8 13.3 -- more blocks
        # shorten the fill and drain of the block pipeline, fewer keep the launches large (a level-0 pass of one block is
        # 16384 / blocks / 4 chunks for 240 workgroups: with 6 blocks 2.8 rounds, with 8 blocks 2.1)
        # ... and only where the coarsest solve is long against the sweeps: Heat1D states wider than one group (config 3).
        # Measured on config 5 (advection_1d, 4 levels, 2 groups per state): one block 26.3 ms per F-cycle, two 30.1, six 32.1
        # (round 2); with the general whole-level passes (round 3) one block 16.3, two 19.0, four 18.9, six 19.4: a chain part
        # that runs beside sweeps shares its CU with their workgroups and finds its rows of g in HBM instead of the Infinity
        # Cache (1.2-1.7 us per step instead of 0.89), which costs more than the overlap of an F-cycle's few sweeps hides.
        if any(d["kind"] != "heat1d" for d in self.desc) or max(self.n) <= 1024:
            return 1
        if self.block_r.get(self.mg.lvl_max - 1):
            return 1     # the coarsest-level solve is time-parallel: nothing sequential is left to run beside sweeps
        if self.mg.comm_time_size > 1:
            # a rank of a sharded run (bench.py --emulate-rank, config 3): its share of the coarsest level is short, and a launch
            # over a fraction of a rank's intervals costs nearly what the launch over all of them costs (a few rounds of
            # workgroups, each a handful of sequential Phi) -- measured per cycle of one rank: 8 ranks (513 coarsest points) one
            # block 1.71 ms, two 1.77-2.07, three 1.97-2.56; 4 ranks (1025) one 3.3, two 2.92, three 2.93-3.07; 2 ranks (2049)
            # two 6.25, three 5.9-6.2, four 5.27
            return int(max(1, min(4, n_c // 512)))
        return int(max(1, min(8, n_c // 680)))

    def plan_allowed(self):
        """planned cycles run sweeps and chain parts on two streams at once: the 1-D one-point steppers and Heat2D (whose
        coarsest-level chain has work buffers of its own since round 3: H2DHost::Wc0); the two-point kernels share per-level work
        buffers between launches"""
        return all(d["kind"] in ("heat1d", "advection1d", "heat2d") for d in self.desc)

    def plan_single_block(self):
        """a cycle that is not cut into blocks may still be planned (as one block, program order) and replayed as one hipGraph
        (plan_run) -- on request only (options.plan_graph = "1" / "require"): measured in round 4 on configs 2, 3 and 5, the
        replay of a one-block cycle is no faster than its launches issued one by one (0.266 against 0.244 ms, 6.78 against
        6.71 ms, 5.18 against 5.17 ms), and recording the plan costs a solve 27 ms of host time"""
        return (options.plan_graph in ("1", "require") and
                all(d["kind"] in ("heat1d", "advection1d") for d in self.desc) and self.mg.lvl_max >= 2 and
                not self._host_transfers())      # host round trips cannot be part of a captured graph

    def _use_stream(self, stream):
        if stream is not self._cur_stream:
            check(self.lib.mgrit_hip_set_stream(self.h, C.c_void_p(stream.cuda_stream)))
            self._cur_stream = stream

    def plan_run(self, plan):
        """Replay a planned cycle. The first two executions issue the launches one by one (they also create the device-side
        index lists); from the third on the whole cycle -- both streams, with the events between them -- is ONE hipGraph
        captured from exactly those launches and replayed with a single call: a cycle of a small hierarchy is a dozen kernels
        of 5-20 us each, and without the graph their launch cost, not their run time, is what a cycle takes.
        PYMGRIT_AMD_PLAN_GRAPH=0 keeps the launch-by-launch form."""
        graph_ok = options.plan_graph != "0" and not getattr(self, "_timing_on", False)
        if any(d["kind"] == "heat2d" for d in self.desc):
            graph_ok = False     # (six launches per step of the coarsest-level solve: tens of thousands of nodes per cycle)
        comm = self.mg.comm_time if getattr(plan, "sends", None) or getattr(plan, "recvs", None) else None
        if comm is not None and plan.n_blocks > 2 and options.plan_graph != "1":
            # a rank's cycle of three and more blocks runs faster launch by launch than as one graph (measured, one rank of two
            # on config 3, four blocks: 5.27 against 6.13 ms; with two blocks the graph wins: 2.92 against 3.33 on one rank of four)
            graph_ok = False
        if comm is not None:     # several ranks: the cycle's messages shake hands once (LoopbackComm; RCCL links match by order)
            comm.cycle_begin(self, plan.sends, plan.recvs)
        try:
            self._plan_run(plan, graph_ok)
        finally:
            if comm is not None:
                comm.cycle_end(self, plan.sends, plan.recvs)

    def _plan_run(self, plan, graph_ok):
        state = plan.__dict__.setdefault("_hip", {"runs": 0, "graph": None, "failed": False})
        if state["graph"] is not None and graph_ok:
            with torch.cuda.stream(self.stream):      # (the engine's stream, whatever the caller has made current since)
                state["graph"].replay()
            self._f_stale = max(self._f_stale, state.get("f_stale", 0))   # what the replayed launches did to the F-points
            if state.get("mirror_hit") and getattr(self, "_mirror_on", False):
                self._mirror_hit = True
            return
        if graph_ok and state["runs"] >= 2 and not state["failed"]:
            import gc
            gc_was_on = gc.isenabled()
            try:
                self.sync()
                graph = torch.cuda.CUDAGraph()
                cap = self._capture_stream = getattr(self, "_capture_stream", None) or torch.cuda.Stream(device=self.device)
                gc.collect()
                gc.disable()     # a collection inside the capture could run the destructor of an old engine (hipFree,
                                 # hipStreamSynchronize): calls that invalidate a capture in progress
                was, self._f_stale = self._f_stale, 0     # (as in the launch-by-launch executions the capture repeats: the plan's
                try:                                      # nodes are the cycle's own sweeps, nothing is put in place in between)
                    with torch.cuda.graph(graph, stream=cap, capture_error_mode="thread_local"):
                        self._plan_issue(plan, cap)
                finally:
                    self._f_stale = was
                    if gc_was_on:
                        gc.enable()
                state["graph"] = graph
                with torch.cuda.stream(self.stream):
                    graph.replay()
                self._f_stale = max(self._f_stale, state.get("f_stale", 0))
                if state.get("mirror_hit") and getattr(self, "_mirror_on", False):
                    self._mirror_hit = True
                return
            except Exception as exc:   # noqa: BLE001 - capture is an optimisation: any refusal falls back to plain launches
                state["failed"] = True
                self._use_stream(self.stream)
                try:
                    torch.cuda.synchronize(self.device)   # surfaces (and clears) what the broken capture left behind
                except Exception:   # noqa: BLE001
                    pass
                if options.plan_graph == "require":
                    raise
                import warnings
                warnings.warn(f"pymgrit_amd: cycle graph capture failed ({exc!r}); launching the cycle kernel by kernel")
        state["runs"] += 1
        was, self._f_stale = self._f_stale, 0
        try:
            self._plan_issue(plan, self.stream)
            state["f_stale"] = self._f_stale          # does this cycle leave level-0 F-points to materialise()?
            state["mirror_hit"] = bool(getattr(self, "_mirror_hit", False))
        finally:
            self._f_stale = max(self._f_stale, was)   # (also when a launch failed: rows that awaited materialise() still do)

    def _masked_streams(self):
        """Heat2D planned cycle with a step-by-step coarsest-level solve (theta < 1, or fewer than 64 coarsest steps: DESIGN.md 3.8
        covers the rest): (sweep stream, chain stream) on disjoint sets of CUs (mgrit_hip_stream_create_masked), or None. 32 CUs =
        one XCD for the solve (measured in round 3 on config 4: 16 CUs 676-696 ms, 24 620, 32 501, 40 517-539, 64 528-569)."""
        if not hasattr(self, "_masked"):
            self._masked = None
            n_chain = 32
            total = torch.cuda.get_device_properties(self.device).multi_processor_count
            # (Heat2D only: its plans run launch by launch -- capturing the two masked streams into one hipGraph crashed the host
            # process; tried for config 5's F-cycle as well, launch by launch: 22.3-23.3 ms with 2-8 blocks against 20.2 with one)
            if any(d["kind"] == "heat2d" for d in self.desc) and 0 < n_chain < total:
                a, b = C.c_void_p(), C.c_void_p()
                n_sweep = total - n_chain
                if self.lib.mgrit_hip_stream_create_masked(C.byref(a), 0, n_sweep) == 0 and \
                        self.lib.mgrit_hip_stream_create_masked(C.byref(b), total - n_chain, n_chain) == 0:
                    self._masked = (torch.cuda.ExternalStream(a.value, device=self.device), torch.cuda.ExternalStream(b.value, device=self.device))
        return self._masked

    def _plan_issue(self, plan, main):
        masked = self._masked_streams() if (plan.has_chain and plan.n_blocks > 1) else None
        if masked is not None:
            # sweeps and chain on CU partitions of their own: fork both from the caller's stream, join both at the end
            sweep, side = masked
            if not hasattr(self, "_fork"):
                self._fork = torch.cuda.Event()
            self._fork.record(main)
            sweep.wait_event(self._fork)
            side.wait_event(self._fork)
            last = {}
            try:
                for node in plan.order:
                    st = side if node.stream == "chain" else sweep
                    for p in node.cross_preds:
                        st.wait_event(p.event)
                    self._use_stream(st)
                    node.fn()
                    if node.needs_event:
                        if node.event is None:
                            node.event = torch.cuda.Event()
                        node.event.record(st)
                    last[st] = node
            finally:
                self._use_stream(self.stream)
            for st in (sweep, side):
                ev = torch.cuda.Event()
                ev.record(st)
                main.wait_event(ev)
            return
        if self._chain_stream is None:
            self._chain_stream = torch.cuda.Stream(device=self.device, priority=-1)
            self._fork = torch.cuda.Event()
        side = self._chain_stream
        reserve = 0
        if plan.has_chain and plan.n_blocks > 1:
            self._fork.record(main)
            side.wait_event(self._fork)
            # the sweeps leave one CU of XCD 0 per chain worker (one worker per group of 1024 values) to the chain
            groups = max(self.ld[lvl] // 1024 for lvl in {n.lvl for n in plan.order if n.stream == "chain"})
            reserve = int(min(32, max(1, groups)))
        two = plan.has_chain and plan.n_blocks > 1
        last_side = None
        try:
            if reserve:
                check(self.lib.mgrit_hip_set_reserve(self.h, reserve))
            for node in plan.order:
                st = side if (two and node.stream == "chain") else main
                if two:
                    for p in node.cross_preds:
                        st.wait_event(p.event)
                self._use_stream(st)
                node.fn()
                if two and node.needs_event:
                    if node.event is None:
                        node.event = torch.cuda.Event()
                    node.event.record(st)
                if st is side:
                    last_side = node
        finally:
            self._use_stream(self.stream)
            if reserve:
                check(self.lib.mgrit_hip_set_reserve(self.h, 0))
        if last_side is not None:      # whatever follows on the engine's stream sees the whole cycle
            if last_side.event is None:
                last_side.event = torch.cuda.Event()
            if not last_side.needs_event:
                last_side.event.record(side)
            main.wait_event(last_side.event)

    def residual_norms(self, points):
        if getattr(self, "_res_open", None) is not None:      # several ranks: the way up has pre-filled most of the values
            handle = self.residual_begin(points)
            return self.residual_end(handle)
        cache = getattr(self, "_residual_cache", None)
        if not (cache is not None and len(cache) == len(points) and (cache is points or tuple(cache) == tuple(points))):
            self._settle(0)     # the residual kernel reads the last F-points
        if not len(points):
            return []
        host = np.empty(len(points), dtype=np.float64)
        cache = getattr(self, "_residual_cache", None)
        if cache is not None and len(cache) == len(points) and (cache is points or tuple(cache) == tuple(points)):
            check(self.lib.mgrit_hip_residual_fetch(self.h, len(points), _ptr(host)))
            return np.sqrt(host)
        check(self.lib.mgrit_hip_residual_host(self.h, 0, self._point_run_id(0, points), _ptr(host)))
        return np.sqrt(host)

    def _ring_slot(self, n):
        if not hasattr(self, "_res_ring"):
            self._res_ring, self._res_next = [], 0
        if len(self._res_ring) < 8:
            self._res_ring.append(torch.empty(n, dtype=torch.float64, pin_memory=True))
        buf = self._res_ring[self._res_next % len(self._res_ring)]
        self._res_next += 1
        return buf

    def residual_reserve(self, points, n_head):
        """several ranks: the slot the next residual_begin() / residual_norms() of exactly these points will deliver. The way up
        (ec_relax_res with out=) fills points[n_head:] -- the closing C-points of the rank's complete intervals --, the residual
        kernel later only the n_head points in front of them (they need the ghost point that arrives with op 7)."""
        buf = self._ring_slot(len(points))
        self._res_open = (tuple(points), int(n_head), buf)
        return buf

    def _open_slot(self, points):
        """(buffer, head points still to compute) when the way up has pre-filled a slot for exactly these points"""
        op, self._res_open = getattr(self, "_res_open", None), None
        if op is not None and op[0] == tuple(points):
            return op[2], list(points[:op[1]])
        return None, None

    def residual_begin(self, points):
        """launch the residual kernel and return at once; the per-point sums of squares land in pinned host memory that the
        kernel writes directly (no copy command), residual_end() waits for the event recorded behind the kernel"""
        cache = getattr(self, "_residual_cache", None)
        if len(points) and cache is not None and len(cache) == len(points) and (cache is points or tuple(cache) == tuple(points)):
            # the way up (ec_relax_res) has left exactly these values in the engine's pinned buffer: a copy of their own for a
            # solver that looks at them some cycles late -- the next cycle overwrites the engine's buffer
            buf = self._ring_slot(len(points))
            check(self.lib.mgrit_hip_residual_stash(self.h, len(points), C.c_void_p(buf.data_ptr())))
            ev = torch.cuda.Event()
            ev.record(self.stream)
            return buf, ev
        self._settle(0)
        if not len(points):
            return None
        buf, head = self._open_slot(points)
        if buf is None:
            buf, head = self._ring_slot(len(points)), points
        if len(head):
            check(self.lib.mgrit_hip_residual(self.h, 0, self._point_run_id(0, head), C.c_void_p(buf.data_ptr())))
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return buf, ev

    def _wait_event(self, ev):
        """several ranks on RCCL links: a neighbour that never sends or never receives must end in an ERROR of this rank (and with
        it of the job), not in a silent stall -- bounded wait, then the links are aborted (ncclCommAbort)"""
        if not getattr(self, "device_links", False):
            ev.synchronize()
            return
        import time
        limit = float(getattr(self.mg.comm_time, "timeout_s", 120.0))
        t0 = time.perf_counter()
        while not ev.query():
            if time.perf_counter() - t0 > limit:
                self.lib.mgrit_hip_links_close(self.h, 1)      # ncclCommAbort: the waiting kernels end, the stream drains
                abort_all = getattr(self.mg.comm_time, "abort_all", None)
                if abort_all is not None:
                    abort_all()                                 # ... and the communicator object holds none of them any longer
                raise MgritHipError(f"rank {self.mg.comm_time_rank}: the cycle did not finish within {limit} s (a neighbouring rank "
                                    f"never sent or never received): exchange links aborted")
            time.sleep(2e-5)

    def residual_end(self, handle):
        if handle is None:
            return []
        buf, ev = handle
        self._wait_event(ev)
        return np.sqrt(buf.numpy())

    def save_last(self):
        self.materialise()
        self.prev = self._U[0].clone()
        self.mg.save_values_last_iter = SlabVectorList(self.prev, self.n[0], self.mg.problem[0].vector_template,
                                                       self.perm[0])

    # -- C-point snapshots of level 0 (pipelined solve, Mgrit._solve_pipelined): rows copied inside HBM ------------
    def _snap_slot(self, slot, points):
        if not hasattr(self, "_snap"):
            self._snap, self._snap_idx = {}, torch.as_tensor(np.asarray(points, dtype=np.int64), device=self.device)
        if slot not in self._snap:
            self._snap[slot] = torch.empty((len(points), self.ld[0]), dtype=torch.float64, device=self.device)
            if len(points):   # every row valid from the start: the mirror never writes a point nobody corrects (the first time point)
                torch.index_select(self._U[0], 0, self._snap_idx, out=self._snap[slot])
        return self._snap[slot]

    def mirror_cpoints(self, slot, points):
        """the level-0 pass of the way up of the NEXT cycle (mgrit_hip_ec_relax_res) writes every corrected C-point into snapshot
        slot `slot` as well (mgrit_hip_cpoint_mirror): snapshot_cpoints(slot) after that cycle then has nothing left to copy.
        slot None: off."""
        if slot is None:
            if getattr(self, "_mirror_on", False):
                check(self.lib.mgrit_hip_cpoint_mirror(self.h, None, self._mirror_row0))
            self._mirror_on, self._mirror_slot, self._mirror_hit = False, None, False
            return
        if not len(points):
            return
        rows = self._snap_slot(slot, points)
        # row of a corrected C-point in the slot = its position among the relaxed C-points (res_pos) + the points in front of them
        # that nobody corrects (the first time point on rank 0)
        self._mirror_row0 = len(points) - len(self.mg._c_points(0))
        check(self.lib.mgrit_hip_cpoint_mirror(self.h, C.c_void_p(rows.data_ptr()), self._mirror_row0))
        self._mirror_on, self._mirror_slot, self._mirror_hit = True, slot, False

    def snapshot_cpoints(self, slot, points):
        if getattr(self, "_mirror_slot", None) == slot and getattr(self, "_mirror_hit", False) and slot in getattr(self, "_snap", {}):
            return     # the cycle's own pass has written them there
        self._snap_slot(slot, points)
        if len(points):
            torch.index_select(self._U[0], 0, self._snap_idx, out=self._snap[slot])

    def restore_cpoints(self, slot, points):
        self._residual_cache = None
        if len(points):
            self._U[0].index_copy_(0, self._snap_idx, self._snap[slot])

    def jump_norms(self, points):
        self.materialise()
        out = []
        if len(points):
            host = np.empty(len(points), dtype=np.float64)
            check(self.lib.mgrit_hip_jump_host(self.h, 0, self._point_run_id(0, points),
                                               C.c_void_p(self.prev.data_ptr()), _ptr(host)))
            out = np.sqrt(host)
        self.prev.copy_(self._U[0])
        return out

    def restrict_u(self, lvl, pairs):
        if not pairs:
            return
        if not self._device_transfer(lvl):     # the user's restriction (mgrit.py:498-500), row by row
            mg = self.mg
            for i, j in pairs:
                mg.u[lvl + 1][j] = mg.restriction[lvl](mg.u[lvl][i])
            return
        check(self.lib.mgrit_hip_restrict_u(self.h, lvl, self._pair_id(lvl, pairs)))

    def copy_u_to_v(self, lvl):
        check(self.lib.mgrit_hip_copy_u_to_v(self.h, lvl))

    def fas_rhs(self, lvl, pairs):
        self._settle(lvl)
        if not pairs:
            return
        if not self._device_transfer(lvl):
            # mgrit.py:524-547 around the user's restriction: fine half on the device (one row per pair), the rows through
            # restriction() into g of the coarse level, coarse half on the device
            mg, pid = self.mg, self._pair_id(lvl, pairs)
            rows = torch.zeros(len(pairs), self.ld[lvl], dtype=torch.float64, device=self.device)
            check(self.lib.mgrit_hip_fas_fine_rows(self.h, lvl, pid, C.c_void_p(rows.data_ptr()), self.ld[lvl]))
            self.sync()
            defects = SlabVectorList(rows, self.n[lvl], mg.problem[lvl].vector_template, self.perm[lvl])
            for p, (_, j) in enumerate(pairs):
                mg.g[lvl + 1][j] = mg.restriction[lvl](defects[p])
            check(self.lib.mgrit_hip_fas_coarse(self.h, lvl, pid))
            return
        check(self.lib.mgrit_hip_fas_rhs(self.h, lvl, self._pair_id(lvl, pairs)))

    # fused FAS residual (identity transfer, like steppers on both levels): see include/mgrit_hip.h
    def _resident(self, lvl):
        """both levels of the pair hold a state in one workgroup's registers (n <= 16384): the fused passes' precondition; wider
        Heat1D states run sweep by sweep through the three-launch Phi (csrc/mgrit_hip_wide.inc)"""
        return max(self.n[lvl], self.n[min(lvl + 1, len(self.n) - 1)]) <= hip_lib.MAX_N or self.desc[lvl]["kind"] == "heat2d"

    def can_fuse_fas(self, lvl):
        tr = self.mg.transfer_objects[lvl]
        da, db = self.desc[lvl], self.desc[lvl + 1]
        if not self._resident(lvl):
            return False
        same_forcing = len(da.get("forcing_time", [])) == len(db.get("forcing_time", []))
        return (self._device_transfer(lvl) and int(tr.device_transfer()) == hip_lib.TRANSFER_COPY and
                da["kind"] == db["kind"] and da["kind"] in ("heat1d", "advection1d") and
                self.n[lvl] == self.n[lvl + 1] and same_forcing)

    def fas_fused(self, lvl, triples, with_f_relax=False, skip_coarse_u=False):
        """fused FAS sweep (mgrit_hip_fas_fused_opts): with_f_relax folds the F-relaxation in front of it into the pass (F-points
        not stored), skip_coarse_u leaves u of a coarsest level that forward_solve overwrites unwritten"""
        self._settle(lvl)
        if not triples:
            return

        def create():
            tid = C.c_int(-1)
            fi, pr, co = _cols(triples, 3)
            check(self.lib.mgrit_hip_triples_create(self.h, lvl, len(triples), _ptr(fi), _ptr(pr), _ptr(co), C.byref(tid)))
            return tid.value
        opts = (hip_lib.FAS_WITH_F_RELAX if with_f_relax else 0) | (hip_lib.FAS_SKIP_COARSE_U if skip_coarse_u else 0)
        check(self.lib.mgrit_hip_fas_fused_opts(self.h, lvl, self._handle(self._pairs, lvl, triples, "triples", create), opts))

    def copy_pairs_u_to_v(self, lvl, pairs):
        if pairs:
            check(self.lib.mgrit_hip_copy_pairs_u_to_v(self.h, lvl, self._pair_id(lvl, pairs)))

    def at_forward_solve(self, lvl, k):
        """AtMgrit.forward_solve (at_mgrit.py:37-87): truncated, mutually independent coarsest-level solves. One rank: one
        launch on the level itself. Several ranks: a point needs the old u and the g of the k-1 points before it, which
        may live on the previous rank, and the steps in between: a private WORK level (the engine's spare level) is
        described once on the time grid [halo | own points]; per solve the halo rows arrive from the previous rank (two
        messages: u, g), the own rows are copied in, the same launch runs there and the own results are copied back. The
        arithmetic of every point is that of the one-rank run (the reference instead gathers one point per rank over its
        black / green communicators, at_mgrit.py:47-72)."""
        mg, k = self.mg, int(k)
        if mg.comm_time_size == 1:
            check(self.lib.mgrit_hip_at_solve(self.h, lvl, k))
            return
        if self.desc[lvl]["kind"] not in ("heat1d", "advection1d"):
            raise MgritHipError("AT-MGRIT on several ranks of the HIP engine: 1-D single-point steppers only")
        own_g = [int(i) for i in mg.cpts[lvl]]                 # global indices of the owned points
        counts = mg.comm_time.allgather_object(len(own_g)) if not hasattr(self, "_at") else None
        if not hasattr(self, "_at"):
            rank, size = mg.comm_time_rank, mg.comm_time_size
            holders = [r for r in range(size) if counts[r] > 0]
            prev = max([r for r in holders if r < rank], default=None) if own_g else None
            nxt = min([r for r in holders if r > rank], default=None) if own_g else None
            halo = min(k - 1, own_g[0]) if own_g else 0
            for a, b in zip(holders[:-1], holders[1:]):            # the same verdict on every rank
                if min(k - 1, sum(counts[:b])) > counts[a]:
                    raise MgritHipError(f"AT-MGRIT distance k={k} reaches past the {counts[a]} coarsest points of rank {a}: "
                                        f"use fewer ranks or a smaller k")
            first_of_next = sum(counts[:nxt]) if nxt is not None else 0
            give = min(k - 1, first_of_next) if nxt is not None else 0      # = the halo of the next holder
            at = {"prev": prev, "next": nxt, "halo": halo, "give": give, "n_own": len(own_g)}
            if own_g:
                t_at = np.ascontiguousarray(mg.global_t[lvl][own_g[0] - halo:own_g[-1] + 1])
                n, ld = self.n[lvl], self.ld[lvl]
                at["u"] = torch.zeros((t_at.size, ld), dtype=torch.float64, device=self.device)
                at["g"] = torch.zeros_like(at["u"])
                work = mg.lvl_max
                if self.desc[lvl]["kind"] == "heat1d":
                    self._describe_heat1d(work, self.desc[lvl], t_at, n, ld)
                else:
                    check(self.lib.mgrit_hip_level_advection1d(self.h, work, t_at.size, _ptr(t_at), n, ld, float(self.desc[lvl]["fac"])))
                check(self.lib.mgrit_hip_level_bind(self.h, work, C.c_void_p(at["u"].data_ptr()), C.c_void_p(at["u"].data_ptr()),
                                                    C.c_void_p(at["g"].data_ptr())))
            self._at = at
        at = self._at
        if not at["n_own"]:
            return
        own = slice(int(mg.index_local[lvl][0]), int(mg.index_local[lvl][-1]) + 1)   # rows of the owned points in the slab
        U, G = self._U[lvl], self.G[lvl]
        for slab, work in ((U, at["u"]), (G, at["g"])):        # old u, then g: last rows to the next holder, halo from the previous
            send = (slab[own][at["n_own"] - at["give"]:], at["next"]) if at["give"] else None
            recv = (work[:at["halo"]], at["prev"]) if at["halo"] else None
            if send is not None or recv is not None:
                mg.comm_time.exchange(send=send, recv=recv)
            work[at["halo"]:].copy_(slab[own])
        check(self.lib.mgrit_hip_at_solve(self.h, mg.lvl_max, k))
        U[own].copy_(at["u"][at["halo"]:])

    def can_fuse_ec(self, lvl):
        tr = self.mg.transfer_objects[lvl]
        da, db = self.desc[lvl], self.desc[lvl + 1]
        if not self._resident(lvl):
            return False
        return (self._device_transfer(lvl) and int(tr.device_transfer()) == hip_lib.TRANSFER_COPY and
                da["kind"] == db["kind"] and da["kind"] in ("heat1d", "advection1d") and self.n[lvl] == self.n[lvl + 1])

    def ec_relax(self, lvl, triples):
        """error correction of the C-point in front of each run + the run's F-relaxation in one launch"""
        self._settle(lvl)
        self._before_c_write(lvl)
        if lvl == 0:
            self._residual_cache = None
        if not triples:
            return

        def create():
            rid = C.c_int(-1)
            st, ln, co = _cols(triples, 3)
            check(self.lib.mgrit_hip_ec_runs_create(self.h, lvl, len(triples), _ptr(st), _ptr(ln), _ptr(co), C.byref(rid)))
            return rid.value
        check(self.lib.mgrit_hip_ec_relax(self.h, lvl, self._handle(self._runs, lvl, triples, "ecruns", create)))

    # -- whole-level sweeps in one pass (include/mgrit_hip.h: mgrit_hip_cf_fas / mgrit_hip_ec_relax_res) -------------------
    def can_fuse_level(self, lvl):
        """level 0, Heat1D with a separable forcing on both levels, identity transfer (weight and layout: the caller)"""
        tr = self.mg.transfer_objects[lvl]
        da, db = self.desc[lvl], self.desc[lvl + 1]
        return (lvl == 0 and self._resident(lvl) and not options.no_level_fusion and self._device_transfer(lvl) and
                int(tr.device_transfer()) == hip_lib.TRANSFER_COPY and da["kind"] == db["kind"] == "heat1d" and
                self.n[lvl] == self.n[lvl + 1] and da.get("forcing_rows") is None and db.get("forcing_rows") is None and
                len(da.get("forcing_time", [])) == len(db.get("forcing_time", [])))

    def can_fuse_coarse_down(self, lvl):
        """lvl > 0, Heat1D with a separable forcing on lvl and lvl+1, identity transfer: the way down of the level as two
        passes -- F-relaxation + C-relaxation (relax mode FC), F-relaxation + FAS residual (fas_fused with_f_relax)"""
        da = self.desc[lvl]
        return (lvl > 0 and not options.no_level_fusion and self.can_fuse_fas(lvl) and
                da["kind"] == "heat1d" and da.get("forcing_rows") is None and self.desc[lvl + 1].get("forcing_rows") is None)

    def can_fuse_level_up(self, lvl):
        """any level pair of Heat1D with a separable forcing and the identity transfer: error correction + F-relaxation in one
        pass (mgrit_hip_ec_relax_res; with the rows of g and without the residual on lvl > 0)"""
        tr = self.mg.transfer_objects[lvl]
        da, db = self.desc[lvl], self.desc[lvl + 1]
        # Its intervals correct the C-point they END on, so in a planned cycle a block waits for the chain part of its own block
        # only (ecf_kernel: of the next block too). The launch itself is slower than ecf_kernel (every interval reads its two
        # boundary corrections), so it pays only together with the merged launch of the first blocks' way up (cycle_plan.py,
        # Recorder.up_merge) that it makes possible: config 3, six blocks, 8.60 -> 8.24 ms; alone 8.46. On by default exactly
        # there (a planned cycle of five or more blocks); options.fuse_up_coarse = True / False forces it on / off.
        want = {True: "1", False: "0"}.get(options.fuse_up_coarse, "")
        if want not in ("0", "1"):
            # (a rank of a sharded run: from two blocks on -- its way up of a block would otherwise wait for the chain part of
            # the next block, and the rank has few blocks to hide that behind)
            blocks = self.mg.plan_blocks()
            want = "1" if (blocks >= 5 or (blocks >= 2 and self.mg.comm_time_size > 1)) else "0"
        return (want == "1" and self._resident(lvl) and
                not options.no_level_fusion and self._device_transfer(lvl) and
                int(tr.device_transfer()) == hip_lib.TRANSFER_COPY and da["kind"] == db["kind"] == "heat1d" and
                self.n[lvl] == self.n[lvl + 1] and da.get("forcing_rows") is None and db.get("forcing_rows") is None and
                len(da.get("forcing_time", [])) == len(db.get("forcing_time", [])))

    def _intervals_id(self, lvl, intervals, chunk=None):
        def create():
            iid = C.c_int(-1)
            cols = _cols(intervals, 6)
            # level 0: chunks of 4 intervals (one extra row + Phi per chunk start); coarser levels: one interval per item --
            # a block of a planned cycle holds only a few hundred of their intervals, and 4 in a row would leave CUs idle
            # (0 = chosen by the library from the level's size: 4 on config 3, 1 where the level has fewer intervals than the chip
            # holds workgroups)
            ch = chunk
            if ch is None:      # the Heat1D whole-level passes: level 0 by the library's rule for them (up to 16 intervals in a row)
                ch = hip_lib.CHUNK_LONG if lvl == 0 else 1
            res_len = len(self.mg._c_points(lvl))
            check(self.lib.mgrit_hip_intervals_create(self.h, lvl, len(intervals), _ptr(cols[0]), _ptr(cols[1]), _ptr(cols[2]),
                                                      _ptr(cols[3]), _ptr(cols[4]), res_len, ch, _ptr(cols[5]), C.byref(iid)))
            return iid.value
        return self._handle(self._runs, lvl, intervals, "ivals", create)

    def cf_fas(self, lvl, intervals):
        """c_relax + f_relax + fas_residual of level lvl for the intervals (cstart, cend, cstart_coarse, cend_coarse, res_pos,
        keep): keep = which rows of lvl+1 the closing C-point needs (bit 0: u, bit 1: v; include/mgrit_hip.h)"""
        if intervals:
            self._residual_cache = None
            check(self.lib.mgrit_hip_cf_fas(self.h, lvl, self._intervals_id(lvl, intervals), 1 if (lvl == 0 and self._cycle_pre) else 0))

    def ec_relax_res_to(self, lvl, intervals, buf):
        """several ranks: error_correction + f_relax + the residual sums of the rank's complete intervals (every F-point stored:
        neighbours and the generic sweeps read them), sums into the reserved slot `buf` (residual_reserve)"""
        if intervals:
            check(self.lib.mgrit_hip_ec_relax_res_to(self.h, lvl, self._intervals_id(lvl, intervals), 1, C.c_void_p(buf.data_ptr())))

    def ec_relax_res(self, lvl, intervals, base=0):
        """error_correction + f_relax + compute_residual; the per-point sums of squares stay in pinned host memory until
        residual_norms() asks for exactly these points. F-points: C-point storage (materialise()) unless
        PYMGRIT_AMD_STORE_ALL_F=1"""
        if intervals:
            lazy = lvl == 0 and not options.store_all_f
            # 2: the last F-point's row gets Phi of it, which the next cycle's cf_fas takes as its C-relaxation -- when that pass
            # IS the next reader of the level (cf_iter = 1: no plain C-relaxation in front of it)
            mode = 1 if not lazy else (2 if (self.mg.cf_iter[0] == 1 and not options.no_pre_relax) else 0)
            check(self.lib.mgrit_hip_ec_relax_res(self.h, lvl, self._intervals_id(lvl, intervals), mode))
            if lvl == 0 and getattr(self, "_mirror_on", False):
                self._mirror_hit = True
            if lazy:
                self._f_stale = max(self._f_stale, 2 if mode == 2 else 1)

    # -- the same two passes for any 1-D stepper pair and any of the library's transfers (mgrit_hip_gen_down / mgrit_hip_gen_up) --
    def can_gen_level(self, lvl):
        """Heat1D (any forcing) or Advection1D on lvl and lvl+1, joined by a transfer the kernels apply (copy, full weighting with
        Dirichlet ends, its periodic analogue), both states register-resident"""
        if lvl + 1 >= len(self.desc) or options.no_level_fusion or options.no_gen_passes:
            return False
        da, db = self.desc[lvl], self.desc[lvl + 1]
        return (self._device_transfer(lvl) and da["kind"] == db["kind"] and da["kind"] in ("heat1d", "advection1d") and
                max(self.n[lvl], self.n[lvl + 1]) <= hip_lib.MAX_N)

    @staticmethod
    def _gen_chunk():
        """intervals a workgroup of the general passes walks in a row: 0 = the library's choice from the level's size (a chunk's
        first C-point costs the way down a row and a Phi more, the way up a row of the side slab), on every level"""
        return 0

    def gen_down(self, lvl, intervals, parts=3):
        """c_relax + f_relax + fas_residual of level lvl for the intervals (cstart, cend, cstart_coarse, cend_coarse, res_pos, keep);
        parts: 1 = the fine level's pass with the restriction, 2 = the coarse half (a rank's op 4 sits between them), 3 = both"""
        self._settle(lvl)
        if intervals:
            if lvl == 0:
                self._residual_cache = None
            check(self.lib.mgrit_hip_gen_down_part(self.h, lvl, self._intervals_id(lvl, intervals, chunk=self._gen_chunk()), int(parts)))

    def gen_up(self, lvl, intervals, residual=False):
        """error_correction + f_relax (+ compute_residual on level 0, values kept for residual_norms) of level lvl"""
        if intervals:
            if lvl == 0:
                self._residual_cache = None
            check(self.lib.mgrit_hip_gen_up(self.h, lvl, self._intervals_id(lvl, intervals, chunk=self._gen_chunk()), 1 if residual else 0, None))

    def residual_ready(self, points):
        """the residual of exactly these level-0 points has been produced by the last ec_relax_res sweep(s) and level 0 has not
        been touched since (Mgrit._ec_f_relax sets it, every other sweep on level 0 clears it)"""
        self._residual_cache = points      # (compared by identity first: Mgrit hands over the same cached list every time)

    def error_correction(self, lvl, pairs):
        self._before_c_write(lvl)
        if lvl == 0:
            self._residual_cache = None
        if pairs and not self._device_transfer(lvl):     # mgrit.py:724-726 through the user's interpolation
            mg = self.mg
            for i, j in pairs:
                mg.u[lvl][i] = mg.u[lvl][i] + mg.interpolation[lvl](mg.u[lvl + 1][j] - mg.v[lvl + 1][j])
        elif pairs:
            check(self.lib.mgrit_hip_error_correction(self.h, lvl, self._pair_id(lvl, pairs)))

    def interpolate(self, lvl, pairs):
        self._before_c_write(lvl)
        if lvl == 0:
            self._residual_cache = None
        if pairs and not self._device_transfer(lvl):     # mgrit.py:559-563
            mg = self.mg
            for i, j in pairs:
                mg.u[lvl][i] = mg.interpolation[lvl](u=mg.u[lvl + 1][j])
        elif pairs:
            check(self.lib.mgrit_hip_interpolate(self.h, lvl, self._pair_id(lvl, pairs)))

    def sync(self):
        if getattr(self, "device_links", False):   # a neighbour that never sends or never receives must end in an error here
            rc = self.lib.mgrit_hip_sync_bounded(self.h, float(getattr(self.mg.comm_time, "timeout_s", 120.0)))
            if rc != 0:     # the library has aborted this engine's links: the communicator object must not keep their handles
                abort_all = getattr(self.mg.comm_time, "abort_all", None)
                if abort_all is not None:
                    abort_all()
            check(rc)
        else:
            check(self.lib.mgrit_hip_sync(self.h))

    # -- measurement hooks (bench.py) ------------------------------------------------------------------
    def set_timing(self, on):
        self._timing_on = bool(on)
        check(self.lib.mgrit_hip_set_timing(self.h, int(bool(on))))

    def chain_clock(self):
        """(shader MHz, us per step) of the most recent overlapped chain launch (diagnostics)"""
        mhz, us = C.c_double(0.0), C.c_double(0.0)
        check(self.lib.mgrit_hip_chain_clock(self.h, C.byref(mhz), C.byref(us)))
        return mhz.value, us.value

    def timing_drain(self, max_records=4096):
        """[(sweep kind, level, milliseconds)] of every timed entry-point call since the last drain (waits for them)"""
        kinds = np.zeros(max_records, dtype=np.int32)
        lvls = np.zeros(max_records, dtype=np.int32)
        ms = np.zeros(max_records, dtype=np.float32)
        n = C.c_int(0)
        check(self.lib.mgrit_hip_timing_drain(self.h, max_records, _ptr(kinds), _ptr(lvls), _ptr(ms), C.byref(n)))
        return [(hip_lib.TIMED_KINDS[int(kinds[i])], int(lvls[i]), float(ms[i])) for i in range(n.value)]

    def last_kernel_ms(self):
        ms = C.c_float(0.0)
        check(self.lib.mgrit_hip_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value
