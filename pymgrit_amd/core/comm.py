"""Time communicator: the rank-to-rank ghost exchange of the reference's mpi4py path
(``Mgrit.send/receive``, reference src/pymgrit/core/mgrit.py:693-713, pickled isend + blocking recv) re-expressed on
``torch.distributed`` -- one process per GPU, backend "nccl" (= RCCL over xGMI on MI355X) for device rows, "gloo" on
CPU. Payloads are either device/CPU tensors (sent in place, no pickling) or small picklable Python objects.
"""
import os
import pickle
import warnings

import numpy as np
import torch
import torch.distributed as dist

from pymgrit_amd.core.options import options


class SerialComm:
    """A communicator of size 1 (no exchange ever happens)."""
    rank, size = 0, 1

    def Get_rank(self):
        return 0

    def Get_size(self):
        return 1

    def barrier(self):
        return None

    def allgather_object(self, obj):
        return [obj]

    def exchange(self, send=None, recv=None):
        raise RuntimeError("exchange on a size-1 communicator")

    def prepare(self):
        return None

    def drain(self):
        return None


class _Gather:
    """Handle of a posted all-gather of per-rank float lists (TorchTimeComm.iallgather_floats)."""

    def __init__(self, work, out, counts_work, counts):
        self.work, self.out, self.counts_work, self.counts = work, out, counts_work, counts

    def result(self):
        """blocks until every rank has posted the same gather; returns the per-rank lists in rank order"""
        self.counts_work.wait()
        self.work.wait()
        return [self.out[r, :int(self.counts[r].item())].tolist() for r in range(self.out.shape[0])]


class TorchTimeComm:
    """Nearest-owner point-to-point on a ``torch.distributed`` process group.

    Streams of messages that must not wait for each other get their own communicator:
      * one 2-rank group per pair of neighbouring ranks (r, r+1): a send to the next rank that its receiver has not posted
        yet (the receiver is still busy with earlier work) must not hold back this rank's receive from the previous rank,
        which it would on a shared NCCL communicator (one stream, rendezvous sends);
      * a small gloo group for the convergence values, which travel as host floats and are collected asynchronously
        (``iallgather_floats``) while the sweeps of the following iterations already run.
    Sends never block the caller: the payload is copied into a staging buffer that lives until the send has completed
    (``exchange`` reaps finished sends), so the slab row it came from may be overwritten at once."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self._inflight = []
        self._links = None
        self._side = None
        self.stats = {"messages": 0, "bytes": 0, "device_messages": 0}   # sends posted by this rank (bench.py reports them)

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def _global(self, r):
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def barrier(self):
        dist.barrier(group=self.group)

    def allgather_object(self, obj):
        out = [None] * self.size
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def _device(self):
        return torch.device("cuda", torch.cuda.current_device()) if self.backend == "nccl" else torch.device("cpu")

    # ---- communicators ----------------------------------------------------------------------------------------------
    def _link(self, a, b):
        """process group for the pair of ranks (a, b): its own 2-rank group for neighbours, the main group otherwise.
        Groups are created collectively by all ranks the first time any link is needed."""
        if self._links is None:
            self._links = {}
            for r in range(self.size - 1):   # every rank creates every group, in the same order (new_group is collective)
                self._links[r] = dist.new_group([self._global(r), self._global(r + 1)], backend=self.backend)
        lo, hi = min(a, b), max(a, b)
        return self._links.get(lo, self.group) if hi == lo + 1 else self.group

    def _side_group(self):
        if self._side is None:
            self._side = dist.new_group([self._global(r) for r in range(self.size)], backend="gloo")
        return self._side

    def prepare(self):
        """create all communicators now (collective; call at the same program point on every rank). If the creation is
        refused (a backend without sub-groups, PYMGRIT_AMD_NO_EXTRA_GROUPS set) every rank falls back to the main group
        alone: exchanges still work (one shared stream), the asynchronous gather is switched off and with it the
        pipelined solve loop (Mgrit.pipeline_depth). (A rank that dies INSIDE a collective creation cannot be survived.)"""
        if self.size <= 1 or self._links is not None:
            return
        ok = True
        try:
            if options.no_extra_groups:   # switch (and test hook): main group only, on every rank
                raise RuntimeError("extra communicators disabled by PYMGRIT_AMD_NO_EXTRA_GROUPS")
            self._link(0, 1)
            self._side_group()
        except Exception as exc:   # noqa: BLE001 - any failure of the optional communicators is handled the same way
            ok = False
            warnings.warn(f"pymgrit_amd: extra communicators unavailable on rank {self.rank} ({exc!r}); using the main group only")
        if not all(self.allgather_object(ok)):
            self._links, self._side, self.async_gather = {}, None, False

    async_gather = True

    # ---- convergence values -------------------------------------------------------------------------------------------
    def iallgather_floats(self, values, max_count):
        """post an all-gather of this rank's list of floats (at most max_count per rank, the same bound on all ranks)"""
        g = self._side_group()
        mine = torch.zeros(max_count, dtype=torch.float64)
        if len(values):
            mine[:len(values)] = torch.from_numpy(np.ascontiguousarray(np.asarray(values, dtype=np.float64)))
        out = torch.zeros(self.size * max_count, dtype=torch.float64)   # flat: rank r fills [r*max_count, (r+1)*max_count)
        counts = torch.zeros(self.size, dtype=torch.int64)
        cw = dist.all_gather_into_tensor(counts, torch.tensor([len(values)], dtype=torch.int64), group=g, async_op=True)
        w = dist.all_gather_into_tensor(out, mine, group=g, async_op=True)
        return _Gather(w, out.view(self.size, max_count), cw, counts)

    # ---- point to point -------------------------------------------------------------------------------------------------
    def _reap(self, block=False):
        keep = []
        for work, staged in self._inflight:
            if block:
                work.wait()
            elif not work.is_completed():
                keep.append((work, staged))
        self._inflight = keep

    def drain(self):
        """wait for every send posted so far (end of a solve)"""
        self._reap(block=True)

    def exchange(self, send=None, recv=None):
        """One exchange point: optionally send ``(payload, dest)`` and/or receive ``(buffer_or_None, src)``.
        Tensors travel as they are (the receive buffer must be a tensor of the same shape); other payloads are
        pickled. The send is posted from a staging copy and not waited for; the receive is waited for. Returns the received
        object (the filled tensor, or the unpickled payload)."""
        self._reap()
        result = None
        if send is not None:
            payload, dest = send
            g = self._link(self.rank, dest)
            if torch.is_tensor(payload):
                staged = payload.detach().cpu() if (payload.is_cuda and self.backend != "nccl") else payload.detach().clone()
                self.stats["messages"] += 1
                self.stats["bytes"] += staged.numel() * staged.element_size()
                self.stats["device_messages"] += int(staged.is_cuda)
                self._inflight.append((dist.isend(staged, self._global(dest), group=g), staged))
            else:
                raw = torch.frombuffer(bytearray(pickle.dumps(payload)), dtype=torch.uint8).to(self._device())
                size = torch.tensor([raw.numel()], dtype=torch.int64, device=self._device())
                self.stats["messages"] += 1
                self.stats["bytes"] += raw.numel()
                self._inflight.append((dist.isend(size, self._global(dest), group=g), size))
                self._inflight.append((dist.isend(raw, self._global(dest), group=g), raw))
        if recv is not None:
            buf, src = recv
            g = self._link(self.rank, src)
            if torch.is_tensor(buf):
                staged = torch.empty(buf.shape, dtype=buf.dtype, device="cpu") if (buf.is_cuda and self.backend != "nccl") else None
                dist.irecv(staged if staged is not None else buf, self._global(src), group=g).wait()
                if staged is not None:
                    buf.copy_(staged)
                result = buf
            else:
                size = torch.zeros(1, dtype=torch.int64, device=self._device())
                dist.irecv(size, self._global(src), group=g).wait()
                raw = torch.empty(int(size.item()), dtype=torch.uint8, device=self._device())
                dist.irecv(raw, self._global(src), group=g).wait()
                result = pickle.loads(raw.cpu().numpy().tobytes())
        return result


# ----------------------------------------------------------------------------------------------------------------------
# Device exchange: ghost rows as stream operations of the HIP engine (include/mgrit_hip.h: mgrit_hip_exchange)
# ----------------------------------------------------------------------------------------------------------------------
CH_SWEEP, CH_CHAIN = 0, 1     # channel of a link: every exchange point but op 5 / the hand-over of forward_solve (op 5)


class _EngineLinks:
    """link handles of one engine: (peer, direction, channel) -> handle index of include/mgrit_hip.h"""

    def __init__(self):
        self.handle = {}

    def add(self, key):
        from pymgrit_amd.core import hip_lib
        if key not in self.handle:
            if len(self.handle) >= hip_lib.MAX_LINKS:
                raise RuntimeError("more exchange links than the engine has handles")
            self.handle[key] = len(self.handle)
        return self.handle[key]


def links_needed(mg):
    """the directed links (src, dst, channel) this rank takes part in, from the layout of every level (Mgrit.send_to /
    get_from, reference mgrit.py:816-827): channel CH_CHAIN for the hand-over of the coarsest level's forward solve"""
    rank, need = mg.comm_time_rank, set()
    for lvl in range(mg.lvl_max):
        for src, dst in ((mg.get_from[lvl], rank), (rank, mg.send_to[lvl])):
            if src >= 0 and dst >= 0 and src != dst:
                need.add((int(src), int(dst), CH_SWEEP))
                if lvl == mg.lvl_max - 1:
                    need.add((int(src), int(dst), CH_CHAIN))
    return sorted(need)


class RcclTimeComm(TorchTimeComm):
    """TorchTimeComm whose ghost rows travel under the C ABI: ncclSend / ncclRecv issued by libmgrit_hip.so on the engine's
    streams (mgrit_hip_exchange), one two-rank RCCL communicator per directed link and channel. torch.distributed keeps the
    host-side collectives (rendezvous of the unique ids, the stopping values, barriers). Default time communicator when the
    process group's backend is "nccl"; PYMGRIT_AMD_EXCHANGE=torch keeps every exchange in torch.distributed."""
    device_exchange = True

    def __init__(self, group=None):
        super().__init__(group)
        self._comms = {}       # (src, dst, channel) -> ncclComm_t (c_void_p), created once per process group
        self._engines = {}     # id(backend) -> _EngineLinks
        self.timeout_s = float(options.exchange_timeout)

    def open_links(self, backend, need):
        """collective: every rank names the links it takes part in; the sending rank of a link makes the unique id, everybody
        learns all ids, and the two ends of each link create its communicator -- links in ONE global order on all ranks
        (senders of even rank first: those run in parallel), so no two ranks ever wait for each other crosswise."""
        import ctypes as C
        from pymgrit_amd.core import hip_lib
        lib = hip_lib.load()
        all_need = sorted({tuple(k) for part in self.allgather_object(list(need)) for k in part})
        missing = [k for k in all_need if k not in self._comms]
        if missing:
            mine, failed = {}, None
            try:
                for k in missing:
                    if k[0] == self.rank:
                        buf = C.create_string_buffer(128)
                        hip_lib.check(lib.mgrit_hip_comm_unique_id(buf))
                        mine[k] = buf.raw
            except Exception as exc:      # noqa: BLE001 - told to every rank with the ids: nobody enters a rendezvous alone
                failed = f"rank {self.rank}: {exc!r}"[:300]
            ids, errs = {}, []
            for part, err in self.allgather_object((mine, failed)):
                ids.update(part)
                if err:
                    errs.append(err)
            if errs:
                raise RuntimeError(errs[0])
            for k in sorted(missing, key=lambda k: (k[0] % 2, k)):
                if self.rank in k[:2]:
                    comm = C.c_void_p()
                    hip_lib.check(lib.mgrit_hip_comm_init_rank(C.byref(comm), ids[k], 2, 0 if k[0] == self.rank else 1))
                    self._comms[k] = comm
        links = self._engines[id(backend)] = _EngineLinks()
        for k in need:
            k = tuple(k)
            send = k[0] == self.rank
            h = links.add((k[1] if send else k[0], 'send' if send else 'recv', k[2]))
            hip_lib.check(lib.mgrit_hip_link_attach(backend.h, h, self._comms[k], 1 if send else 0))

    def open_links_agreed(self, backend, need):
        """open_links, then ONE word from every rank: either all ranks have their links or none keeps any (the caller then runs
        the exchange through torch.distributed). Returns None or the first error. A rank that fails while its peer is already
        inside the communicator's rendezvous cannot be helped from here -- this covers the failures every rank sees (no librccl,
        a transport that does not come up) and those in front of the first rendezvous."""
        err = None
        try:
            self.open_links(backend, need)
        except Exception as exc:      # noqa: BLE001 - reported to every rank below
            err = f"rank {self.rank}: {exc!r}"[:300]
        errs = [e for e in self.allgather_object(err) if e is not None]
        if not errs:
            return None
        from pymgrit_amd.core import hip_lib
        lib = hip_lib.load()
        if getattr(backend, "h", None) is not None:
            lib.mgrit_hip_links_close(backend.h, 0)      # detach only: the communicators are this object's to end, once (below)
        self.abort_all()
        self._engines.pop(id(backend), None)
        return errs[0]

    def abort_all(self):
        """end every communicator exactly once (ncclCommAbort; one that the engine has aborted already -- a bounded wait that
        gave up -- is only forgotten: mgrit_hip_comm_destroy knows) and hold none afterwards"""
        from pymgrit_amd.core import hip_lib
        lib = hip_lib.load()
        comms, self._comms = self._comms, {}
        for comm in comms.values():
            lib.mgrit_hip_comm_destroy(comm, 1)

    def ping_links(self, backend, buf_ptr, reps=5, timeout=20.0):
        """collective pre-flight of every link of the job, one after the other in ONE global order (a barrier in front of each):
        the sending rank enqueues `reps` messages of 8 doubles from buf_ptr (a device buffer of >= 8 doubles) and waits for its
        stream, the receiving rank likewise; both waits are bounded. Returns {"src->dst/ch": microseconds per message (max of the
        two ends)} on every rank, or raises the same RuntimeError on every rank naming the first pair that did not answer."""
        import ctypes as C
        import time
        from pymgrit_amd.core import hip_lib
        lib = hip_lib.load()
        mine = self._engines[id(backend)].handle
        every = sorted({tuple(k) for part in self.allgather_object([list(k) for k in self._comms]) for k in part})
        res = {}
        for src, dst, ch in every:
            self.barrier()
            us, err = None, None
            role = 'send' if src == self.rank else 'recv' if dst == self.rank else None
            if role is not None:
                h = mine.get((dst if role == 'send' else src, role, ch))
                if h is None:        # the link exists in the job but this engine has not attached it
                    err = f"link {src}->{dst} channel {ch}: not attached on rank {self.rank}"
                else:
                    fn = lib.mgrit_hip_send if role == 'send' else lib.mgrit_hip_recv
                    t0 = time.perf_counter()
                    rc = 0
                    for _ in range(reps):
                        rc = rc or fn(backend.h, h, 0, C.c_void_p(buf_ptr), 8)
                    rc = rc or lib.mgrit_hip_sync_bounded(backend.h, float(timeout))
                    us = 1e6 * (time.perf_counter() - t0) / reps
                    if rc != 0:
                        err = f"link {src}->{dst} channel {ch}: rank {self.rank} ({role}) got no answer within {timeout} s"
            got = self.allgather_object((us, err))
            errs = [e for _, e in got if e]
            if errs:
                self.abort_all()     # a bounded wait that gave up has aborted links inside the library: hold no handle of them
                raise RuntimeError("exchange links: " + errs[0])
            res[f"{src}->{dst}/{ch}"] = max(u for u, _ in got if u is not None)
        return res

    def send_begin(self, backend, dest, channel):
        self.stats["messages"] += 1
        self.stats["device_messages"] += 1
        return self._engines[id(backend)].handle[(int(dest), 'send', channel)], 0

    def recv_begin(self, backend, src, channel):
        return self._engines[id(backend)].handle[(int(src), 'recv', channel)], 0

    def send_end(self, backend, dest, channel):
        return None

    recv_end = send_end

    def cycle_slot(self, backend, peer, direction, channel, ordinal):
        """(handle, slot) of the ordinal-th message of a planned cycle on a link: RCCL matches by order, no slots"""
        return self._engines[id(backend)].handle[(int(peer), direction, channel)], 0

    def cycle_begin(self, backend, sends, recvs):
        return None

    def cycle_end(self, backend, sends, recvs):
        for key, n in sends.items():
            self.stats["messages"] += n
            self.stats["device_messages"] += n

    def close(self):
        from pymgrit_amd.core import hip_lib
        lib = hip_lib.load()
        comms, self._comms = self._comms, {}
        for comm in comms.values():
            lib.mgrit_hip_comm_destroy(comm, 0)


class _LoopLink:
    """one directed link of a LoopbackWorld: a mailbox in device memory + the host-side counters of the hand-shake"""

    def __init__(self, mailbox, n_ring, n_cycle):
        self.mailbox, self.n_ring, self.n_cycle = mailbox, n_ring, n_cycle
        self.sent = self.received = 0            # single messages (ring slots)
        self.cyc_sent = self.cyc_received = 0    # whole cycles of a planned run (cycle slots)


class LoopbackWorld:
    """All ranks of a run in ONE process on ONE GPU: ranks are threads (or, frozen, a single rank replaying against what its
    neighbours last sent: bench.py --emulate-rank). Ghost rows go through mailboxes in device memory (mgrit_hip_link_mailbox):
    a send copies the row into a slot, a receive copies it out, both as operations of the ONE stream all ranks share -- so
    the device executes them in the order the hosts enqueued them, and the hand-shake only has to make a receiver enqueue
    after its sender (and a sender reuse a slot after its receiver): host-side counters under one condition variable.
    Single messages use a ring of slots; the messages of a planned cycle (core/cycle_plan.py) use fixed slots -- the k-th
    message of the cycle on a link always the same one, as a captured graph needs it -- and shake hands once per cycle."""

    N_RING, N_CYCLE = {CH_SWEEP: 64, CH_CHAIN: 8}, {CH_SWEEP: 128, CH_CHAIN: 8}

    def __init__(self, size):
        import threading
        self.size = int(size)
        self.cond = threading.Condition()
        self.barrier_obj = threading.Barrier(self.size)
        self.links, self.objects, self.gathers, self.queues = {}, {}, {}, {}
        self.frozen = False        # no hand-shake: a single rank against the slots' last contents
        self.stream = None
        self.timeout = float(options.loopback_timeout)

    def comm(self, rank):
        return LoopbackComm(self, rank)

    def link(self, key, slot_doubles):
        import ctypes as C
        from pymgrit_amd.core import hip_lib
        with self.cond:
            ln = self.links.get(key)
            if ln is None:
                lib = hip_lib.load()
                n_ring, n_cycle = self.N_RING[key[2]], self.N_CYCLE[key[2]]
                mb = C.c_void_p()
                hip_lib.check(lib.mgrit_hip_mailbox_create(C.byref(mb), n_ring + n_cycle, int(slot_doubles)))
                ln = self.links[key] = _LoopLink(mb, n_ring, n_cycle)
                ln.slot_doubles = int(slot_doubles)
            elif ln.slot_doubles < slot_doubles:
                raise RuntimeError("loopback link reused with wider rows")
            return ln

    def wait(self, pred, what):
        """cond is held by the caller"""
        if not self.cond.wait_for(pred, timeout=self.timeout):
            raise RuntimeError(f"loopback exchange: {what} did not happen within {self.timeout} s (deadlock?)")

    def close(self):
        from pymgrit_amd.core import hip_lib
        lib = hip_lib.load()
        for ln in self.links.values():
            lib.mgrit_hip_mailbox_destroy(ln.mailbox)
        self.links = {}


class LoopbackComm:
    """time communicator of one rank of a LoopbackWorld (see there)"""
    device_exchange = True
    async_gather = True

    def __init__(self, world, rank):
        self.world, self.rank, self.size = world, int(rank), world.size
        self._engines, self._objseq, self._gseq = {}, 0, 0
        self.stats = {"messages": 0, "bytes": 0, "device_messages": 0}

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def prepare(self):
        return None

    def drain(self):
        return None

    def barrier(self):
        if not self.world.frozen:
            self.world.barrier_obj.wait(timeout=self.world.timeout)

    def allgather_object(self, obj):
        w = self.world
        if w.frozen:
            raise RuntimeError("a frozen loopback world has no collectives")
        self._objseq += 1
        with w.cond:
            w.objects.setdefault(self._objseq, {})[self.rank] = obj
            w.cond.notify_all()
            w.wait(lambda: len(w.objects[self._objseq]) == w.size, "all-gather")
            return [w.objects[self._objseq][r] for r in range(w.size)]

    def iallgather_floats(self, values, max_count):
        w = self.world
        self._gseq += 1
        seq = self._gseq
        with w.cond:
            w.gathers.setdefault(seq, {})[self.rank] = [float(v) for v in values]
            w.cond.notify_all()

        class _Handle:
            def result(_self):
                with w.cond:
                    w.wait(lambda: len(w.gathers[seq]) == w.size, "stopping values")
                    return [list(w.gathers[seq][r]) for r in range(w.size)]
        return _Handle()

    def exchange(self, send=None, recv=None):
        """host-driven exchange (small Python payloads; rows of runs that do not use the device exchange): first-in first-out
        queue per directed pair; tensors are cloned on the shared stream"""
        import copy
        import queue
        w, result = self.world, None
        if send is not None:
            payload, dest = send
            item = payload.detach().clone() if torch.is_tensor(payload) else copy.deepcopy(payload)
            with w.cond:
                q = w.queues.setdefault((self.rank, int(dest)), queue.Queue())
            q.put(item)
        if recv is not None:
            buf, src = recv
            with w.cond:
                q = w.queues.setdefault((int(src), self.rank), queue.Queue())
            item = q.get(timeout=w.timeout)
            if buf is not None and hasattr(buf, "copy_"):
                buf.copy_(item)
                result = buf
            else:
                result = item
        return result

    # ---- device exchange ------------------------------------------------------------------------------------------------
    def open_links(self, backend, need):
        from pymgrit_amd.core import hip_lib
        lib = hip_lib.load()
        w = self.world
        with w.cond:    # mailboxes order their copies by the ONE stream all ranks enqueue on
            if w.stream is None:
                w.stream = backend.stream.cuda_stream
            elif w.stream != backend.stream.cuda_stream:
                raise RuntimeError("loopback ranks must share one stream")
        links = self._engines[id(backend)] = _EngineLinks()
        links.of = {}
        for k in need:
            k = tuple(k)
            send = k[0] == self.rank
            ln = w.link(k, 2 * max(backend.ld) + 64 if k[2] == CH_CHAIN else max(backend.ld))
            h = links.add((k[1] if send else k[0], 'send' if send else 'recv', k[2]))
            links.of[h] = ln
            hip_lib.check(lib.mgrit_hip_link_mailbox(backend.h, h, ln.mailbox))

    def _link(self, backend, peer, direction, channel):
        links = self._engines[id(backend)]
        h = links.handle[(int(peer), direction, channel)]
        return h, links.of[h]

    def send_begin(self, backend, dest, channel):
        h, ln = self._link(backend, dest, 'send', channel)
        w = self.world
        self.stats["messages"] += 1
        self.stats["device_messages"] += 1
        if w.frozen:
            return h, 0
        with w.cond:     # the slot is free once the message n_ring places back has been taken
            w.wait(lambda: ln.sent - ln.received < ln.n_ring, "a free mailbox slot")
            return h, ln.sent % ln.n_ring

    def send_end(self, backend, dest, channel):
        _, ln = self._link(backend, dest, 'send', channel)
        with self.world.cond:
            ln.sent += 1
            self.world.cond.notify_all()

    def recv_begin(self, backend, src, channel):
        h, ln = self._link(backend, src, 'recv', channel)
        w = self.world
        if w.frozen:
            return h, 0
        with w.cond:     # ... and the copy out is enqueued behind the copy in
            w.wait(lambda: ln.sent > ln.received, f"the send of rank {src} to rank {self.rank}")
            return h, ln.received % ln.n_ring

    def recv_end(self, backend, src, channel):
        _, ln = self._link(backend, src, 'recv', channel)
        with self.world.cond:
            ln.received += 1
            self.world.cond.notify_all()

    def cycle_slot(self, backend, peer, direction, channel, ordinal):
        """(handle, slot) of the ordinal-th message of a planned cycle on a link"""
        h, ln = self._link(backend, peer, direction, channel)
        if ordinal >= ln.n_cycle:
            raise RuntimeError("a planned cycle with more messages on a link than the loopback mailbox has cycle slots")
        return h, ln.n_ring + ordinal

    def cycle_begin(self, backend, sends, recvs):
        """before a planned cycle is issued (launch by launch or as one graph): its receives after the sender's cycle, its
        sends after the receiver has issued the cycle before"""
        w = self.world
        if w.frozen:
            return
        with w.cond:
            for (peer, ch) in recvs:
                _, ln = self._link(backend, peer, 'recv', ch)
                w.wait(lambda ln=ln: ln.cyc_sent > ln.cyc_received and ln.sent == ln.received, f"the cycle of rank {peer}")
            for (peer, ch) in sends:
                _, ln = self._link(backend, peer, 'send', ch)
                w.wait(lambda ln=ln: ln.cyc_sent == ln.cyc_received, f"rank {peer} taking the cycle before")

    def cycle_end(self, backend, sends, recvs):
        w = self.world
        for n in sends.values():
            self.stats["messages"] += n
            self.stats["device_messages"] += n
        if w.frozen:
            return
        with w.cond:
            for (peer, ch) in sends:
                self._link(backend, peer, 'send', ch)[1].cyc_sent += 1
            for (peer, ch) in recvs:
                self._link(backend, peer, 'recv', ch)[1].cyc_received += 1
            w.cond.notify_all()


def run_loopback_ranks(size, target, timeout=600):
    """target(comm) on `size` threads that share the current GPU and stream; returns (world, per-rank results)"""
    import threading
    world = LoopbackWorld(size)
    out, err = [None] * size, [None] * size

    def work(r):
        try:
            out[r] = target(world.comm(r))
        except BaseException as exc:   # noqa: BLE001 - handed to the caller
            err[r] = exc
            world.barrier_obj.abort()
    threads = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(size)]
    for t in threads:
        t.start()
    import time
    deadline = time.time() + timeout
    for t in threads:
        t.join(max(0.0, deadline - time.time()))
    if any(e is not None for e in err):
        raise [e for e in err if e is not None][0]
    if any(t.is_alive() for t in threads):
        raise RuntimeError(f"loopback ranks still running after {timeout} s")
    return world, out


_default_comm = {}   # id(process group) -> (the group, its time communicator): the per-neighbour links and the side group are
                     # created once per group and shared by every solver instance of the process (new_group is collective and
                     # never freed)


def _drop_default_comms():
    """the process group they were made for is gone: two-rank communicators are freed before the entries are forgotten"""
    for _, comm in list(_default_comm.values()):
        close = getattr(comm, "close", None)
        if close is not None:
            try:
                close()
            except Exception:      # noqa: BLE001 - a communicator of a destroyed process group may refuse; nothing to do about it
                pass
    _default_comm.clear()


def resolve_comm(comm_time):
    """``comm_time=None`` -> the default process group when torch.distributed is initialised, else serial: RcclTimeComm
    (ghost rows under the C ABI) on the "nccl" backend, TorchTimeComm (rows through torch.distributed, staged through the host
    for device slabs) on "gloo"."""
    if comm_time is None:
        if not (dist.is_available() and dist.is_initialized()):
            _drop_default_comms()
            return SerialComm()
        world = dist.group.WORLD
        held = _default_comm.get(id(world))
        # (the id of a destroyed group may be reused: the entry must hold THIS group object)
        if held is None or held[0] is not world or held[1].backend != dist.get_backend():
            _drop_default_comms()
            device = dist.get_backend() == "nccl" and options.exchange != "torch"
            held = _default_comm[id(world)] = (world, RcclTimeComm() if device else TorchTimeComm())
        return held[1]      # (its counters are per process; Mgrit.solve() reports its own share as mg.exchange_stats)
    if hasattr(comm_time, "exchange") and hasattr(comm_time, "Get_rank"):
        return comm_time
    raise Exception("comm_time must be None or a pymgrit_amd TimeComm (mpi4py communicators are not used on MI355X: "
                    "ghost exchange runs over torch.distributed / RCCL)")
