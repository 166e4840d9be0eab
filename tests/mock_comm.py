"""An in-process stand-in for the time communicator with RCCL-like semantics, to test the exchange protocol for liveness
where gloo (eager, buffering sends) cannot: ranks are threads; a send is a RENDEZVOUS -- it completes only when the
receiver has posted the matching receive -- and the sends of one link are processed strictly in order by one worker
thread per link (the analogue of one stream per communicator). The caller never waits for its sends (as in
pymgrit_amd.core.comm.TorchTimeComm), receives block. Convergence values are gathered asynchronously."""
import queue
import threading


class _World:
    def __init__(self, size):
        self.size = size
        self.barrier = threading.Barrier(size)
        self.lock = threading.Lock()
        self.links = {}        # (src, dst) -> rendezvous channel
        self.gathers = {}      # sequence number -> {rank: values}
        self.objects = {}
        self.cond = threading.Condition(self.lock)

    def channel(self, src, dst):
        with self.lock:
            if (src, dst) not in self.links:
                self.links[(src, dst)] = _Channel()
            return self.links[(src, dst)]


class _Channel:
    """rendezvous hand-off: put() returns only after a get() has taken the item"""

    def __init__(self):
        self.slot = queue.Queue(maxsize=1)
        self.taken = queue.Queue(maxsize=1)

    def put(self, item):
        self.slot.put(item)
        self.taken.get()

    def get(self):
        item = self.slot.get()
        self.taken.put(True)
        return item


class _Gather:
    def __init__(self, world, seq):
        self.world, self.seq = world, seq

    def result(self):
        w = self.world
        with w.cond:
            w.cond.wait_for(lambda: len(w.gathers[self.seq]) == w.size)
            return [list(w.gathers[self.seq][r]) for r in range(w.size)]


class MockComm:
    """shared_stream=False: one in-order worker per link (the per-pair communicators of TorchTimeComm).
    shared_stream=True: ONE in-order worker per rank for all its sends AND receives (what a single shared RCCL stream
    would do): a receive queued behind an unmatched send cannot complete before that send -- the harshest ordering."""

    def __init__(self, world, rank, shared_stream=False):
        self.world, self.rank, self.size = world, rank, world.size
        self.shared_stream = shared_stream
        self._outbox = {}      # dest -> FIFO of payloads, drained by one worker thread per link
        self._stream = None    # shared_stream: FIFO of ("send", dest, payload) / ("recv", src, future)
        self._seq = 0
        self._objseq = 0

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def barrier(self):
        self.world.barrier.wait()

    def prepare(self):
        return None

    def drain(self):
        for q in self._outbox.values():
            q.join()
        if self._stream is not None:
            self._stream.join()

    def _stream_worker(self):
        while True:
            kind, peer, item = self._stream.get()
            if kind == "send":
                self.world.channel(self.rank, peer).put(item)
            else:
                item.put(self.world.channel(peer, self.rank).get())
            self._stream.task_done()

    def allgather_object(self, obj):
        w = self.world
        self._objseq += 1
        with w.cond:
            w.objects.setdefault(self._objseq, {})[self.rank] = obj
            w.cond.notify_all()
            w.cond.wait_for(lambda: len(w.objects[self._objseq]) == w.size)
            return [w.objects[self._objseq][r] for r in range(w.size)]

    def iallgather_floats(self, values, max_count):
        w = self.world
        self._seq += 1
        with w.cond:
            w.gathers.setdefault(self._seq, {})[self.rank] = [float(v) for v in values]
            w.cond.notify_all()
        return _Gather(w, self._seq)

    def _sender(self, dest):
        ch = self.world.channel(self.rank, dest)
        q = self._outbox[dest]
        while True:
            item = q.get()
            ch.put(item)          # blocks until the receiver has posted the matching receive
            q.task_done()

    def exchange(self, send=None, recv=None):
        result = None
        if self.shared_stream:
            import copy
            if self._stream is None:
                self._stream = queue.Queue()
                threading.Thread(target=self._stream_worker, daemon=True).start()
            if send is not None:
                self._stream.put(("send", send[1], copy.deepcopy(send[0])))
            if recv is not None:
                fut = queue.Queue(maxsize=1)
                self._stream.put(("recv", recv[1], fut))
                result = self._deliver(recv[0], fut.get())
            return result
        if send is not None:
            payload, dest = send
            if dest not in self._outbox:
                self._outbox[dest] = queue.Queue()
                threading.Thread(target=self._sender, args=(dest,), daemon=True).start()
            import copy
            self._outbox[dest].put(copy.deepcopy(payload))
        if recv is not None:
            buf, src = recv
            result = self._deliver(buf, self.world.channel(src, self.rank).get())
        return result

    @staticmethod
    def _deliver(buf, item):
        """a tensor receive buffer is filled in place, as TorchTimeComm.exchange does (the HIP backend hands out slab rows)"""
        if buf is not None and hasattr(buf, "copy_"):
            buf.copy_(item)
            return buf
        return item


def run_ranks(size, target, timeout=120, shared_stream=False):
    """run target(comm) on `size` threads; returns the per-rank results, raises on error or when a rank is still running
    after `timeout` seconds (deadlock)"""
    world = _World(size)
    out, err = [None] * size, [None] * size

    def work(r):
        try:
            out[r] = target(MockComm(world, r, shared_stream=shared_stream))
        except BaseException as exc:   # noqa: BLE001 - reported to the test below
            err[r] = exc
            world.barrier.abort()
    threads = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(size)]
    for t in threads:
        t.start()
    import time
    deadline = time.time() + timeout        # one deadline for the whole world, not one per rank
    for t in threads:
        t.join(max(0.0, deadline - time.time()))
    stuck = [r for r, t in enumerate(threads) if t.is_alive()]
    if any(e is not None for e in err):
        raise [e for e in err if e is not None][0]
    if stuck:
        raise AssertionError(f"ranks {stuck} still running after {timeout} s: the exchange protocol deadlocked")
    return out
