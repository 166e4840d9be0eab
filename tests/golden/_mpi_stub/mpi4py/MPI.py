"""Stand-in for the subset of ``mpi4py.MPI`` the reference touches (Get_rank/Get_size/barrier/isend/recv/gather/bcast/
allgather/Split, Request.Waitall), used ONLY by tests/golden/make_golden.py to import and run the reference in the build
container (mpi4py is not installable there). By default a single rank. ``run_world(size, fn)`` runs ``fn`` on ``size``
threads that see themselves as the ranks of COMM_WORLD: sends are eager (buffered, matched by (source, dest, tag) in
order), collectives are rendezvous -- enough to execute the reference's true multi-rank code paths."""
import pickle
import queue
import threading

UNDEFINED = -32766
_tls = threading.local()


class Request:
    @staticmethod
    def Waitall(requests):
        return None


class _World:
    def __init__(self, size):
        self.size = size
        self.lock = threading.Lock()
        self.cond = threading.Condition(self.lock)
        self.queues = {}
        self.coll = {}
        self.bar = threading.Barrier(size)

    def q(self, src, dst, tag):
        with self.lock:
            return self.queues.setdefault((src, dst, tag), queue.Queue())


class Comm:
    """COMM_WORLD as seen by the calling thread (rank 0 of 1 outside run_world)"""

    def _w(self):
        return getattr(_tls, "world", None)

    @property
    def rank(self):
        return getattr(_tls, "rank", 0)

    @property
    def size(self):
        w = self._w()
        return w.size if w else 1

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def barrier(self):
        w = self._w()
        if w:
            w.bar.wait()

    Barrier = barrier

    def isend(self, obj, dest=0, tag=0):
        w = self._w()
        if not w:
            raise RuntimeError("single rank: isend is unreachable")
        w.q(self.rank, dest, tag).put(pickle.dumps(obj))
        return Request()

    def recv(self, source=0, tag=0):
        w = self._w()
        if not w:
            raise RuntimeError("single rank: recv is unreachable")
        return pickle.loads(w.q(source, self.rank, tag).get())

    def _collect(self, obj):
        """every rank contributes one object; returns the list in rank order"""
        w = self._w()
        if not w:
            return [obj]
        _tls.seq = getattr(_tls, "seq", 0) + 1
        with w.cond:
            slot = w.coll.setdefault(_tls.seq, {})
            slot[self.rank] = pickle.dumps(obj)
            w.cond.notify_all()
            w.cond.wait_for(lambda: len(slot) == w.size)
            return [pickle.loads(slot[r]) for r in range(w.size)]

    def gather(self, obj, root=0):
        out = self._collect(obj)
        return out if self.rank == root else None

    def allgather(self, obj):
        return self._collect(obj)

    def bcast(self, obj, root=0):
        return self._collect(obj)[root]

    def Split(self, color=0, key=0):
        """collective over the world: ranks with the same colour form a communicator (ordered by key); UNDEFINED -> COMM_NULL"""
        w = self._w()
        if not w:
            return _Single()
        _tls.splits = getattr(_tls, "splits", 0) + 1
        table = self._collect((color, key, self.rank))
        if color == UNDEFINED:
            return COMM_NULL
        members = [r for c, k, r in sorted((t for t in table if t[0] == color), key=lambda t: (t[1], t[2]))]
        return _Sub(w, (_tls.splits, color), members)


class _Single(Comm):
    def _w(self):
        return None

    @property
    def rank(self):
        return 0


class _Sub:
    """communicator over a subset of the world's ranks (result of Split): the collectives the reference uses on it"""

    def __init__(self, world, ident, members):
        self.world, self.ident, self.members = world, ident, members
        self.seq = 0

    def Get_rank(self):
        return self.members.index(_tls.rank)

    def Get_size(self):
        return len(self.members)

    def _collect(self, obj):
        w = self.world
        self.seq += 1
        key = (self.ident, self.seq)
        with w.cond:
            slot = w.coll.setdefault(key, {})
            slot[_tls.rank] = pickle.dumps(obj)
            w.cond.notify_all()
            w.cond.wait_for(lambda: len(slot) == len(self.members))
            return [pickle.loads(slot[r]) for r in self.members]

    def allgather(self, obj):
        return self._collect(obj)

    def bcast(self, obj, root=0):
        return self._collect(obj)[root]

    def barrier(self):
        self._collect(None)


COMM_WORLD = Comm()
COMM_NULL = None


def run_world(size, fn, timeout=300):
    """run fn(rank) on `size` threads as the ranks of COMM_WORLD; returns the results in rank order"""
    world = _World(size)
    out, err = [None] * size, [None] * size

    def work(r):
        _tls.rank, _tls.world, _tls.seq = r, world, 0
        try:
            out[r] = fn(r)
        except BaseException as exc:   # noqa: BLE001
            err[r] = exc
            world.bar.abort()
    threads = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout)
    for e in err:
        if e is not None:
            raise e
    if any(t.is_alive() for t in threads):
        raise RuntimeError("run_world: ranks still running (deadlock?)")
    return out
