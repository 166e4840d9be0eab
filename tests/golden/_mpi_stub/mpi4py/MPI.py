"""Single-rank stand-in for the subset of ``mpi4py.MPI`` the reference touches
(Get_rank/Get_size/barrier/isend/recv/gather/bcast/allgather/Split, Request.Waitall)."""

UNDEFINED = -32766


class Request:
    @staticmethod
    def Waitall(requests):
        return None


class Comm:
    def __init__(self, rank=0, size=1):
        self.rank = rank
        self.size = size

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def barrier(self):
        return None

    Barrier = barrier

    def isend(self, obj, dest=0, tag=0):
        raise RuntimeError("size-1 stand-in: isend is unreachable")

    def recv(self, source=0, tag=0):
        raise RuntimeError("size-1 stand-in: recv is unreachable")

    def gather(self, obj, root=0):
        return [obj]

    def allgather(self, obj):
        return [obj]

    def bcast(self, obj, root=0):
        return obj

    def Split(self, color=0, key=0):
        return Comm()


COMM_WORLD = Comm()
COMM_NULL = None
