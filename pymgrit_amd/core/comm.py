"""Time communicator: the rank-to-rank ghost exchange of the reference's mpi4py path
(``Mgrit.send/receive``, reference src/pymgrit/core/mgrit.py:693-713, pickled isend + blocking recv) re-expressed on
``torch.distributed`` -- one process per GPU, backend "nccl" (= RCCL over xGMI on MI355X) for device rows, "gloo" on
CPU. Payloads are either device/CPU tensors (sent in place, no pickling) or small picklable Python objects.
"""
import pickle

import torch
import torch.distributed as dist


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


class TorchTimeComm:
    """Nearest-owner point-to-point on a ``torch.distributed`` process group."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.backend = dist.get_backend(group)

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

    def exchange(self, send=None, recv=None):
        """One exchange point: optionally send ``(payload, dest)`` and/or receive ``(buffer_or_None, src)``.
        Tensors travel as they are (the receive buffer must be a tensor of the same shape); other payloads are
        pickled. Send and receive of one point are posted together (ncclGroup on RCCL), so a ring of ranks executing
        the same point cannot deadlock. Returns the received object (the filled tensor, or the unpickled payload)."""
        ops, result, staged = [], None, None
        if send is not None:
            payload, dest = send
            if torch.is_tensor(payload):
                if payload.is_cuda and self.backend != "nccl":  # e.g. gloo: device rows are staged through the host
                    payload = payload.cpu()
                ops.append(dist.P2POp(dist.isend, payload, self._global(dest), self.group))
            else:
                raw = torch.frombuffer(bytearray(pickle.dumps(payload)), dtype=torch.uint8).to(self._device())
                size = torch.tensor([raw.numel()], dtype=torch.int64, device=self._device())
                for w in [dist.isend(size, self._global(dest), group=self.group),
                          dist.isend(raw, self._global(dest), group=self.group)]:
                    ops.append(w)
        if recv is not None:
            buf, src = recv
            if torch.is_tensor(buf):
                if buf.is_cuda and self.backend != "nccl":
                    staged = torch.empty(buf.shape, dtype=buf.dtype, device="cpu")
                ops.append(dist.P2POp(dist.irecv, staged if staged is not None else buf, self._global(src), self.group))
                result = buf
            else:
                size = torch.zeros(1, dtype=torch.int64, device=self._device())
                dist.recv(size, self._global(src), group=self.group)
                raw = torch.empty(int(size.item()), dtype=torch.uint8, device=self._device())
                dist.recv(raw, self._global(src), group=self.group)
                result = pickle.loads(raw.cpu().numpy().tobytes())
        p2p = [o for o in ops if isinstance(o, dist.P2POp)]
        works = [o for o in ops if not isinstance(o, dist.P2POp)]
        if p2p:
            works += dist.batch_isend_irecv(p2p)
        for w in works:
            w.wait()
        if staged is not None:
            result.copy_(staged)
        return result


def resolve_comm(comm_time):
    """``comm_time=None`` -> the default process group when torch.distributed is initialised, else serial."""
    if comm_time is None:
        return TorchTimeComm() if dist.is_available() and dist.is_initialized() else SerialComm()
    if hasattr(comm_time, "exchange") and hasattr(comm_time, "Get_rank"):
        return comm_time
    raise Exception("comm_time must be None or a pymgrit_amd TimeComm (mpi4py communicators are not used on MI355X: "
                    "ghost exchange runs over torch.distributed / RCCL)")
