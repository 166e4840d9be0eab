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
            if os.environ.get("PYMGRIT_AMD_NO_EXTRA_GROUPS"):   # switch (and test hook): main group only, on every rank
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


_default_comm = {}   # process group -> its TorchTimeComm: the per-neighbour links and the side group are created once per
                     # group and shared by every solver instance of the process (new_group is collective and never freed)


def resolve_comm(comm_time):
    """``comm_time=None`` -> the default process group when torch.distributed is initialised, else serial."""
    if comm_time is None:
        if not (dist.is_available() and dist.is_initialized()):
            return SerialComm()
        key = id(dist.group.WORLD)
        comm = _default_comm.get(key)
        if comm is None or comm.backend != dist.get_backend():
            comm = _default_comm[key] = TorchTimeComm()
        return comm
    if hasattr(comm_time, "exchange") and hasattr(comm_time, "Get_rank"):
        return comm_time
    raise Exception("comm_time must be None or a pymgrit_amd TimeComm (mpi4py communicators are not used on MI355X: "
                    "ghost exchange runs over torch.distributed / RCCL)")
