"""RcclTimeComm.open_links without a GPU: two and four gloo ranks with a stand-in for libmgrit_hip.so whose
mgrit_hip_comm_init_rank BLOCKS until the other end of the link has called it with the same unique id (as ncclCommInitRank does):
every rank must create exactly the links it takes part in, both ends with the same id and opposite roles, in an order that
never lets two ranks wait for each other crosswise."""
import os
import subprocess
import sys
import tempfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

WORKER = r'''
import ctypes as C, json, os, sys, time
sys.path.insert(0, sys.argv[5]); sys.path.insert(0, os.path.join(sys.argv[5], "tests"))
import torch.distributed as dist
rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
from pymgrit_amd.core import comm as cm, hip_lib
store = dist.distributed_c10d._get_default_store()
log = []

class FakeLib:
    n = 0
    def mgrit_hip_comm_unique_id(self, buf):
        if FAIL_IDS == "all" or FAIL_IDS == str(rank):
            return -4                                # MGRIT_HIP_EUNSUPPORTED: no librccl on this rank
        FakeLib.n += 1
        raw = (f"id-{rank}-{FakeLib.n}".encode()).ljust(128, b".")
        C.memmove(buf, raw, 128)
        return 0
    def mgrit_hip_comm_init_rank(self, out_ref, uid, nranks, r):
        key = "link/" + bytes(uid).decode().strip(".")
        store.add(key, 1)                       # rendezvous of the two ends: blocks like ncclCommInitRank
        t0 = time.time()
        while int(store.add(key, 0)) < 2:
            if time.time() - t0 > 30:
                raise RuntimeError("deadlock in link creation: " + key)
            time.sleep(0.01)
        log.append(("init", bytes(uid).decode().strip("."), nranks, r))
        return 0
    def mgrit_hip_link_attach(self, h, handle, comm, peer):
        log.append(("attach", handle, peer))
        return 0
    def mgrit_hip_send(self, h, handle, slot, ptr, count):
        log.append(("send", handle, count))
        store.add(f"msg/{rank}/{handle}", 1)        # (one counter per sending end)
        return 0
    def mgrit_hip_recv(self, h, handle, slot, ptr, count):
        log.append(("recv", handle, count))
        if DEAD_LINK and handle == int(DEAD_LINK):
            log.append(("dead", handle))
        return 0
    def mgrit_hip_sync_bounded(self, h, timeout):
        return -2 if any(e[0] == "dead" for e in log[-8:]) else 0   # a receive nobody answers: the bounded wait gives up
    def mgrit_hip_links_close(self, h, abort):
        log.append(("links_close", abort))
        return 0
    def mgrit_hip_comm_destroy(self, comm, abort):
        log.append(("destroy", abort))
        return 0
FAIL_IDS = sys.argv[6] if len(sys.argv) > 6 else ""
DEAD_LINK = sys.argv[7] if len(sys.argv) > 7 and int(sys.argv[1]) == 1 else ""     # rank 1's receive on this handle never completes
hip_lib._lib = FakeLib()
hip_lib.check = lambda rc: None if rc == 0 else (_ for _ in ()).throw(RuntimeError(rc))

class Backend:
    h = None
# a chain of ranks (every rank talks to rank+1 on both channels) plus one link that skips a rank (ranks without coarse points)
need = []
if rank + 1 < world:
    need += [(rank, rank + 1, cm.CH_SWEEP), (rank, rank + 1, cm.CH_CHAIN)]
if rank > 0:
    need += [(rank - 1, rank, cm.CH_SWEEP), (rank - 1, rank, cm.CH_CHAIN)]
if world >= 4 and rank in (0, 2):
    need += [(0, 2, cm.CH_SWEEP)]
tc = cm.RcclTimeComm()
be = Backend()
if FAIL_IDS:
    err = tc.open_links_agreed(be, need)
    json.dump({"log": log, "error": err, "comms": len(tc._comms), "engines": len(tc._engines)}, open(os.path.join(out, f"r{rank}.json"), "w"))
else:
    tc.open_links(be, need)
    links = tc._engines[id(be)].handle
    ping, ping_err = None, None
    try:
        ping = tc.ping_links(be, 0, reps=3, timeout=1.0)
    except RuntimeError as exc:
        ping_err = str(exc)
    json.dump({"log": log, "links": {f"{k[0]}/{k[1]}/{k[2]}": v for k, v in links.items()}, "need": need, "ping": ping, "ping_error": ping_err},
              open(os.path.join(out, f"r{rank}.json"), "w"))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 4, 8])
def test_open_links_is_collective_and_crosswise_free(world):
    """(8 ranks: the layout of BASELINE configs[2] on one node -- 7 neighbour pairs x 2 channels = 14 communicators, plus the link
    that skips a rank)"""
    import json
    import socket
    out = tempfile.mkdtemp(prefix="rccl_links_")
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    script = os.path.join(out, "worker.py")
    open(script, "w").write(WORKER)
    procs = [subprocess.Popen([sys.executable, script, str(r), str(world), str(port), out, os.path.dirname(HERE)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate(timeout=120)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), logs
    res = [json.load(open(os.path.join(out, f"r{r}.json"))) for r in range(world)]
    ids = {}
    for r, d in enumerate(res):
        inits = [e for e in d["log"] if e[0] == "init"]
        assert len(inits) == len(d["need"])                      # one communicator per link this rank takes part in
        assert len(d["links"]) == len(d["need"])                 # ... and one engine handle each
        for _, uid, nranks, role in inits:
            assert nranks == 2
            ids.setdefault(uid, []).append((r, role))
    for uid, ends in ids.items():                                # both ends, the id's maker (the sender) as rank 0 of the pair
        assert sorted(role for _, role in ends) == [0, 1], (uid, ends)
        sender = int(uid.split("-")[1])
        assert dict(ends)[sender] == 0, (uid, ends)
    # the pre-flight ping: every rank holds the latency of EVERY link of the job, and on every link the sender's messages are
    # matched one to one by the receiver's (same number, same size)
    n_links = len(ids)
    for d in res:
        assert d["ping_error"] is None and len(d["ping"]) == n_links, (d["ping_error"], d["ping"])
    assert len({json.dumps(d["ping"], sort_keys=True) for d in res}) == 1
    sends = sum(1 for d in res for e in d["log"] if e[0] == "send")
    recvs = sum(1 for d in res for e in d["log"] if e[0] == "recv")
    assert sends == recvs == 3 * n_links
    assert {e[2] for d in res for e in d["log"] if e[0] in ("send", "recv")} == {8}


def test_a_dead_link_is_named_by_every_rank():
    """a receive that is never answered: the bounded wait gives up on that rank, and EVERY rank ends the pre-flight with the same
    error naming the pair (bench.py --gpus N turns it into a non-zero exit)"""
    import json
    import socket
    world = 3
    out = tempfile.mkdtemp(prefix="rccl_links_dead_")
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    script = os.path.join(out, "worker.py")
    open(script, "w").write(WORKER)
    # handle numbering on rank 1: its engine attaches [1->2 sweep, 1->2 chain, 0->1 sweep, 0->1 chain] = handles 0..3: kill 0->1 sweep
    procs = [subprocess.Popen([sys.executable, script, str(r), str(world), str(port), out, os.path.dirname(HERE), "", "2"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate(timeout=120)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), logs
    res = [json.load(open(os.path.join(out, f"r{r}.json"))) for r in range(world)]
    errs = {d["ping_error"] for d in res}
    assert len(errs) == 1 and None not in errs, errs
    assert "0->1" in errs.pop()


@pytest.mark.parametrize("who", ["all", "1"])
def test_links_that_cannot_be_made_are_given_up_by_every_rank(who):
    """no librccl (every rank), or one rank whose id cannot be made: every rank learns it with the ids -- nobody enters a
    communicator's rendezvous alone --, open_links_agreed returns the same error everywhere and leaves no link behind (the
    backend then sends its ghost rows through torch.distributed)"""
    import json
    import socket
    world = 3
    out = tempfile.mkdtemp(prefix="rccl_links_fail_")
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    script = os.path.join(out, "worker.py")
    open(script, "w").write(WORKER)
    procs = [subprocess.Popen([sys.executable, script, str(r), str(world), str(port), out, os.path.dirname(HERE), who],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate(timeout=120)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), logs
    res = [json.load(open(os.path.join(out, f"r{r}.json"))) for r in range(world)]
    assert len({d["error"] for d in res}) == 1 and res[0]["error"] and "rank" in res[0]["error"], [d["error"] for d in res]
    for d in res:
        assert d["comms"] == 0 and d["engines"] == 0
        assert not [e for e in d["log"] if e[0] == "init"]          # no rendezvous was entered
