"""Liveness of the exchange protocol under RCCL-like semantics (rendezvous sends, one in-order stream per link): the
threads-based communicator of tests/mock_comm.py. gloo cannot show a deadlock of this kind because its sends are buffered.
Every schedule the solver uses is run: plain and pipelined loops, V and F cycles, aligned and unaligned splits (comm_front /
comm_back, ghost C- and F-points), ranks without coarse points, stops by tolerance (rollback) and by max_iter."""
import numpy as np
import pytest

import cases
from mock_comm import run_ranks
from pymgrit_amd import Dahlquist, Heat1D, Mgrit


def heat(nts, nx=17):
    prob = [Heat1D(x_start=0, x_end=1, nx=nx, a=1, init_cond=cases.init_cond, rhs_separable=[(cases.rhs_space, cases.rhs_time)],
                   t_start=0, t_stop=2, nt=nt) for nt in nts]
    for p in prob:
        p.device_stepper = lambda: None     # host steppers: this test is about the protocol, it runs on the CPU
    return prob


def dahlquist(ts):
    return [Dahlquist(t_interval=np.asarray(t)) for t in ts]


def solve(make, opts, size, depth, shared_stream=False):
    def target(comm):
        mg = Mgrit(make(), comm_time=comm, logging_lvl=30, pipeline_depth=depth, **opts)
        conv = mg.solve()["conv"]
        owned = [int(i) for i in mg.index_local[0]]
        return conv, [np.asarray(mg.u[0][i].pack(), dtype=np.float64).ravel() for i in owned], mg.pipeline_depth()
    res = run_ranks(size, target, shared_stream=shared_stream)
    conv = res[0][0]
    for r in res:
        assert np.array_equal(r[0], conv)
    return conv, np.array([v for r in res for v in r[1]]), res[0][2]


T129 = np.linspace(0, 5, 129)
L1 = T129[::16]
SCHEDULES = {
    "heat_V_aligned": (lambda: heat([65, 17, 5]), dict(tol=1e-9, max_iter=8)),
    "heat_V_maxiter": (lambda: heat([65, 17, 5]), dict(tol=1e-30, max_iter=3)),
    "heat_F_nonested": (lambda: heat([65, 17, 5]), dict(tol=1e-9, max_iter=8, cycle_type='F', nested_iteration=False)),
    "heat_2lvl_cf2": (lambda: heat([33, 9]), dict(tol=1e-9, max_iter=8, cf_iter=2)),
    "dahlquist_procs_without_points": (lambda: dahlquist([T129, L1, L1[::2], L1[::4]]), dict(tol=1e-10, max_iter=10)),
}


@pytest.mark.parametrize("size,depth", [(2, 0), (3, 0), (3, 2), (4, 4), (5, 1), (7, 3)])
@pytest.mark.parametrize("name", sorted(SCHEDULES))
def test_protocol_is_live_and_exact_under_rendezvous_sends(name, size, depth):
    make, opts = SCHEDULES[name]
    ref = Mgrit(make(), logging_lvl=30, **opts)
    conv1 = ref.solve()["conv"]
    u1 = np.array([np.asarray(ref.u[0][i].pack(), dtype=np.float64).ravel() for i in range(len(ref.t[0]))])
    conv, u, used_depth = solve(make, opts, size, depth)
    assert used_depth == depth
    assert np.array_equal(conv, conv1), (conv, conv1)
    assert np.array_equal(u, u1)


@pytest.mark.parametrize("size,depth", [(2, 0), (3, 3), (4, 4), (7, 2)])
@pytest.mark.parametrize("name", sorted(SCHEDULES))
def test_protocol_is_live_even_on_one_shared_stream_per_rank(name, size, depth):
    """every send and receive of a rank on ONE in-order stream: the ranks cannot run ahead of each other any more (a receive
    waits behind the rank's own unmatched sends) but nothing deadlocks -- data only flows towards higher ranks"""
    make, opts = SCHEDULES[name]
    ref = Mgrit(make(), logging_lvl=30, **opts)
    conv1 = ref.solve()["conv"]
    conv, u, _ = solve(make, opts, size, depth, shared_stream=True)
    assert np.array_equal(conv, conv1)
