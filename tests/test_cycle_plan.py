"""The cycle plan (pymgrit_amd/core/cycle_plan.py) on the CPU: a cycle recorded as (sweep, level, block) parts and replayed
in ANOTHER order must compute bit for bit what the program order computes. The plugin backend executes a plan serially, so
what is tested here are the dependency rules (which cells of u, v, g every part reads and writes): with a rule missing, the
scheduled order -- and, harsher, random topological orders of the same graph -- would give different numbers."""
import numpy as np
import pytest

import dist_worker
from pymgrit_amd import Mgrit
from pymgrit_amd.core import cycle_plan

CASES = ["heat_nx33_V_nested", "heat_nx33_F_nested", "heat_nx33_V_weight13", "heat_nx33_V_cf2", "heat_nx33_V_cflist",
         "heat_nx33_V_cf0", "heat_nx33_F_weight13_cf2", "heat_nx33_2lvl_m8", "heat_nx257_nt257", "dahlquist_3lvl",
         "dahlquist_F", "dahlquist_varying_coarsening", "heat_nx17_spatial_coarsening"]


def solve(case, blocks, shuffle=None):
    prob, tr, opts = dist_worker.build_problem(case, "plugin")
    mg = Mgrit(prob, transfer=tr, logging_lvl=30, plan_blocks=blocks, **opts)
    if shuffle is not None:
        rng = np.random.default_rng(shuffle)
        plan_of = mg._planned

        def shuffled(*a):
            plan = plan_of(*a)
            if plan is not None and not getattr(plan, "_shuffled", False):
                plan.order = random_topological(plan.nodes, rng)
                plan._shuffled = True
            return plan
        mg._planned = shuffled
    conv = mg.solve()["conv"]
    states = [np.array([np.asarray(v.pack(), dtype=np.float64).ravel() for v in mg.u[lvl]]) for lvl in range(mg.lvl_max)]
    return conv, states, mg


def random_topological(nodes, rng):
    remaining = [len(n.preds) for n in nodes]
    ready = [n.idx for n in nodes if not n.preds]
    order = []
    while ready:
        idx = ready.pop(int(rng.integers(len(ready))))
        order.append(nodes[idx])
        for s in nodes[idx].succs:
            remaining[s] -= 1
            if remaining[s] == 0:
                ready.append(s)
    assert len(order) == len(nodes)
    return order


def all_cases():
    import cases
    known = {**cases.solve_cases(), **cases.extra_cases()}
    return [c for c in CASES if c in known]


@pytest.mark.parametrize("case", all_cases())
def test_planned_cycle_equals_program_order(case):
    conv0, u0, _ = solve(case, 1)
    for blocks in (2, 3, 8):
        conv, u, mg = solve(case, blocks)
        assert any(p is not None and p.n_blocks > 1 for p in mg._plans.values()) or len(mg.t[-1]) < 3, "no plan was recorded"
        assert np.array_equal(conv, conv0), (case, blocks)
        for a, b in zip(u, u0):
            assert np.array_equal(a, b), (case, blocks)


@pytest.mark.parametrize("case", ["heat_nx33_V_nested", "heat_nx33_F_weight13_cf2", "heat_nx257_nt257", "dahlquist_varying_coarsening"])
def test_random_topological_orders(case):
    conv0, u0, _ = solve(case, 1)
    for seed in range(6):
        conv, u, _ = solve(case, 4, shuffle=seed)
        assert np.array_equal(conv, conv0), (case, seed)
        for a, b in zip(u, u0):
            assert np.array_equal(a, b), (case, seed)


def test_blocks_end_on_coarsest_points():
    t0 = np.linspace(0, 1, 65)
    k, maps = cycle_plan.block_maps([t0, t0[::4], t0[::16]], 4)
    assert k == 4 and maps[2].tolist() == [0, 0, 1, 2, 3]
    assert maps[0][16] == 0 and maps[0][17] == 1 and maps[1][4] == 0 and maps[1][5] == 1   # a block ends ON a coarsest point
    k, maps = cycle_plan.block_maps([t0, t0[::4], t0[::16]], 1)
    assert k == 1 and not maps[0].any()


def test_chain_parts_go_to_the_chain_stream_in_time_order():
    _, _, mg = solve("heat_nx257_nt257", 4)
    plan = next(p for p in mg._plans.values() if p is not None)
    chain = [n for n in plan.order if n.stream == "chain"]
    assert [n.chunk for n in chain] == sorted(n.chunk for n in chain) and len(chain) == plan.n_blocks
    pos = {n.idx: k for k, n in enumerate(plan.order)}
    for n in plan.nodes:
        assert all(pos[p] < pos[n.idx] for p in n.preds)      # the issue order is a topological order


def test_unplannable_cycle_runs_in_program_order():
    prob, tr, opts = dist_worker.build_problem("heat_nx33_V_nested", "plugin")

    class Odd(Mgrit):
        def c_relax(self, lvl):
            return super().c_relax(lvl)
    mg = Odd(prob, transfer=tr, logging_lvl=30, plan_blocks=4, **opts)
    assert mg.plan_blocks() == 1
