"""Seeded random MGRIT hierarchies for the exchange-schedule fuzz (shared by tests/test_exchange_fuzz.py and the fixture
generator tests/golden/make_golden.py, which runs the reference on them)."""
import numpy as np

SEED0, N_CASES = 7000, 200


def random_case(seed):
    rng = np.random.default_rng(seed)
    levels = int(rng.integers(2, 6))
    nt = int(rng.integers(20, 140))
    t = np.linspace(0, float(rng.uniform(1, 6)), nt)
    grids = [t]
    for _ in range(levels - 1):
        prev = grids[-1]
        if len(prev) < 4:
            break
        if rng.random() < 0.5:     # uniform factor
            m = int(rng.integers(2, 5))
            nxt = prev[::m]
            if nxt[-1] != prev[-1] and rng.random() < 0.5:
                nxt = np.append(nxt, prev[-1])
        else:                      # arbitrary subset containing the first point
            keep = np.sort(rng.choice(np.arange(1, len(prev)), size=max(1, len(prev) // int(rng.integers(2, 4))), replace=False))
            nxt = prev[np.concatenate(([0], keep))]
        if len(nxt) < 2:
            break
        grids.append(nxt)
    crit = int(rng.choice([0, 0, 0, 1, 2, 3]))
    opts = dict(cycle_type='F' if rng.random() < 0.35 else 'V', cf_iter=int(rng.integers(0, 3)),
                weight_c=float(rng.choice([1.0, 1.0, 1.25])), nested_iteration=bool(rng.random() < 0.6),
                max_iter=int(rng.integers(1, 7)), tol=float(rng.choice([1e-30, 1e-4, 1e-8])), conv_crit=crit)
    size = int(rng.integers(2, 9))
    depth = int(rng.integers(0, 5))
    return grids, opts, size, depth
