"""Seeded random configurations of the 1-D steppers on the HIP path against the oracle (bit-exact states and residual norms):
sizes around the group boundaries, coarsening factors 2..8, two to four levels, V and F cycles, weights, cf_iter, forcing on /
off, non-uniform time grids. A wider sweep of the same generator: `python tests/test_hip_fuzz.py 200` (round 1, final build:
10000 of 10000 seeds bit-exact; round 2, with the whole-level passes, the closed-form correction and C-point storage: 20000 of
20000; end of round 2, with the one-workgroup chain, the non-temporal row accesses and the library-chosen chunks:
`python tests/test_hip_fuzz.py 40000 50000` -- 40000 of 40000 new seeds, 12000 of 12000 from seed 30000, and on the final build
30000 of 30000 from seed 100000; round 3, final build -- general whole-level passes, three-wave chains, the generator with
spatial coarsening on a third of the cases: 14000 of 14000 from seed 200000 before that extension, 8000 of 8000 from seed 300000
with it, 2061 of them with full weighting or the periodic transfer on at least one level pair, and 30000 of 30000 from seed
400000 at the very end of the round)."""
import sys

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def random_case(seed):
    rng = np.random.default_rng(seed)
    kind = "heat" if rng.random() < 0.7 else "advection"
    n_int = int(rng.choice([3, 17, 63, 64, 65, 1000, 1023, 1024, 1025, 2047, 2049, 3071, 4100, 8191, 16384]))
    nx = n_int + 2 if kind == "heat" else n_int + 1
    levels = int(rng.integers(2, 5))
    m = int(rng.integers(2, 9)) if levels < 4 else int(rng.integers(2, 4))
    nc = int(rng.integers(3, 6))
    nt = (nc - 1) * m ** (levels - 1) + 1
    while nt > 400 or (n_int > 4000 and nt > 80):
        levels -= 1
        nt = (nc - 1) * m ** (levels - 1) + 1
    levels = max(levels, 2)
    t0 = cases.lin(float(rng.uniform(0.01, 3.0)), nt)
    if rng.random() < 0.3:   # non-uniform fine grid (several coefficient sets)
        t0 = np.cumsum(np.concatenate(([0.0], rng.uniform(0.5, 1.5, nt - 1)))) * (t0[1] - t0[0])
    grids = [t0[::m ** l] for l in range(levels)]
    opts = dict(cycle_type='F' if rng.random() < 0.3 else 'V', cf_iter=int(rng.integers(0, 3)),
                weight_c=float(rng.choice([1.0, 1.0, 1.3, 0.8])), nested_iteration=bool(rng.random() < 0.5),
                max_iter=int(rng.integers(1, 4)), tol=0.0)
    forcing = bool(rng.random() < 0.7)
    # (round 3, drawn last so that everything above is what it was) spatial coarsening on some of the level pairs: full weighting /
    # linear interpolation for Heat1D (n_f = 2 n_c + 1), its periodic analogue for Advection1D (n_f = 2 n_c) -- the transfers the
    # general whole-level passes apply in registers, at sizes on both sides of the lane, wave and group boundaries
    transfer = None
    if rng.random() < 0.3:
        n = int(rng.choice([15, 31, 63, 127, 1023, 2047, 4095, 8191, 16383] if kind == "heat" else [16, 20, 40, 64, 100, 128, 1000, 1024, 2048, 2064, 4000, 4096, 8192, 16384]))
        ns, transfer = [n], []
        for _ in range(len(grids) - 1):
            halve = n >= 15 and (kind == "heat" or n % 2 == 0) and rng.random() < 0.7   # (ragged periodic sizes: coarse counts off the lane size)
            n = ((n - 1) // 2 if kind == "heat" else n // 2) if halve else n
            ns.append(n)
            transfer.append((1 if kind == "heat" else 2) if halve else 0)
        nx = [k + 2 if kind == "heat" else k + 1 for k in ns]
    # (round 5, drawn last again) a coarsest level long enough for the time-parallel forward solve (>= 64 steps): Heat1D on short time
    # intervals, where up to 256 sine modes survive a block; Advection1D on periodic grids that are and are not powers of two (radix-2
    # transforms / ordered sums); copy transfer, two or three levels
    if rng.random() < 0.12:
        m2 = int(rng.choice([2, 4]))
        lv2 = int(rng.choice([2, 3])) if m2 == 2 else 2
        nc2 = int(rng.integers(65, 100))
        nt2 = (nc2 - 1) * m2 ** (lv2 - 1) + 1
        n2 = int(rng.choice([63, 255, 1000, 1023, 2050] if kind == "heat" else [64, 100, 200, 256, 1001, 1024, 1500]))
        t2 = cases.lin(float(rng.choice([0.002, 0.02, 0.2, 2.0])), nt2)
        if rng.random() < 0.3:
            t2 = np.cumsum(np.concatenate(([0.0], rng.uniform(0.5, 1.5, nt2 - 1)))) * (t2[1] - t2[0])
        grids = [t2[::m2 ** l] for l in range(lv2)]
        nx, transfer = (n2 + 2 if kind == "heat" else n2 + 1), None
    return kind, nx, grids, forcing, opts, transfer


def run_case(oracle, seed):
    from test_hip_parity import make_pair
    kind, nx, grids, forcing, opts, transfer = random_case(seed)
    nested = opts.pop("nested_iteration")
    mg, op = make_pair(oracle, kind, nx, grids, transfer=transfer, forcing=forcing, nested_iteration=False, **opts)
    rng = np.random.default_rng(seed + 1)
    ref = op.state("u", 0)
    ref[1:] = rng.standard_normal(ref[1:].shape)       # random initial guess, the same on both sides
    mg.backend.set_natural("u", 0, ref)
    tag = (seed, kind, nx, [len(g) for g in grids], forcing, nested, opts, transfer)
    if nested:
        mg.nested_iteration(); op.nested_iteration()
        for lvl in range(len(grids)):
            assert np.array_equal(mg.backend.natural("u", lvl), op.state("u", lvl)), ("nested", lvl, tag)
    for it in range(opts["max_iter"]):
        mg.iteration(0, opts["cycle_type"], it, True); op.iteration(0, opts["cycle_type"], it, True)
        got, want = np.array(mg.compute_residual()), op.residual_norms()
        assert np.array_equal(got, want), ("residual", it, tag)
    for lvl in range(len(grids)):
        assert np.array_equal(mg.backend.natural("u", lvl), op.state("u", lvl)), ("u", lvl, tag)


@pytest.mark.parametrize("seed", list(range(40)) + [20005, 20007, 20022, 20042])      # (the last four: long coarsest levels, round 5)
def test_random_configurations_bit_exact(oracle, seed):
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    run_case(oracle, 1000 + seed)


if __name__ == "__main__":
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle as orc
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    bad = 0
    for s in range(n):
        try:
            run_case(orc, first + s)
        except AssertionError as exc:
            bad += 1
            print("FAIL", str(exc)[:300], flush=True)
        if (s + 1) % 500 == 0:
            print(f"... {s + 1} seeds, {bad} failures", flush=True)     # (a long run must keep talking: the GPU box kills silent jobs)
    print(f"{n - bad} of {n} random configurations bit-exact (seeds {first}..{first + n - 1})")
