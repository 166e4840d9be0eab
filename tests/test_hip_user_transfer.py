"""GPU: a user's GridTransfer on the device path (reference core/grid_transfer.py:31-55 is an open interface). The engine keeps
every Phi on the device and applies such a transfer through its Python methods between the kernels (MGRIT_HIP_TRANSFER_CALLER,
mgrit_hip_fas_fine_rows / mgrit_hip_fas_coarse); with the library's arithmetic restated in Python the solve is bit-identical to
the one through the device transfer kernels."""
import numpy as np
import pytest

import dist_worker

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _solve(case, swap):
    from pymgrit_amd import Mgrit
    prob, tr, opts = dist_worker.build_problem(case, "hip")
    tr = swap(tr, len(prob))
    mg = Mgrit(prob, transfer=tr, logging_lvl=30, **opts)
    assert mg.backend.name == "hip"
    conv = mg.solve()["conv"]
    return conv, [mg.backend.natural("u", lvl) for lvl in range(mg.lvl_max)], mg


@pytest.mark.parametrize("case", ["heat_spatial_coarsening", "heat_spatial_coarsening_F", "heat_nx33_V_nested"])
def test_user_transfer_on_the_device_path(case):
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from pymgrit_amd import GridTransfer, GridTransferCopy, GridTransferHeat
    from pymgrit_amd.heat.heat_1d import VectorHeat1D
    calls = {"R": 0, "P": 0}

    class UserFullWeighting(GridTransfer):       # no device_transfer(): the reference's example class, written by a user
        def restriction(self, u):
            calls["R"] += 1
            f = u.get_values()
            n_c = (len(f) - 1) // 2
            out = VectorHeat1D(n_c)
            out.set_values(f[0:2 * n_c:2] * 1 / 4 + f[1:2 * n_c:2] * 1 / 2 + f[2:2 * n_c + 1:2] * 1 / 4)
            return out

        def interpolation(self, u):
            calls["P"] += 1
            c = u.get_values()
            vals = np.zeros(2 * len(c) + 1)
            vals[1::2] += c
            vals[2::2] += 1 / 2 * c
            vals[0:len(vals) - 1:2] += 1 / 2 * c
            out = VectorHeat1D(len(vals))
            out.set_values(vals)
            return out

    class CountingCopy(GridTransferCopy):        # a library transfer with an overridden method: runs through the override
        def restriction(self, u):
            calls["R"] += 1
            return super().restriction(u)

    def swap(tr, n_levels):
        tr = tr if tr is not None else [GridTransferCopy() for _ in range(n_levels - 1)]
        return [UserFullWeighting() if isinstance(t, GridTransferHeat) else CountingCopy() for t in tr]
    conv0, u0, mg0 = _solve(case, lambda tr, n: tr)
    conv1, u1, mg1 = _solve(case, swap)
    assert calls["R"] > 0
    assert not any(mg1.backend._device_transfer(lvl) for lvl in range(mg1.lvl_max - 1))
    assert np.array_equal(conv0, conv1), (conv0, conv1)
    for a, b in zip(u0, u1):
        assert np.array_equal(a, b)


def test_user_transfer_between_heat2d_levels():
    """a user's 2-D GridTransfer (the reference's interface is open, core/grid_transfer.py:31-55; it ships no 2-D transfer class):
    spatial coarsening 17 x 21 -> 9 x 11 by full weighting / bilinear interpolation written in Python. The device path applies it
    row by row between the kernels (MGRIT_HIP_TRANSFER_CALLER: mgrit_hip_fas_fine_rows / mgrit_hip_fas_coarse for Heat2D) while
    every Phi stays on the GPU; the yardstick is the same hierarchy on the plugin path (host steppers, the reference's semantics)."""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    import cases
    from pymgrit_amd import GridTransfer, Mgrit
    from pymgrit_amd.heat.heat_2d import Heat2D, VectorHeat2D
    calls = {"R": 0, "P": 0}

    class Coarsen2D(GridTransfer):
        def restriction(self, u):
            calls["R"] += 1
            f = np.asarray(u.get_values())
            nxc, nyc = (f.shape[0] + 1) // 2, (f.shape[1] + 1) // 2
            c = f[::2, ::2].copy()        # boundary values by injection, the interior by full weighting
            c[1:-1, 1:-1] = (4 * f[2:-2:2, 2:-2:2] + 2 * (f[1:-3:2, 2:-2:2] + f[3:-1:2, 2:-2:2] + f[2:-2:2, 1:-3:2] + f[2:-2:2, 3:-1:2]) +
                             f[1:-3:2, 1:-3:2] + f[1:-3:2, 3:-1:2] + f[3:-1:2, 1:-3:2] + f[3:-1:2, 3:-1:2]) / 16
            out = VectorHeat2D(nxc, nyc)
            out.set_values(c)
            return out

        def interpolation(self, u):
            calls["P"] += 1
            c = np.asarray(u.get_values())
            f = np.zeros((2 * c.shape[0] - 1, 2 * c.shape[1] - 1))
            f[::2, ::2] = c
            f[1::2, ::2] = (c[:-1, :] + c[1:, :]) / 2
            f[::2, 1::2] = (c[:, :-1] + c[:, 1:]) / 2
            f[1::2, 1::2] = (c[:-1, :-1] + c[1:, :-1] + c[:-1, 1:] + c[1:, 1:]) / 4
            out = VectorHeat2D(*f.shape)
            out.set_values(f)
            return out

    def hierarchy(host):
        t0 = np.linspace(0, 1, 33)
        prob = [Heat2D(x_start=0, x_end=cases.H2D_X_END, y_start=0, y_end=cases.H2D_Y_END, nx=nx, ny=ny, a=cases.H2D_A,
                       rhs_separable=[(cases.h2d_s0, lambda t: 1.0)], t_interval=t)
                for (nx, ny), t in (((17, 21), t0), ((9, 11), t0[::2]), ((9, 11), t0[::4]))]
        if host:
            for p in prob:
                p.device_stepper = lambda: None
        from pymgrit_amd import GridTransferCopy
        return prob, [Coarsen2D(), GridTransferCopy()]

    out = {}
    for host in (True, False):
        prob, tr = hierarchy(host)
        mg = Mgrit(prob, transfer=tr, logging_lvl=30, tol=1e-9, max_iter=8)
        assert (type(mg.backend).__name__ == "HipBackend") != host
        conv = mg.solve()["conv"]
        out[host] = (conv, np.array([np.asarray(mg.u[0][i].get_values()) for i in (5, 16, 32)]))
    assert calls["R"] > 0 and calls["P"] > 0
    (ch, uh), (cd, ud) = out[True], out[False]
    # (host steppers = another arithmetic of the same Phi: the fixtures' tolerance, 1e-9 relative above the rounding floor)
    assert len(ch) == len(cd) and np.all(np.abs(ch - cd) <= 1e-9 * ch + 2e-11), (ch, cd)
    assert np.max(np.abs(uh - ud)) <= 1e-11 * max(1.0, np.max(np.abs(uh)))
