"""Test-only plugin Applications whose ``step`` calls the parity oracle's Phi (oracle/mgrit_oracle.c). They carry no
device description, so ``pymgrit_amd.Mgrit`` runs them on the plugin path: this is how the HOST logic (cycle control,
index sets, exchange schedule over torch.distributed/gloo) is exercised on CPU for the heat / advection hierarchies."""
import numpy as np

from pymgrit_amd.advection.advection_1d import VectorAdvection1D
from pymgrit_amd.core.application import Application
from pymgrit_amd.heat.heat_1d import VectorHeat1D


class OracleApp(Application):
    """level spec (tests/cases.py) -> plugin Application; Phi evaluated by the oracle for the step t[i-1] -> t[i]"""

    def __init__(self, orc, spec, variant=1):
        super().__init__(t_interval=np.asarray(spec["t"], dtype=np.float64))
        self.spec = spec
        self.vec = VectorHeat1D if spec["kind"] == "heat1d" else VectorAdvection1D
        self.n = int(spec["n"])
        self.vector_template = self.vec(self.n)
        self.vector_t_start = self.vec(self.n)
        self.vector_t_start.set_values(np.asarray(spec["u0"], dtype=np.float64).copy())
        self._orc, self._variant = orc, variant
        self._p = None

    def _problem(self):
        if self._p is None:
            self._p = self._orc.OracleProblem([self.spec], variant=self._variant)
        return self._p

    def __deepcopy__(self, memo):
        raise NotImplementedError

    def step(self, u_start, t_start, t_stop):
        i = int(np.searchsorted(self.t, t_stop))
        assert self.t[i] == t_stop and self.t[i - 1] == t_start, "oracle apps step between consecutive grid points"
        out = self.vec(self.n)
        out.set_values(self._problem().phi(0, i, u_start.get_values()))
        return out


class OracleApp2D(Application):
    """Heat2D level spec -> plugin Application whose step is the oracle's fast-diagonalisation Phi"""

    def __init__(self, orc, spec):
        from pymgrit_amd.heat.heat_2d import VectorHeat2D
        super().__init__(t_interval=np.asarray(spec["t"], dtype=np.float64))
        self.spec, self.nx, self.ny = spec, int(spec["nx"]), int(spec["ny"])
        self.vec = VectorHeat2D
        self.vector_template = VectorHeat2D(self.nx, self.ny)
        self.vector_t_start = VectorHeat2D(self.nx, self.ny)
        self.vector_t_start.set_values(np.asarray(spec["u0"], dtype=np.float64).reshape(self.nx, self.ny).copy())
        self._orc, self._p = orc, None

    def step(self, u_start, t_start, t_stop):
        if self._p is None:
            self._p = self._orc.OracleProblem([self.spec])
        i = int(np.searchsorted(self.t, t_stop))
        assert self.t[i] == t_stop and self.t[i - 1] == t_start
        out = self.vec(self.nx, self.ny)
        out.set_values(self._p.phi(0, i, np.asarray(u_start.get_values()).ravel()).reshape(self.nx, self.ny))
        return out
