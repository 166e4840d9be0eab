"""pymgrit_amd: MI355X-native MGRIT relaxation engine behind PyMGRIT's Application / Vector / GridTransfer plugin API
and ``Mgrit(...).solve()`` surface (reference src/pymgrit/__init__.py:1-17 for the export list)."""
from .advection.advection_1d import Advection1D
from .advection.grid_transfer_advection import GridTransferAdvection
from .core.application import Application
from .core.grid_transfer import GridTransfer
from .core.grid_transfer_copy import GridTransferCopy
from .core.mgrit import Mgrit
from .core.simple_setup_problem import simple_setup_problem
from .core.vector import Vector
from .dahlquist.dahlquist import Dahlquist
from .heat.heat_1d import Heat1D
from .heat.heat_2d import Heat2D
from .heat.grid_transfer_heat import GridTransferHeat

__all__ = [s for s in dir() if not s.startswith('_')]
