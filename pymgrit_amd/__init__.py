"""pymgrit_amd -- MI355X-native MGRIT relaxation engine behind PyMGRIT's plugin API.

Same public names as the reference package for everything on the hot path (``Mgrit``, ``AtMgrit``, ``Application``, ``Vector``,
``GridTransfer``, ``GridTransferCopy``, ``simple_setup_problem``, ``Dahlquist``, ``Heat1D``, ``Heat1DBDF1``, ``Heat1DBDF2``,
``Heat2D``, ``Advection1D``) plus the two spatial-coarsening transfers that run as HIP kernels.
"""
from pymgrit_amd.core.application import Application
from pymgrit_amd.core.vector import Vector
from pymgrit_amd.core.grid_transfer import GridTransfer
from pymgrit_amd.core.grid_transfer_copy import GridTransferCopy
from pymgrit_amd.core.simple_setup_problem import simple_setup_problem
from pymgrit_amd.core.mgrit import Mgrit
from pymgrit_amd.core.at_mgrit import AtMgrit

from pymgrit_amd.dahlquist.dahlquist import Dahlquist
from pymgrit_amd.heat.heat_1d import Heat1D
from pymgrit_amd.heat.heat_1d_2pts_bdf1 import Heat1DBDF1
from pymgrit_amd.heat.heat_1d_2pts_bdf2 import Heat1DBDF2
from pymgrit_amd.heat.heat_2d import Heat2D
from pymgrit_amd.heat.grid_transfer_heat import GridTransferHeat
from pymgrit_amd.advection.advection_1d import Advection1D
from pymgrit_amd.advection.grid_transfer_advection import GridTransferAdvection



def elementwise(f):
    """Declare a forcing time factor tau(t) elementwise: called with an array of times it returns, entry by entry, exactly what
    it returns for each time alone (no dependence on position, length, other entries or earlier calls). The device path then
    evaluates it once per level instead of once per time point (core/backend_hip._time_factor)."""
    f.elementwise = True
    return f


__all__ = ["elementwise", "Application", "Vector", "GridTransfer", "GridTransferCopy", "simple_setup_problem", "Mgrit", "AtMgrit", "Dahlquist",
           "Heat1D", "Heat1DBDF1", "Heat1DBDF2", "Heat2D", "GridTransferHeat", "Advection1D", "GridTransferAdvection"]
