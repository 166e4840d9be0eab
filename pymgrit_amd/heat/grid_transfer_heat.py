"""Spatial coarsening for Heat1D by a factor of two: full-weighting restriction and linear interpolation on interior
grids (fine n = 2*coarse n + 1). Same arithmetic as the class defined in the reference's
examples/example_spatial_coarsening.py:18-82 (which ships it only as an example); available as a HIP kernel via
``device_transfer()``."""
import numpy as np

from pymgrit_amd.core.grid_transfer import GridTransfer
from pymgrit_amd.heat.heat_1d import VectorHeat1D

TRANSFER_HEAT1D = 1  # MGRIT_HIP_TRANSFER_HEAT1D


class GridTransferHeat(GridTransfer):
    def __init__(self):
        super().__init__()

    def restriction(self, u: VectorHeat1D) -> VectorHeat1D:
        fine = u.get_values()
        n_c = int((len(fine) - 1) / 2)
        out = VectorHeat1D(n_c)
        # c_i = f_{2i}/4 + f_{2i+1}/2 + f_{2i+2}/4, summed left to right
        out.set_values(fine[0:2 * n_c:2] * 1 / 4 + fine[1:2 * n_c:2] * 1 / 2 + fine[2:2 * n_c + 1:2] * 1 / 4)
        return out

    def interpolation(self, u: VectorHeat1D) -> VectorHeat1D:
        coarse = u.get_values()
        n_f = int(len(coarse) * 2 + 1)
        vals = np.zeros(n_f)
        vals[1::2] += coarse
        vals[2::2] += 1 / 2 * coarse            # contribution of c_{i} to f_{2i+2} is added first ...
        vals[0:n_f - 1:2] += 1 / 2 * coarse     # ... then that of c_{i} to f_{2i}
        out = VectorHeat1D(n_f)
        out.set_values(vals)
        return out

    def device_transfer(self) -> int:
        return TRANSFER_HEAT1D
