"""Identity spatial transfer (reference src/pymgrit/core/grid_transfer_copy.py:12-47): both directions clone."""
from pymgrit_amd.core.grid_transfer import GridTransfer
from pymgrit_amd.core.vector import Vector

TRANSFER_COPY = 0  # MGRIT_HIP_TRANSFER_COPY


class GridTransferCopy(GridTransfer):
    def __init__(self):
        super().__init__()

    def restriction(self, u: Vector) -> Vector:
        return u.clone()

    def interpolation(self, u: Vector) -> Vector:
        return u.clone()

    def device_transfer(self) -> int:
        return TRANSFER_COPY
