"""GridTransfer plugin type: spatial restriction / interpolation between two consecutive time levels
(reference src/pymgrit/core/grid_transfer.py:14-55). One instance per level pair; ``len(problem) == len(transfer)+1``.

MI355X extension: a transfer may expose ``device_transfer()`` -> int kind (``MGRIT_HIP_TRANSFER_*`` of
include/mgrit_hip.h) so the sweep runs as a HIP kernel on the slabs.
"""
from abc import ABC, abstractmethod

from pymgrit_amd.core.vector import Vector


class GridTransfer(ABC):
    def __init__(self):
        pass

    @abstractmethod
    def restriction(self, u: Vector) -> Vector:
        """Map the state of a time point to the same time point on the next coarser level."""

    @abstractmethod
    def interpolation(self, u: Vector) -> Vector:
        """Map the state of a time point to the same time point on the next finer level."""
