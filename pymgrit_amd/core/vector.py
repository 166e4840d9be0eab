"""Vector plugin type: the state of ONE time point.

Same contract as the reference's ``pymgrit.core.vector.Vector`` (reference src/pymgrit/core/vector.py:19-151):
value semantics (``+ - *`` return new objects), ``norm``, ``clone*``, ``set_values/get_values`` and the
``pack/unpack`` pair used for ghost exchange. On the MI355X engine the states live in one HBM slab per level and
Vector objects are only materialised at the boundary (initial condition, ``mgrit.u[lvl][i]`` reads, output_fcn).
"""
from abc import ABC, abstractmethod


class Vector(ABC):
    """Abstract per-time-point state. Subclasses implement the eleven abstract methods below."""

    def __init__(self):
        pass

    # -- algebra -------------------------------------------------------------------------------------
    @abstractmethod
    def __add__(self, other): ...

    @abstractmethod
    def __sub__(self, other): ...

    @abstractmethod
    def __mul__(self, other): ...

    @abstractmethod
    def norm(self): ...

    # -- construction --------------------------------------------------------------------------------
    @abstractmethod
    def clone(self): ...

    @abstractmethod
    def clone_zero(self): ...

    @abstractmethod
    def clone_rand(self): ...

    # -- data access ---------------------------------------------------------------------------------
    @abstractmethod
    def set_values(self, *args, **kwargs): ...

    @abstractmethod
    def get_values(self, *args, **kwargs): ...

    # -- communication payload -----------------------------------------------------------------------
    @abstractmethod
    def pack(self, *args, **kwargs): ...

    @abstractmethod
    def unpack(self, *args, **kwargs): ...

    # derived operators (reference vector.py:113-151): all defined through __mul__/__add__/__sub__
    def __rmul__(self, other):
        return self * other

    def __imul__(self, other):
        return self * other

    def __iadd__(self, other):
        return self + other

    def __isub__(self, other):
        return self - other
