"""Application plugin type: a time grid plus the time stepper Phi (``step``).

Contract of the reference's ``pymgrit.core.application.Application`` (reference src/pymgrit/core/application.py:17-107):
  * the grid comes from ``(t_start, t_stop, nt)`` (``np.linspace``) or from an explicit ``t_interval`` ndarray;
  * subclasses MUST set ``vector_template`` and ``vector_t_start`` in ``__init__`` (checked after construction,
    ``ValueError`` otherwise);
  * ``step(u_start, t_start, t_stop)`` returns a NEW vector and never mutates its input.

MI355X extension: an Application may additionally implement ``device_stepper()`` returning a plain dict that
describes Phi declaratively (see ``pymgrit_amd.heat.heat_1d.Heat1D``); such applications run on the HIP engine.
Applications without it are user plugins and run through their own Python ``step`` (plugin path).
"""
from abc import ABCMeta, abstractmethod

import numpy as np

from pymgrit_amd.core.vector import Vector


class MetaApplication(ABCMeta):
    """Post-construction check that the required attributes exist (reference application.py:17-29)."""
    required_attributes = []

    def __call__(cls, *args, **kwargs):
        instance = super().__call__(*args, **kwargs)
        missing = [a for a in instance.required_attributes if not hasattr(instance, a)]
        if missing:
            raise ValueError('required attribute (%s) not set' % missing[0])
        return instance


class Application(object, metaclass=MetaApplication):
    required_attributes = ['vector_template', 'vector_t_start']

    def __init__(self, t_start: float = None, t_stop: float = None, nt: int = None,
                 t_interval: np.ndarray = None) -> None:
        if t_interval is not None:
            if not isinstance(t_interval, np.ndarray):
                raise Exception('t_interval has the wrong type. Should be a numpy array')
            self.t = t_interval
            self.t_start, self.t_end, self.nt = t_interval[0], t_interval[-1], len(t_interval)
        else:
            if t_start is None or t_stop is None or nt is None:
                raise Exception('Specify an interval by t_start, t_stop and nt or by t_interval')
            self.t_start, self.t_end, self.nt = t_start, t_stop, nt
            self.t = np.linspace(self.t_start, self.t_end, nt)

    # vector_template (prototype of a per-time-point state) and vector_t_start (initial condition) are plain attributes
    # that every subclass assigns in its __init__; MetaApplication verifies their presence right after construction.

    @abstractmethod
    def step(self, u_start: Vector, t_start: float, t_stop: float) -> Vector:
        """Advance ``u_start`` from ``t_start`` to ``t_stop``; must return a new Vector."""
