"""Size-1 stand-in for ``mpi4py`` used ONLY by tests/golden/make_golden.py to import the
reference in the build container (mpi4py is not installable there). Never shipped to the GPU box."""
from . import MPI  # noqa: F401
