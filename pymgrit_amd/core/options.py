"""Run-time options of the device path, in ONE place. Every field has an environment variable of the same meaning (read once,
at import) and may be set from code before a solver is constructed::

    from pymgrit_amd.core.options import options
    options.coarse_solve = "sequential"

None of them changes what is computed beyond rounding; they select between equivalent forms (and exist for measurements and
for tests that compare the forms with each other).
"""
import os


class Options:
    def __init__(self):
        # forward_solve on the coarsest level (reference src/pymgrit/core/mgrit.py:459-486):
        #   "auto"        the time-parallel form wherever the level qualifies (Heat1D, >= 64 steps, at most 64 sine
        #                 modes survive a block of 16 steps), else step by step
        #   "sequential"  always step by step (the chain kernels of csrc/mgrit_hip_chain.inc)
        self.coarse_solve = os.environ.get("PYMGRIT_AMD_COARSE_SOLVE", "auto")

    def __repr__(self):
        return "Options(" + ", ".join(f"{k}={v!r}" for k, v in sorted(self.__dict__.items())) + ")"


options = Options()
