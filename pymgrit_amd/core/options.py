"""Run-time options of the device path, in ONE place. Every option has a default, an environment variable that overrides the
default, and may be set from code, which overrides both::

    from pymgrit_amd.core.options import options
    options.coarse_solve = "sequential"     # ... or PYMGRIT_AMD_COARSE_SOLVE=sequential
    options.reset("coarse_solve")           # back to environment / default

The environment is looked at when an option is READ (not at import), so a test may set a variable around one solver. None of
the options changes what is computed beyond rounding: they select between equivalent forms and exist for measurements and for
the tests that compare the forms with each other (each names the test that needs it). Switches that selected a superseded
path no default configuration reaches were removed in round 4 (PLAN_SHAPE, PLAN_RESERVE, PLAN_CHAIN_US, H2D_CHAIN_CUS,
H2D_SWEEP_ALL, GEN_CHUNK, FUSE_CHUNK[_COARSE], NO_RANK_FUSION_UP / _COARSE, NO_RANK_GEN, NO_CPOINT_MIRROR, and the library's
MGRIT_HIP_FAS_TWO_PHASE, MGRIT_HIP_MAILBOX_MEMCPY, MGRIT_HIP_H2D_UNFUSED).
"""
import os

_UNSET = object()


class _Opt:
    def __init__(self, env, default, conv, doc):
        self.env, self.default, self.conv, self.doc = env, default, conv, doc

    def __set_name__(self, owner, name):
        self.name = name

    def __get__(self, obj, owner=None):
        if obj is None:
            return self
        val = obj.__dict__.get(self.name, _UNSET)
        if val is not _UNSET:
            return val
        raw = os.environ.get(self.env)
        if raw is None or raw == "":
            return self.default
        return self.conv(raw)

    def __set__(self, obj, value):
        obj.__dict__[self.name] = value


def _flag(raw):
    return raw not in ("0", "false", "False", "no")


def _opt_int(raw):
    return int(raw)


class Options:
    # ---- what the solver computes with (equivalent forms) ----------------------------------------------------------------
    coarse_solve = _Opt("PYMGRIT_AMD_COARSE_SOLVE", "auto", str,
                        "forward_solve on the coarsest level (reference mgrit.py:459-486): 'auto' = the time-parallel form wherever "
                        "the level qualifies (DESIGN.md 3.8), 'sequential' = always step by step (tests/test_hip_block_solve.py, "
                        "the tests of the chain kernels)")
    chain_plain = _Opt("MGRIT_HIP_CHAIN_PLAIN", False, _flag,
                       "step-by-step solve in the plain per-step form 3.3 everywhere, no overlapped chain 3.7 (also read by the "
                       "library; tests/test_hip_distributed.py)")
    no_level_fusion = _Opt("PYMGRIT_AMD_NO_LEVEL_FUSION", False, _flag,
                           "no whole-level passes: every sweep a launch of its own (tests/test_hip_level_fusion.py)")
    no_gen_passes = _Opt("PYMGRIT_AMD_NO_GEN_PASSES", False, _flag,
                         "no general whole-level passes (spatial coarsening, Advection1D, forcing rows); tests/test_hip_gen_passes.py")
    store_all_f = _Opt("PYMGRIT_AMD_STORE_ALL_F", False, _flag,
                       "the way up stores every F-point: no C-point storage (tests/test_hip_level_fusion.py)")
    no_pre_relax = _Opt("PYMGRIT_AMD_NO_PRE_RELAX", False, _flag,
                        "the residual's Phi of the last F-point is not kept for the next cycle's C-relaxation")
    fuse_up_coarse = _Opt("PYMGRIT_AMD_FUSE_UP_COARSE", None, _flag,
                          "coarser levels' way up through the interval pass: True / False; None = in planned cycles of >= 5 blocks")
    no_rank_fusion = _Opt("PYMGRIT_AMD_NO_RANK_FUSION", False, _flag,
                          "several ranks: no whole-level passes over a rank's complete intervals")
    no_aligned = _Opt("PYMGRIT_AMD_NO_ALIGNED", False, _flag,
                      "several ranks whose shares end on C-points: the generic rank path instead of the one-rank machinery")
    time_factor = _Opt("PYMGRIT_AMD_TIME_FACTOR", "auto", str,
                       "forcing time factors tau(t_i) of a level: 'auto' = the callable evaluated on the whole time grid where that "
                       "provably equals the point-by-point calls (backend_hip._time_factor), 'pointwise' = always one call per point")
    # ---- how a cycle is issued -------------------------------------------------------------------------------------------------
    plan_blocks = _Opt("PYMGRIT_AMD_PLAN_BLOCKS", None, _opt_int,
                       "blocks of time points of a planned cycle (core/cycle_plan.py); None = the backend's choice (1 = program order "
                       "wherever the coarsest-level solve is time-parallel)")
    plan_blocks_heat2d = _Opt("PYMGRIT_AMD_PLAN_BLOCKS_HEAT2D", None, _opt_int, "the same for Heat2D hierarchies solved step by step")
    plan_graph = _Opt("PYMGRIT_AMD_PLAN_GRAPH", "auto", str,
                      "'auto' = a cycle cut into blocks (two streams) is replayed as one hipGraph from its third execution on, a one-block "
                      "cycle is issued launch by launch (round 4: its replay measured no faster); '0' = always launch by launch, "
                      "'1' = one-block cycles are planned and replayed too, 'require' = likewise and a failed capture raises")
    plan_up_merge = _Opt("PYMGRIT_AMD_PLAN_UP_MERGE", None, _opt_int,
                         "blocks at the front of the time grid whose way up goes into one launch (None = half of them from 5 on)")
    pipeline_depth = _Opt("PYMGRIT_AMD_PIPELINE_DEPTH", 4, _opt_int,
                          "several ranks: iterations by which the stopping test lags (0 = the reference's check-every-cycle loop)")
    # ---- exchange --------------------------------------------------------------------------------------------------------------
    exchange = _Opt("PYMGRIT_AMD_EXCHANGE", "rccl", str,
                    "'rccl' = ghost rows as ncclSend / ncclRecv under the C ABI on the nccl backend, 'torch' = torch.distributed")
    exchange_timeout = _Opt("PYMGRIT_AMD_EXCHANGE_TIMEOUT", 120.0, float, "seconds after which a rank gives up waiting for a neighbour")
    loopback_timeout = _Opt("PYMGRIT_AMD_LOOPBACK_TIMEOUT", 120.0, float, "the same for ranks emulated as threads on one GPU")
    no_extra_groups = _Opt("PYMGRIT_AMD_NO_EXTRA_GROUPS", False, _flag,
                           "no side process groups: every collective on the main group (test hook, tests/test_distributed.py)")

    def reset(self, *names):
        """forget values set from code (all of them when no name is given)"""
        for name in (names or list(self.__dict__)):
            self.__dict__.pop(name, None)

    def describe(self):
        return {name: {"value": getattr(self, name), "env": opt.env, "default": opt.default, "doc": opt.doc}
                for name, opt in vars(type(self)).items() if isinstance(opt, _Opt)}

    def __repr__(self):
        return "Options(" + ", ".join(f"{k}={v['value']!r}" for k, v in sorted(self.describe().items())) + ")"


options = Options()
