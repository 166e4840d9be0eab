"""ctypes binding of libmgrit_hip.so (C ABI: include/mgrit_hip.h). The library is mandatory for device
applications: there is no CPU fallback -- a missing library or a missing GPU raises."""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("PYMGRIT_AMD_LIB") or os.path.join(_PKG, "lib", "libmgrit_hip.so")   # (override: experiment builds)

RELAX_F, RELAX_C, RELAX_CHAIN, RELAX_FC = 0, 1, 2, 3
FAS_WITH_F_RELAX, FAS_SKIP_COARSE_U = 1, 2
# MGRIT_HIP_T_*: kinds of timed entry-point calls (mgrit_hip_timing_drain)
TIMED_KINDS = ("relax_f", "relax_c", "chain", "residual", "jump", "restrict", "copy", "fas_rhs", "fas_fused",
               "error_correction", "interpolate", "ec_relax", "at_solve", "cf_fas", "ec_relax_res", "relax_fc", "f_fas", "exchange",
               "gen_down", "gen_up")
STEPPER_HEAT1D, STEPPER_ADVECTION1D = 1, 2
TRANSFER_COPY, TRANSFER_HEAT1D, TRANSFER_CALLER = 0, 1, 3
MAX_N = 16384
MAX_LINKS = 16
BLOCK_K, BLOCK_RMAX = 16, 256     # time-parallel forward solve
CHUNK_LONG = -1                   # mgrit_hip_intervals_create: the library's rule for the Heat1D whole-level passes (up to 16)

EXPORTS = {
    # name: (restype, argtypes)
    "mgrit_hip_abi_version": (C.c_int, []),
    "mgrit_hip_last_error": (C.c_char_p, []),
    "mgrit_hip_device_count": (C.c_int, []),
    "mgrit_hip_row_stride": (C.c_int, [C.c_int]),
    "mgrit_hip_row_position": (C.c_int, [C.c_int, C.c_int]),
    "mgrit_hip_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]),
    "mgrit_hip_destroy": (C.c_int, [C.c_void_p]),
    "mgrit_hip_sync": (C.c_int, [C.c_void_p]),
    "mgrit_hip_level_heat1d": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_double,
                                         C.c_int, C.c_void_p, C.c_void_p]),
    "mgrit_hip_level_heat1d_2pts": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_double,
                                              C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgrit_hip_level_forcing_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mgrit_hip_ec_runs_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "mgrit_hip_ec_relax": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mgrit_hip_at_solve": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mgrit_hip_level_advection1d": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_double]),
    "mgrit_hip_level_heat2d": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double,
                                         C.c_double, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mgrit_hip_heat2d_padded": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mgrit_hip_level_heat2d_forcing_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mgrit_hip_level_bind": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgrit_hip_chain_enable": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mgrit_hip_chain_state_len": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "mgrit_hip_chain_bind": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mgrit_hip_chain_resume": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mgrit_hip_block_solve_rank": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.POINTER(C.c_int)]),
    "mgrit_hip_block_solve_config": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mgrit_hip_block_solve": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mgrit_hip_block_solve_state": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "mgrit_hip_block_solve_form": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "mgrit_hip_level_transfer": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mgrit_hip_runs_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "mgrit_hip_pairs_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "mgrit_hip_relax": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double]),
    "mgrit_hip_residual": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "mgrit_hip_jump": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mgrit_hip_restrict_u": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mgrit_hip_copy_u_to_v": (C.c_int, [C.c_void_p, C.c_int]),
    "mgrit_hip_fas_rhs": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mgrit_hip_triples_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.POINTER(C.c_int)]),
    "mgrit_hip_fas_fused": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mgrit_hip_fas_fused_opts": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "mgrit_hip_copy_pairs_u_to_v": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mgrit_hip_error_correction": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mgrit_hip_interpolate": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mgrit_hip_residual_host": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "mgrit_hip_jump_host": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mgrit_hip_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "mgrit_hip_last_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "mgrit_hip_timing_drain": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "mgrit_hip_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mgrit_hip_set_reserve": (C.c_int, [C.c_void_p, C.c_int]),
    "mgrit_hip_chain_clock": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mgrit_hip_intervals_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int)]),
    "mgrit_hip_fas_fine_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "mgrit_hip_fas_coarse": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mgrit_hip_cf_fas": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "mgrit_hip_ec_relax_res": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "mgrit_hip_ec_relax_res_to": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mgrit_hip_gen_down": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mgrit_hip_gen_down_part": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "mgrit_hip_gen_up": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mgrit_hip_residual_fetch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    # ghost exchange under the ABI (links: RCCL two-rank communicators / mailboxes)
    "mgrit_hip_comm_unique_id": (C.c_int, [C.c_void_p]),
    "mgrit_hip_comm_init_rank": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int]),
    "mgrit_hip_comm_destroy": (C.c_int, [C.c_void_p, C.c_int]),
    "mgrit_hip_link_attach": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "mgrit_hip_mailbox_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "mgrit_hip_mailbox_destroy": (C.c_int, [C.c_void_p]),
    "mgrit_hip_link_mailbox": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mgrit_hip_link_stats": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "mgrit_hip_links_close": (C.c_int, [C.c_void_p, C.c_int]),
    "mgrit_hip_exchange": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mgrit_hip_send": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "mgrit_hip_recv": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "mgrit_hip_sync_bounded": (C.c_int, [C.c_void_p, C.c_double]),
    "mgrit_hip_error_correction_to": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "mgrit_hip_residual_stash": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mgrit_hip_cpoint_mirror": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "mgrit_hip_stream_create_masked": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "mgrit_hip_stream_destroy": (C.c_int, [C.c_void_p]),
}

_lib = None


class MgritHipError(RuntimeError):
    pass


def load():
    """dlopen libmgrit_hip.so and type every export; raises MgritHipError when the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MgritHipError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                                f"g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.mgrit_hip_abi_version() != 3:
            raise MgritHipError("libmgrit_hip.so ABI version mismatch")
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise MgritHipError(f"libmgrit_hip error {rc}: {load().mgrit_hip_last_error().decode()}")


def row_stride(n):
    """slab row stride (doubles) for n DOFs per time point: 1024*ceil(n/1024)"""
    return load().mgrit_hip_row_stride(int(n))


def row_permutation(n):
    """perm[j] = row position of natural spatial index j (the engine's lane-blocked storage order)"""
    import numpy as np
    j = np.arange(int(n), dtype=np.int64)
    perm = ((((j >> 10) * 8 + ((j & 15) >> 1)) * 64 + ((j >> 4) & 63)) << 1) + (j & 1)
    lib = load()
    for probe in (0, n // 2, n - 1):  # the formula must agree with the library
        assert perm[probe] == lib.mgrit_hip_row_position(int(n), int(probe))
    return perm
