"""ctypes binding of ``libabcdez_hip.so`` (C ABI: include/abcdez_hip.h).

Loading fails loudly: there is no CPU fallback for the hot path.
"""
from __future__ import annotations

import ctypes as C
import os

from .model import Model

_LIB = None
# ABCDEZ_HIP_LIB: another build of the same library (the Julia shim reads the same variable) -- used for same-box A/B runs
LIB_PATH = os.environ.get("ABCDEZ_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libabcdez_hip.so")

_vp, _i64, _u32, _i32, _f64 = C.c_void_p, C.c_int64, C.c_uint32, C.c_int32, C.c_double
_pi64, _pf64 = C.POINTER(C.c_int64), C.POINTER(C.c_double)

# name -> argtypes (every function returns int status unless noted)
PROTOTYPES = {
    "abcdez_ctx_create": [C.POINTER(Model), C.c_int, C.POINTER(_vp)],
    "abcdez_ctx_create_user": [C.POINTER(Model), C.c_char_p, C.c_int, C.POINTER(_vp)],
    "abcdez_user_translation_unit": [C.POINTER(Model), C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t],
    "abcdez_ctx_destroy": [_vp],
    "abcdez_ctx_set_stream": [_vp, _vp],
    "abcdez_ctx_reserve": [_vp, _i64],
    "abcdez_ctx_set_lanes": [_vp, C.c_int],
    "abcdez_ctx_set_uniform_weights": [_vp, C.c_int],
    "abcdez_ctx_get_uniform_weights": [_vp, C.POINTER(_i32), _pi64],
    "abcdez_ctx_set_graphs": [_vp, C.c_int],
    "abcdez_graph_stats": [_vp, _pi64, _pi64, _pi64],
    "abcdez_ctx_get_layout": [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)],
    "abcdez_sync": [_vp],
    "abcdez_ctx_set_timing": [_vp, C.c_int],
    "abcdez_ctx_get_timing": [_vp, _pf64, _pi64, _pi64],
    "abcdez_dev_alloc": [C.c_size_t, C.POINTER(_vp)],
    "abcdez_dev_free": [_vp],
    "abcdez_memcpy_h2d": [_vp, _vp, _vp, C.c_size_t],
    "abcdez_memcpy_d2h": [_vp, _vp, _vp, C.c_size_t],
    "abcdez_host_alloc": [C.c_size_t, C.POINTER(_vp)],
    "abcdez_host_free": [_vp],
    "abcdez_memcpy_d2h_async": [_vp, _vp, _vp, C.c_size_t],
    "abcdez_init": [_vp, _vp, _vp, _vp, _i64, _i64],
    "abcdez_ctx_set_stamps": [_vp, _vp, _vp],
    "abcdez_blob_width": [_vp, _vp],
    "abcdez_blob_eval": [_vp, _vp, _vp, _i64, _vp, _vp],
    "abcdez_smc_partition": [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "abcdez_smc_prologue_packed": [_vp, _vp, _vp, _vp, _i64, _i64, _f64, _f64, _f64, _f64, _f64, _vp, _vp, _vp, _vp, _vp,
                                   _pf64, _pf64, _pf64, _pf64, _pi64, C.POINTER(_i32), _pf64, _pf64],
    "abcdez_smc_swarm_packed": [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _f64, _u32, _pi64, _pi64],
    "abcdez_smc_select_ahead": [_vp, _vp, _vp, _i64, _f64, _f64],
    "abcdez_smc_select_discard": [_vp],
    "abcdez_smc_select_stats": [_vp, _pi64, _pi64],
    "abcdez_mc_rank_stats": [_vp, _pi64, _pi64, _pi64],
    "abcdez_smc_sweeps_packed": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _f64, _f64, _f64, _u32, _i32, _f64, _pi64, _pi64,
                                 C.POINTER(_i32)],
    "abcdez_smc_generation_packed": [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _f64, _f64, _f64, _f64, _f64,
                                     _u32, _u32, _i32, _f64, _i32, _pf64, _pf64, _pf64, _pi64, C.POINTER(_i32), C.POINTER(_i32), _pf64, _pi64,
                                     _pi64, _pi64, C.POINTER(_i32), _pf64, _pf64],
    "abcdez_smc_replay_packed": [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _f64, _f64, _u32, _pi64, _pi64],
    "abcdez_smc_group_begin": [_vp, _i64, _f64],
    "abcdez_smc_group_replay": [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _f64, _f64, _u32],
    "abcdez_smc_group_publish": [_vp],
    "abcdez_smc_group_abort": [_vp],
    "abcdez_smc_group_end": [_vp, _pi64, _pi64, C.POINTER(_i32)],
    "abcdez_smc_resample_gather_packed": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "abcdez_packed_gather": [_vp, _vp, _i64, _vp, _vp, _vp],
    "abcdez_smc_reweight": [_vp, _vp, _vp, _vp, _i64, _f64, _f64, _pf64, _pf64, _pi64],
    "abcdez_get_ess": [_vp, _vp, _i64, _pf64],
    "abcdez_tree_sum": [_vp, _vp, _i64, _pf64],
    "abcdez_wsample_stratified": [_vp, _vp, _i64, _u32, _vp],
    "abcdez_quantile_alive": [_vp, _vp, _vp, _i64, _i64, _f64, _pf64, _pf64, _pf64],
    "abcdez_extrema": [_vp, _vp, _i64, _pf64, _pf64],
    "abcdez_count_gt": [_vp, _vp, _i64, _f64, _pi64],
    "abcdez_mc_rank_prepare": [_vp, _vp, _i64, _f64, _f64, _vp, _vp, _vp],
    "abcdez_mc_swarm": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _f64, _f64, _i64, _i64, _u32,
                        _pi64, _pi64, _pf64, _pf64],
    "abcdez_mc_generation": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _f64, _i64, _f64, _f64, _u32,
                             _pi64, _pi64, _pf64, _pf64],
    "abcdez_mc_draw_stats": [_vp, _pi64],
    "abcdez_mc_draws_by_rejection": [_i64, _i64],
    "abcdez_mc_generation_async": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _pf64, _i32, _f64, _f64,
                                   _u32, _pi64],
    "abcdez_mc_generation_wait": [_vp, _i64, _pi64, _pi64, _pf64, _pf64, _pf64],
    "abcdez_mc_generation_sharded_async": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _pf64, _i32, _f64, _f64,
                                           _u32, _pi64],
    "abcdez_comm_unique_id": [_vp, C.c_size_t],
    "abcdez_comm_init": [_vp, _vp, C.c_size_t, C.c_int, C.c_int],
    "abcdez_comm_init_host": [_vp, C.c_int, C.c_int, _vp, _vp, _vp],
    "abcdez_comm_kind": [_vp, C.POINTER(_i32)],
    "abcdez_comm_destroy": [_vp],
    "abcdez_comm_rank": [_vp, C.POINTER(_i32), C.POINTER(_i32)],
    "abcdez_comm_allgather": [_vp, _vp, _i64],
    "abcdez_comm_allreduce": [_vp, _vp, _i64, C.c_int, C.c_int],
    "abcdez_smc_sweeps_sharded": [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _f64, _u32, _i32, _f64,
                                  _pi64, _pi64, C.POINTER(_i32)],
    "abcdez_push_p": [_vp, _vp, _i64, _vp],
    "abcdez_math_eval": [_vp, C.c_int, _vp, _vp, _vp, _i64],
    "abcdez_draws_eval": [_vp, C.c_int, _i64, _i64, _i64, _u32, _f64, _f64, _vp, _vp, _vp, _vp],
}
MIN_VERSION = 610      # abcdez_comm_init_host, abz_model.ext (wrapper priors), abcdez_smc_generation_packed (include/abcdez_hip.h)
# the callbacks of abcdez_comm_init_host (include/abcdez_hip.h): in-place all-gather / all-reduce on HOST memory
HOST_ALLGATHER_FN = C.CFUNCTYPE(C.c_int, _vp, _vp, _i64)
HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, _vp, _vp, _i64, _i32, _i32)
# symbols with a non-status return type
OTHER_SYMBOLS = ("abcdez_version", "abcdez_rng_rounds", "abcdez_last_error", "abcdez_abi_layout")


class AbcdezError(RuntimeError):
    pass


def load():
    """Load the HIP library; raise if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise AbcdezError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C abcdez.jl_amd/csrc`).  The population loop has no CPU fallback."
        )
    # torch (device memory, streams, collectives) bundles its own HIP runtime: it has to be the first one loaded,
    # so that this library binds to the same runtime instead of bringing a second one into the process
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.abcdez_abi_layout.argtypes = [C.POINTER(_i32), C.c_int]
    lib.abcdez_abi_layout.restype = C.c_int
    lib.abcdez_version.restype = C.c_int
    lib.abcdez_rng_rounds.restype = C.c_int
    lib.abcdez_last_error.restype = C.c_char_p
    # a library older than the header this binding was written against would link (C has no signature check) and misread
    # arguments that were added since: refuse it
    if lib.abcdez_version() < MIN_VERSION:
        raise AbcdezError(f"{LIB_PATH} reports abcdez_version() = {lib.abcdez_version()}, this binding needs >= {MIN_VERSION}: "
                          "rebuild it (`make -C abcdez.jl_amd/csrc`)")
    _LIB = lib
    return lib


def check(lib, rc: int) -> None:
    if rc != 0:
        msg = lib.abcdez_last_error()
        raise AbcdezError(f"libabcdez_hip: status {rc}: {msg.decode('utf-8', 'replace') if msg else ''}")
