"""Loader for libuhc_amd.so (the HIP product library).  There is NO fallback: if the library is
missing or does not load, importing the compute path raises."""
from __future__ import annotations

import ctypes as C
import os

from ._capi import UhcCtrlDesc, UhcEnvDesc, UhcModelDesc

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UHC_LIB") or os.path.join(_HERE, "csrc", "libuhc_amd.so")
_lib = None

# every symbol include/uhc_amd.h declares
SYMBOLS = [
    "uhc_last_error", "uhc_abi_version", "uhc_build_flags", "uhc_model_create", "uhc_model_free", "uhc_model_nM",
    "uhc_batch_create", "uhc_batch_free", "uhc_batch_set_stream", "uhc_batch_sync", "uhc_batch_set_rfc_scale",
    "uhc_batch_field", "uhc_batch_set_state", "uhc_batch_simulate", "uhc_batch_forward", "uhc_batch_set_timing",
    "uhc_batch_kernel_time", "uhc_batch_set_overflow_mode", "uhc_batch_set_solver", "uhc_batch_set_kernel_path",
    "uhc_env_create", "uhc_env_free", "uhc_env_obs_dim", "uhc_env_field", "uhc_env_set_bank", "uhc_env_assign",
    "uhc_env_reset", "uhc_env_step", "uhc_env_set_next", "uhc_env_auto_reset", "uhc_env_set_clip_models", "uhc_env_set_end_reward", "uhc_env_set_obj_pose",
    "uhc_rollout_act", "uhc_rollout_record", "uhc_filter_scratch_doubles", "uhc_filter_push", "uhc_filter_apply",
]


class UhcError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime; it must be the one already resident when this library
    # resolves libamdhip64 (two runtimes in one process cannot both see the device)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise UhcError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950).  uhc_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    P = C.c_void_p
    L.uhc_last_error.restype = C.c_char_p
    L.uhc_abi_version.restype = C.c_int32
    if hasattr(L, "uhc_build_flags"):  # (ABI 10; tools/ A/B runs load older builds through UHC_LIB)
        L.uhc_build_flags.restype = C.c_int32
    L.uhc_model_create.argtypes = [C.POINTER(UhcModelDesc), C.POINTER(P)]
    L.uhc_model_free.argtypes = [P]
    L.uhc_model_free.restype = None
    L.uhc_model_nM.argtypes = [P]
    L.uhc_batch_create.argtypes = [C.POINTER(P), C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32,
                                   C.POINTER(UhcCtrlDesc), C.POINTER(P)]
    L.uhc_batch_free.argtypes = [P]
    L.uhc_batch_free.restype = None
    L.uhc_batch_set_stream.argtypes = [P, P]
    L.uhc_batch_sync.argtypes = [P]
    L.uhc_batch_set_rfc_scale.argtypes = [P, C.c_double]
    L.uhc_batch_field.argtypes = [P, C.c_int32, C.POINTER(P), C.POINTER(C.c_int64)]
    L.uhc_batch_set_state.argtypes = [P, P, C.c_int32, P, P]
    L.uhc_batch_simulate.argtypes = [P, P, P, P]
    L.uhc_batch_forward.argtypes = [P]
    L.uhc_batch_set_timing.argtypes = [P, C.c_int32]
    L.uhc_batch_set_overflow_mode.argtypes = [P, C.c_int32]
    L.uhc_batch_set_solver.argtypes = [P, C.c_int32, C.c_int32]
    L.uhc_batch_set_kernel_path.argtypes = [P, C.c_int32]
    L.uhc_batch_kernel_time.argtypes = [P, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.uhc_env_create.argtypes = [P, C.POINTER(UhcEnvDesc), C.POINTER(P)]
    L.uhc_env_free.argtypes = [P]
    L.uhc_env_free.restype = None
    L.uhc_env_obs_dim.argtypes = [P]
    L.uhc_env_field.argtypes = [P, C.c_int32, C.POINTER(P), C.POINTER(C.c_int64)]
    L.uhc_env_set_bank.argtypes = [P, P, C.c_int64, P, P, C.c_int32]
    L.uhc_env_assign.argtypes = [P, P, C.c_int32, P, P, P]
    L.uhc_env_reset.argtypes = [P, P, C.c_int32, P]
    L.uhc_env_set_next.argtypes = [P, P, C.c_int32, P, P, P, P]
    L.uhc_env_auto_reset.argtypes = [P]
    L.uhc_env_set_clip_models.argtypes = [P, P]
    L.uhc_env_set_obj_pose.argtypes = [P, P, C.c_int64]
    L.uhc_env_set_end_reward.argtypes = [P, C.c_double]
    L.uhc_env_step.argtypes = [P, P, P]
    I, D = C.c_int32, C.c_double
    L.uhc_rollout_act.argtypes = [P, I, I, P, I, I, P, P, P, P, P, P, P, P]
    L.uhc_rollout_record.argtypes = [P, I, I, P, P, P, P, P, P, I, I, P, P, P, P, P, P]
    L.uhc_filter_scratch_doubles.argtypes = [I, I]
    L.uhc_filter_scratch_doubles.restype = C.c_int64
    L.uhc_filter_push.argtypes = [P, P, I, I, P, P, P, P, P]
    L.uhc_filter_apply.argtypes = [P, P, I, I, P, P, P, I, I, D, P, P]
    _lib = L
    return L


def check(rc: int):
    if rc != 0:
        raise UhcError(lib().uhc_last_error().decode())
