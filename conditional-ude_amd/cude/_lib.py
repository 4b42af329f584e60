"""ctypes binding of libcude_hip.so (the C ABI in include/cude.h).

The product path has no CPU fallback: if the HIP library cannot be loaded, or a call returns
a negative status, a CudeError is raised.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libcude_hip.so")

MODEL_CPEP = 0
MODEL_SUPP = 1
MODEL_CPEP_SYM = 2
COND_LOG = 0
COND_RAW = 1
UNIQUE_ID_BYTES = 128
XCHG_HANDLE_BYTES = 128


class CudeError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"cude status {status}: {message}")
        self.status = status


class Config(C.Structure):
    _fields_ = [("model", C.c_int32), ("n_state", C.c_int32), ("nn_in", C.c_int32),
                ("nn_width", C.c_int32), ("nn_depth", C.c_int32), ("n_steps", C.c_int32),
                ("device", C.c_int32), ("cond_space", C.c_int32), ("lambda_", C.c_double)]


_dp = C.POINTER(C.c_double)
_SIGNATURES = {
    "cude_last_error": (C.c_char_p, []),
    "cude_device_count": (C.c_int32, [C.POINTER(C.c_int32)]),
    "cude_n_params": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    "cude_create": (C.c_int32, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "cude_destroy": (C.c_int32, [C.c_void_p]),
    "cude_set_network": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "cude_network_info": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "cude_set_tolerances": (C.c_int32, [C.c_void_p, C.c_double, C.c_double]),
    "cude_adaptive_steps": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "cude_set_rng": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_int64]),
    "cude_set_param_mask": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "cude_rng_draws": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "cude_set_population_cpep": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "cude_set_population_supp": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "cude_set_params": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cude_get_params": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cude_forward": (C.c_int32, [C.c_void_p, _dp, C.c_void_p, C.c_void_p]),
    "cude_loss_grad": (C.c_int32, [C.c_void_p, _dp, C.c_void_p, C.c_void_p]),
    "cude_simulate": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "cude_multistart_forward": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cude_screen_candidates": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p]),
    "cude_multistart_loss_grad": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p]),
    "cude_mh_estep": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                  C.c_double, C.c_double, C.c_double, C.c_void_p]),
    "cude_profile_conditional": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "cude_train_restarts": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_int32,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cude_lbfgs_minimize": (C.c_int32, [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cude_lbfgs_minimize_sharded": (C.c_int32, [C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cude_fit_conditional": (C.c_int32, [C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_double,
                                         C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cude_mh_chain": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                  C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "cude_n_failed": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int64)]),
    "cude_adam_init": (C.c_int32, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double]),
    "cude_adam_step": (C.c_int32, [C.c_void_p, _dp]),
    "cude_synchronize": (C.c_int32, [C.c_void_p]),
    "cude_grad_occupancy": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32)]),
    "cude_adaptive_regroup": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "cude_adam_run": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p]),
    "cude_set_global_subjects": (C.c_int32, [C.c_void_p, C.c_double, C.c_void_p]),
    "cude_get_scale": (C.c_int32, [C.c_void_p, C.c_void_p, _dp]),
    "cude_loss_grad_partial": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cude_adam_apply": (C.c_int32, [C.c_void_p, C.c_void_p, _dp]),
    "cude_partial_buffer": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]),
    "cude_loss_grad_partial_device": (C.c_int32, [C.c_void_p]),
    "cude_adam_apply_device": (C.c_int32, [C.c_void_p, _dp]),
    "cude_kernel_time_ms": (C.c_int32, [C.c_void_p, _dp, C.POINTER(C.c_int64)]),
    "cude_kernel_time_stats": (C.c_int32, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                               C.POINTER(C.c_int64)]),
    "cude_set_kernel_timing": (C.c_int32, [C.c_void_p, C.c_int32]),
    "cude_comm_unique_id": (C.c_int32, [C.c_void_p]),
    "cude_comm_init": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "cude_comm_allreduce_host": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32]),
    "cude_comm_info": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "cude_xchg_export": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "cude_xchg_attach": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_double]),
    "cude_xchg_detach": (C.c_int32, [C.c_void_p]),
    "cude_xchg_enable": (C.c_int32, [C.c_void_p, C.c_int32]),
    "cude_xchg_info": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int32)]),
    "cude_set_option": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_char_p]),
}

_lib = None
STRICT = True        # every declared entry point must be there (tools/abl_*.py relax this for older library variants)


def exported_symbols():
    """Names the header declares; tests check the .so exports each of them."""
    return sorted(_SIGNATURES)


def load():
    """Load libcude_hip.so.  Raises CudeError when it is missing: there is no fallback path.

    When PyTorch is used in the same process it must be imported BEFORE this library is loaded so
    that both share one HIP runtime (same soname, libamdhip64.so.7); bench.py and the tests do so.
    """
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CudeError(-2, f"{LIB_PATH} not found: build it with `make -C conditional-ude_amd/csrc` "
                            "(or __graft_entry__.build()); there is no CPU fallback")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise CudeError(-2, f"cannot load {LIB_PATH}: {e}")
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            if STRICT:
                raise CudeError(-2, f"{LIB_PATH} does not export {name}: rebuild it (make -C conditional-ude_amd/csrc)")
            continue            # development A/B runs against an older build (tools/abl_bench.py)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status < 0:
        msg = load().cude_last_error()
        raise CudeError(status, msg.decode() if msg else "")
    return status
