"""ctypes binding of libepi.so (the C ABI declared in include/epi.h).

The library is the product: there is no Python or CPU implementation behind these
calls.  If libepi.so is missing, or no HIP device is usable, the calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EPI_LIB") or os.path.join(_HERE, "libepi.so")   # EPI_LIB: another build of the same library

EPI_OK, EPI_ERR_BAD_ARGUMENT, EPI_ERR_DEVICE, EPI_ERR_LOCAL_MASS, EPI_ERR_WORLD_MASS, EPI_ERR_ASSERT_STATE, \
    EPI_ERR_INT_RANGE, EPI_ERR_UNSUPPORTED = range(8)

STATUS_NAMES = {0: "EPI_OK", 1: "EPI_ERR_BAD_ARGUMENT", 2: "EPI_ERR_DEVICE", 3: "EPI_ERR_LOCAL_MASS",
                4: "EPI_ERR_WORLD_MASS", 5: "EPI_ERR_ASSERT_STATE", 6: "EPI_ERR_INT_RANGE", 7: "EPI_ERR_UNSUPPORTED"}


class EpiError(RuntimeError):
    """A non-zero epi_status.  `.status` holds the code; the text is epi_last_error()."""

    def __init__(self, status, message):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), message))
        self.status = status
        self.message = message


class IllegalArgumentException(EpiError):
    """Raised where the reference's `require(...)` would throw (WorldToy.scala:94,144; Validation.scala:7-10)."""


class Profile(C.Structure):
    _fields_ = [(k, C.c_double) for k in (
        "beta_mul", "ifr_mul", "noise", "population_density", "gdp_per_capita",
        "healthcare_capacity", "age_elderly", "connectivity", "econ_resilience")]


class EnvParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("base_beta", "base_gamma", "base_ifr", "noise", "reward_clip", "risk_factor")]


class CState(C.Structure):
    _fields_ = [("s", C.c_double), ("i", C.c_double), ("r", C.c_double), ("d", C.c_double), ("v", C.c_double),
                ("hosp_cap", C.c_double), ("t", C.c_int32), ("t_max", C.c_int32), ("seed", C.c_int32),
                ("pad_", C.c_int32)]


class CHyper(C.Structure):
    _fields_ = [("epsilon_start", C.c_double), ("epsilon_end", C.c_double), ("epsilon_decay_steps", C.c_int32),
                ("hidden", C.c_int32), ("clip_q", C.c_double), ("reward_clip", C.c_double),
                ("batch_size", C.c_int32), ("learn_every", C.c_int32)]


_PD = C.POINTER(C.c_double)
_PI = C.POINTER(C.c_int32)
_PU8 = C.POINTER(C.c_uint8)


class LearnHyper(C.Structure):
    _fields_ = [("gamma", C.c_double), ("lr", C.c_double), ("clip_q", C.c_double), ("grad_clip", C.c_double),
                ("ema_alpha", C.c_double), ("hidden", C.c_int32), ("replay_capacity", C.c_int32),
                ("target_update_every", C.c_int32), ("max_need", C.c_int32)]


class GridDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("pop", _PI), ("conn", _PD), ("csr_rowptr", _PI), ("csr_col", _PI),
                ("csr_val", _PD), ("seed_idx", C.c_int32), ("heterogeneous", C.c_int32), ("base", CState),
                ("prof", C.POINTER(Profile)), ("hp", CHyper), ("steps", C.c_int32), ("n_beta", C.c_int32),
                ("betas", _PD), ("n_travel", C.c_int32), ("travel", _PD), ("t_arrive", _PD), ("has_arrive", _PU8),
                ("t_peak", _PD), ("has_peak", _PU8), ("device", C.c_int32), ("pad_", C.c_int32),
                ("learn", C.POINTER(LearnHyper)), ("learn_intensity", _PI)]


class WorldDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("batch", C.c_int32), ("pop", _PI), ("conn", _PD), ("csr_rowptr", _PI),
                ("csr_col", _PI), ("csr_val", _PD), ("seed_idx", C.c_int32), ("validate", C.c_int32),
                ("base", CState), ("env", C.POINTER(EnvParams)), ("prof", C.POINTER(Profile)), ("env_seeds", _PI),
                ("hp", CHyper), ("device", C.c_int32), ("pad_", C.c_int32)]


# Every symbol include/epi.h declares; tests check that the library exports all of them.
EXPORTED_SYMBOLS = [
    "epi_version", "epi_last_error", "epi_last_status", "epi_device_count", "epi_set_device", "epi_alloc_int", "epi_alloc_int_batch",
    "epi_env_step_batch", "epi_scale_row", "epi_gravity_matrix", "epi_gravity_csr", "epi_java_gaussians", "epi_java_gaussian_table", "epi_agent_actions", "epi_mlp_init",
    "epi_mlp_greedy_batch", "epi_grid_create", "epi_grid_run", "epi_grid_run_strided", "epi_grid_run_noalloc", "epi_grid_status", "epi_grid_destroy",
    "epi_grid_argmin", "epi_grid_screen_stats", "epi_world_screen_stats", "epi_calibrate_grid", "epi_grid_multi_create", "epi_grid_multi_run", "epi_grid_multi_ranks",
    "epi_grid_multi_set_timing", "epi_grid_multi_times", "epi_grid_multi_peer_access", "epi_grid_multi_inject_failure",
    "epi_grid_multi_destroy", "epi_calibrate_grid_multi", "epi_grid_set_timing", "epi_grid_kernel_time", "epi_world_create", "epi_world_step", "epi_world_step_greedy", "epi_world_rollout_greedy", "epi_world_keep_transitions",
    "epi_world_set_mlp", "epi_world_get_states", "epi_world_set_states", "epi_world_reset", "epi_world_last_flows",
    "epi_world_set_timing", "epi_world_kernel_time", "epi_world_result_buffers", "epi_world_step_packed", "epi_world_packed_buffer", "epi_world_destroy",
    "epi_agents_create", "epi_agents_destroy", "epi_agents_observe", "epi_agents_observe_world", "epi_agents_replay_size", "epi_agents_sample",
    "epi_agents_learn", "epi_agents_q_values", "epi_agents_get_weights", "epi_agents_set_weights",
    "epi_agents_set_timing", "epi_agents_kernel_time",
]

_lib = None


def lib():
    """Loads libepi.so (built by `make -C epidemicsimulator_amd/csrc` or __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `make -C %s` (hipcc, gfx950). "
                              "There is no fallback implementation." % (LIB_PATH, os.path.join(_HERE, "csrc")))
        L = C.CDLL(LIB_PATH)
        for name in EXPORTED_SYMBOLS:
            getattr(L, name).restype = C.c_int32
        L.epi_grid_create.restype = C.c_void_p
        L.epi_world_create.restype = C.c_void_p
        L.epi_last_error.argtypes = [C.c_char_p, C.c_size_t]
        L.epi_grid_run.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                   C.c_void_p]
        L.epi_grid_run_strided.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int32, C.c_void_p]
        L.epi_grid_run_noalloc.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p]
        L.epi_grid_status.argtypes = [C.c_void_p]
        L.epi_calibrate_grid.argtypes = [C.POINTER(GridDesc), C.c_void_p, C.c_void_p, C.c_void_p, _PI, _PD]
        L.epi_grid_create.argtypes = [C.POINTER(GridDesc)]
        L.epi_world_create.argtypes = [C.POINTER(WorldDesc)]
        L.epi_grid_multi_create.restype = C.c_void_p
        L.epi_grid_multi_create.argtypes = [C.POINTER(GridDesc), _PI, C.c_int32]
        L.epi_grid_multi_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _PI, _PD]
        L.epi_grid_multi_ranks.argtypes = [C.c_void_p, _PI, _PI]
        L.epi_grid_multi_destroy.argtypes = [C.c_void_p]
        L.epi_grid_multi_set_timing.argtypes = [C.c_void_p, C.c_int32]
        L.epi_grid_multi_times.argtypes = [C.c_void_p, _PD, _PD, _PD, C.POINTER(C.c_int64)]
        L.epi_grid_multi_peer_access.argtypes = [C.c_void_p, _PU8]
        L.epi_grid_multi_inject_failure.argtypes = [C.c_void_p, C.c_int32]
        L.epi_calibrate_grid_multi.argtypes = [C.POINTER(GridDesc), _PI, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, _PI, _PD]
        L.epi_grid_screen_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.epi_world_screen_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.epi_grid_set_timing.argtypes = [C.c_void_p, C.c_int32]
        L.epi_grid_kernel_time.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.epi_world_result_buffers.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_void_p)]
        L.epi_world_step_packed.argtypes = [C.c_void_p, _PI, _PD, C.c_int32, C.c_void_p, _PD]
        L.epi_world_packed_buffer.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.epi_world_rollout_greedy.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p]
        L.epi_world_set_timing.argtypes = [C.c_void_p, C.c_int32]
        L.epi_world_kernel_time.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.epi_gravity_matrix.argtypes = [C.c_int32, _PI, _PD, _PD, _PD, C.c_double, C.c_double, C.c_double, C.c_double,
                                         C.c_double, _PD, C.c_int32]
        L.epi_gravity_csr.argtypes = [C.c_int32, _PI, _PD, _PD, _PD, C.c_double, C.c_double, C.c_double, C.c_double,
                                      C.c_double, C.c_double, _PI, _PI, _PD, C.c_int32]
        L.epi_grid_destroy.argtypes = [C.c_void_p]
        L.epi_agents_create.restype = C.c_void_p
        L.epi_agents_create.argtypes = [C.c_int32, C.POINTER(LearnHyper), C.c_int32]
        L.epi_agents_destroy.argtypes = [C.c_void_p]
        L.epi_agents_observe.argtypes = [C.c_void_p, _PI, C.c_int32, C.POINTER(CState), _PI, _PD, C.POINTER(CState), _PU8]
        L.epi_agents_observe_world.argtypes = [C.c_void_p, C.c_void_p]
        L.epi_world_keep_transitions.argtypes = [C.c_void_p, C.c_int32]
        L.epi_agents_replay_size.argtypes = [C.c_void_p, _PI]
        L.epi_agents_sample.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _PI, _PI]
        L.epi_agents_learn.argtypes = [C.c_void_p, _PI, C.c_int32, _PI, _PI, C.POINTER(C.c_int64), _PD, _PD]
        L.epi_agents_q_values.argtypes = [C.c_void_p, _PI, C.c_int32, C.POINTER(CState), C.c_int32, _PD, _PI]
        L.epi_agents_get_weights.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [_PD] * 6
        L.epi_agents_set_weights.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [_PD] * 6
        L.epi_agents_set_timing.argtypes = [C.c_void_p, C.c_int32]
        L.epi_agents_kernel_time.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        for name in ("epi_world_destroy", "epi_world_reset"):
            getattr(L, name).argtypes = [C.c_void_p]
        _lib = L
    return _lib


def last_error():
    buf = C.create_string_buffer(1024)
    lib().epi_last_error(buf, 1024)
    return buf.value.decode("utf-8", "replace")


def raise_last(default=EPI_ERR_BAD_ARGUMENT):
    """Raises the calling thread's last failure (after a *_create returned NULL)."""
    status = lib().epi_last_status() or default
    check(status)


def check(status):
    if status != EPI_OK:
        msg = last_error()
        if status in (EPI_ERR_LOCAL_MASS, EPI_ERR_WORLD_MASS, EPI_ERR_ASSERT_STATE):
            raise IllegalArgumentException(status, msg)
        raise EpiError(status, msg)


def ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)
