"""ctypes binding of libmanytor_hip.so (C ABI: include/manytor_hip.h).

The library is the only compute path of this package: if it cannot be loaded the
import of anything that needs it raises -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# MT_LIB_OVERRIDE: load another build of the SAME library (tools/sanitize_host.sh points it at the ASan/UBSan build)
LIB_PATH = os.environ.get("MT_LIB_OVERRIDE") or os.path.join(PKG_DIR, "libmanytor_hip.so")

MT_MAX_DOF = 8
MT_MAX_TARGETS = 32
MT_MAX_RETURN_RING = 64
MT_UNIQUE_ID_BYTES = 128

# mt_status
MT_OK = 0
MT_ERR_INVALID_ARG = -1
MT_ERR_HIP = -2
MT_ERR_NO_DEVICE = -3
MT_ERR_ALLOC = -4
MT_ERR_STATE = -5
MT_ERR_UNSUPPORTED = -6

# mt_field
(F_ACTIONS, F_GOALS, F_POINTS, F_ALIVE, F_OBS, F_REWARD, F_DONE, F_DONE_BITS, F_EE, F_TOTAL_REWARD, F_JOINTS,
 F_EPISODES, F_LAST_RETURN, F_RETURN_RING, F_TRACE, F_ZMIN) = range(16)
# mt_dtype
DT_F32, DT_F64, DT_I32, DT_I64, DT_U8, DT_U32, DT_U64 = range(7)
# mt_layout
ENV_MAJOR, SOA = 0, 1
# flags
FLAG_TERMINATE_ON_GROUND = 0x1
FLAG_HW_TRIG = 0x2
FLAG_DH_IN_LDS = 0x4
FLAG_DIRECT_TRIG = 0x8
FLAG_NO_SPECIALIZE = 0x10
FLAG_TRACE = 0x20
FLAG_DEBUG_ZMIN = 0x40
FLAG_ABLATE_LOOP = 0x100
FLAG_ABLATE_OBS = 0x200


class ManytorError(RuntimeError):
    """A call into libmanytor_hip.so failed."""

    def __init__(self, status, message):
        super().__init__(f"libmanytor_hip: {message} (status {status})")
        self.status = status


class MtReturnStats(C.Structure):
    _fields_ = [("sum", C.c_double), ("min", C.c_double), ("max", C.c_double), ("count", C.c_int64), ("done", C.c_int64)]


class MtConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("device", C.c_int32),
        ("n_envs", C.c_int64),
        ("env_id_base", C.c_int64),
        ("dof", C.c_int32),
        ("n_targets", C.c_int32),
        ("substeps", C.c_int32),
        ("flags", C.c_uint32),
        ("pickup_tol", C.c_float),
        ("radius", C.c_float),
        ("dh_table", C.c_float * (MT_MAX_DOF * 4)),
        ("return_ring", C.c_int32),
        ("obs_frame", C.c_int32),
        ("ee_frame", C.c_int32),
        ("reserved", C.c_int32),
    ]


_HANDLE = C.c_void_p

# name -> (restype, argtypes); exactly the prototypes of include/manytor_hip.h
PROTOTYPES = {
    "mt_version": (C.c_int, []),
    "mt_status_string": (C.c_char_p, [C.c_int]),
    "mt_last_error": (C.c_char_p, [_HANDLE]),
    "mt_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "mt_create": (C.c_int, [C.POINTER(_HANDLE), C.POINTER(MtConfig)]),
    "mt_destroy": (C.c_int, [_HANDLE]),
    "mt_step_kernel_name": (C.c_char_p, [_HANDLE]),
    "mt_describe_dispatch": (C.c_char_p, [_HANDLE]),
    "mt_set_stream": (C.c_int, [_HANDLE, C.c_void_p]),
    "mt_use_own_stream": (C.c_int, [_HANDLE]),
    "mt_sync": (C.c_int, [_HANDLE]),
    "mt_reset": (C.c_int, [_HANDLE, C.c_void_p, C.c_int, C.c_int]),
    "mt_reset_random": (C.c_int, [_HANDLE, C.c_uint64, C.c_uint32]),
    "mt_reset_done": (C.c_int, [_HANDLE, C.c_uint64]),
    "mt_env_reset": (C.c_int, [_HANDLE, C.c_int64, C.c_void_p, C.c_uint64, C.c_uint32]),
    "mt_set_actions": (C.c_int, [_HANDLE, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "mt_sample_actions": (C.c_int, [_HANDLE, C.c_uint64, C.c_uint32]),
    "mt_step": (C.c_int, [_HANDLE]),
    "mt_step_host": (C.c_int, [_HANDLE, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mt_env_step": (C.c_int, [_HANDLE, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mt_bad_action_count": (C.c_int, [_HANDLE, C.POINTER(C.c_uint64)]),
    "mt_step_random": (C.c_int, [_HANDLE, C.c_uint64, C.c_uint32]),
    "mt_rollout": (C.c_int, [_HANDLE, C.c_int, C.c_uint64, C.c_uint32]),
    "mt_rollout_fused": (C.c_int, [_HANDLE, C.c_int, C.c_uint64, C.c_uint32, C.c_int]),
    "mt_observe": (C.c_int, [_HANDLE]),
    "mt_check_done": (C.c_int, [_HANDLE]),
    "mt_get": (C.c_int, [_HANDLE, C.c_int, C.c_void_p, C.c_int64, C.c_int]),
    "mt_set": (C.c_int, [_HANDLE, C.c_int, C.c_void_p, C.c_int64]),
    "mt_set_episode_base": (C.c_int, [_HANDLE, C.c_uint32]),
    "mt_device_ptr": (C.c_int, [_HANDLE, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                C.POINTER(C.c_int)]),
    "mt_comm_unique_id": (C.c_int, [C.c_void_p]),
    "mt_comm_init": (C.c_int, [_HANDLE, C.c_void_p, C.c_int, C.c_int]),
    "mt_comm_destroy": (C.c_int, [_HANDLE]),
    "mt_gather_returns": (C.c_int, [_HANDLE, C.c_int, C.c_int, C.c_void_p, C.c_int64]),
    "mt_gather_returns_begin": (C.c_int, [_HANDLE, C.c_int, C.c_int, C.c_void_p, C.c_int64]),
    "mt_gather_returns_wait": (C.c_int, [_HANDLE, C.c_int, C.POINTER(C.c_float)]),
    "mt_gather_returns_begin_inplace": (C.c_int, [_HANDLE, C.c_int, C.c_int, C.c_void_p, C.c_int64]),
    "mt_comm_total_envs": (C.c_int, [_HANDLE, C.POINTER(C.c_int64)]),
    "mt_reduce_returns": (C.c_int, [_HANDLE, C.c_int, C.c_int, C.c_void_p]),
    "mt_timer_start": (C.c_int, [_HANDLE]),
    "mt_timer_stop": (C.c_int, [_HANDLE, C.POINTER(C.c_float)]),
    "mt_timer_stop_async": (C.c_int, [_HANDLE]),
    "mt_timer_read": (C.c_int, [_HANDLE, C.POINTER(C.c_float)]),
    "mt_timer_lap_begin": (C.c_int, [_HANDLE]),
    "mt_timer_lap_end": (C.c_int, [_HANDLE]),
    "mt_timer_laps_total": (C.c_int, [_HANDLE, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "mt_timer_lap_times": (C.c_int, [_HANDLE, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]),
    "mt_fk_batch": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "mt_route_trace": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mt_r_theta_batch": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mt_stream_probe": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int64)]),
}

_lib = None


def _share_torch_hip_runtime():
    """A torch-ROCm wheel bundles its own libamdhip64.so (soname libamdhip64.so.7) and librccl.so.  Two HIP runtimes
    in one process do not share streams, events or allocations, so if torch is installed its copy is loaded FIRST
    (without importing torch): this library's `libamdhip64.so.7` dependency then resolves to it, whatever the import
    order, and comm.hip is pointed at the RCCL that sits on the same runtime.  Without torch the system ROCm is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    hip = os.path.join(libdir, "libamdhip64.so")
    if os.path.exists(hip):
        try:
            C.CDLL(hip, mode=C.RTLD_GLOBAL)
        except OSError:
            return
        rccl = os.path.join(libdir, "librccl.so")
        if os.path.exists(rccl):
            os.environ.setdefault("MT_RCCL_LIB", rccl)


def load():
    """Load (once) and return the ctypes library.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension has not been built. "
            "Run `python -m manytor_amd.build` (needs hipcc); this package has no CPU fallback."
        )
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)     # AttributeError if the .so does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, handle=None):
    if status == MT_OK:
        return
    lib = load()
    msg = lib.mt_last_error(handle)
    msg = msg.decode() if msg else ""
    if not msg:
        msg = lib.mt_status_string(status).decode()
    if status == MT_ERR_INVALID_ARG:
        raise ValueError(f"libmanytor_hip: {msg}")
    raise ManytorError(status, msg)


def device_count() -> int:
    lib = load()
    n = C.c_int(0)
    rc = lib.mt_device_count(C.byref(n))
    return n.value if rc == MT_OK else 0
