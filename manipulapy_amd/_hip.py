"""ctypes binding of libmanipula_hip.so (include/manipula_hip.h).

This is the whole FFI: no PyTorch, CuPy or Triton on the product path.  There is no CPU fallback
behind it — if the library or a GPU is missing the calls raise (`HipUnavailableError` /
`HipError`), they never silently compute somewhere else.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Optional, Sequence

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_ENV = "MANIPULAPY_HIP_LIB"  # override the library path
DEFAULT_LIB = os.path.join(_PKG, "libmanipula_hip.so")

MP_OK = 0
MP_MAX_DOF = 8    # fully unrolled / specialisable kernels
MP_BIG_DOF = 32   # run-time-n kernels (csrc/mp_dyn.h)
UNIQUE_ID_BYTES = 128


class HipError(RuntimeError):
    """A libmanipula_hip call returned a non-zero code (message from mp_last_error)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"[manipula_hip rc={code}] {message}")
        self.code = code


class HipUnavailableError(RuntimeError):
    """The native library cannot be loaded, or no GPU is visible."""


_c_dp = ctypes.POINTER(ctypes.c_double)
_c_fp = ctypes.POINTER(ctypes.c_float)
_vp = ctypes.c_void_p
_i64 = ctypes.c_int64

# name -> (restype, argtypes); the list is also what tests/test_cabi_symbols.py checks against the header
SIGNATURES = {
    "mp_version": (ctypes.c_int, []),
    "mp_last_error": (ctypes.c_char_p, []),
    "mp_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "mp_ctx_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_vp)]),
    "mp_ctx_destroy": (ctypes.c_int, [_vp]),
    "mp_ctx_synchronize": (ctypes.c_int, [_vp]),
    "mp_ctx_get_stream": (ctypes.c_int, [_vp, ctypes.POINTER(_vp)]),
    "mp_ctx_properties": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint64)]),
    "mp_selftest": (ctypes.c_int, [_vp]),
    "mp_stream_bandwidth": (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    "mp_stream_bandwidth_mix": (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    "mp_clock_sample_begin": (ctypes.c_int, [_vp, ctypes.c_double]),
    "mp_clock_sample_end": (ctypes.c_int, [_vp, _c_dp, _c_dp]),
    "mp_ctx_set_profiling": (ctypes.c_int, [_vp, ctypes.c_int]),
    "mp_ctx_profile": (ctypes.c_int, [_vp, _c_dp, ctypes.POINTER(ctypes.c_int64), _c_dp, ctypes.c_int]),
    "mp_malloc": (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.POINTER(_vp)]),
    "mp_free": (ctypes.c_int, [_vp, _vp]),
    "mp_pool_trim": (ctypes.c_int, [_vp]),
    "mp_memcpy_h2d": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t]),
    "mp_memcpy_d2h": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t]),
    "mp_memset": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_size_t]),
    "mp_event_create": (ctypes.c_int, [_vp, ctypes.POINTER(_vp)]),
    "mp_event_destroy": (ctypes.c_int, [_vp]),
    "mp_event_record": (ctypes.c_int, [_vp, _vp]),
    "mp_event_elapsed_ms": (ctypes.c_int, [_vp, _vp, ctypes.POINTER(ctypes.c_float)]),
    "mp_host_alloc": (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.POINTER(_vp)]),
    "mp_host_free": (ctypes.c_int, [_vp, _vp]),
    "mp_inverse_kinematics_f64": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int64, _c_dp, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, _vp, _vp, _vp, _vp]),
    "mp_inverse_kinematics_host_f64": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int64, _c_dp, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, _vp, _vp, _vp, _vp]),
    "mp_graph_begin": (ctypes.c_int, [_vp]),
    "mp_graph_end": (ctypes.c_int, [_vp, ctypes.POINTER(_vp)]),
    "mp_graph_launch": (ctypes.c_int, [_vp, _vp]),
    "mp_graph_destroy": (ctypes.c_int, [_vp]),
    "mp_model_create": (ctypes.c_int, [ctypes.c_int, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, ctypes.POINTER(_vp)]),
    "mp_model_destroy": (ctypes.c_int, [_vp]),
    "mp_model_dof": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int)]),
    "mp_model_params": (ctypes.c_int, [_vp, _c_dp]),
    "mp_model_fk_host": (ctypes.c_int, [_vp, _c_dp, _c_dp]),
    "mp_model_specialize": (ctypes.c_int, [_vp, _vp]),
    "mp_model_is_specialized": (ctypes.c_int, [_vp, _vp, ctypes.POINTER(ctypes.c_int)]),
    "mp_model_specialize_compile": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int)]),
    "mp_model_specialize_source": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t)]),
    "mp_model_blob": (ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.POINTER(ctypes.c_size_t)]),
    "mp_batch_trajectory_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, ctypes.c_double, ctypes.c_int, _vp, _vp, _vp]),
    "mp_id_trajectory_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _c_dp, _c_dp, _vp]),
    "mp_id_trajectory_f64": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _c_dp, _c_dp, _vp]),
    "mp_traj_id_fused_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, ctypes.c_double, ctypes.c_int, _c_dp, _c_dp, _vp]),
    "mp_fk_jac_id_f64": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _c_dp, _c_dp, _vp, _vp, _vp]),
    "mp_fk_jac_id_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _c_dp, _c_dp, _vp, _vp, _vp]),
    "mp_mass_matrix_f64": (ctypes.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "mp_mass_matrix_f32": (ctypes.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "mp_forward_dynamics_f64": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _c_dp, _c_dp, _vp]),
    "mp_forward_dynamics_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _c_dp, _c_dp, _vp]),
    "mp_fd_trajectory_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _c_dp, ctypes.c_double, ctypes.c_int, _vp, _vp, _vp]),
    "mp_fd_trajectory_f64": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _c_dp, ctypes.c_double, ctypes.c_int, _vp, _vp, _vp]),
    "mp_fd_trajectory_tm_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _c_dp, ctypes.c_double, ctypes.c_int, _vp, _vp, _vp]),
    "mp_fd_trajectory_tm_f64": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _c_dp, ctypes.c_double, ctypes.c_int, _vp, _vp, _vp]),
    "mp_transpose_rows": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _vp]),
    "mp_mass_matrix_host_f64": (ctypes.c_int, [_vp, _vp, _c_dp, _i64, _c_dp]),
    "mp_forward_dynamics_host_f64": (ctypes.c_int, [_vp, _vp, _c_dp, _c_dp, _c_dp, _i64, _c_dp, _c_dp, _c_dp]),
    "mp_fd_trajectory_host_f32": (ctypes.c_int, [_vp, _vp, _c_fp, _c_fp, _c_fp, _c_fp, _i64, _i64, _c_dp, ctypes.c_double, ctypes.c_int, _c_fp, _c_fp, _c_fp]),
    "mp_fd_trajectory_host_f64": (ctypes.c_int, [_vp, _vp, _c_dp, _c_dp, _c_dp, _c_dp, _i64, _i64, _c_dp, ctypes.c_double, ctypes.c_int, _c_fp, _c_fp, _c_fp]),
    "mp_cartesian_trajectory_f32": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i64, ctypes.c_double, ctypes.c_int, _vp, _vp, _vp, _vp]),
    "mp_cartesian_trajectory_host_f32": (ctypes.c_int, [_vp, _c_dp, _c_dp, _i64, _i64, ctypes.c_double, ctypes.c_int, _c_fp, _c_fp, _c_fp, _c_fp]),
    "mp_potential_field_f32": (ctypes.c_int, [_vp, _vp, _c_fp, _vp, _i64, _i64, ctypes.c_float, _vp, _vp]),
    "mp_potential_field_host_f32": (ctypes.c_int, [_vp, _c_fp, _c_fp, _c_fp, _i64, _i64, ctypes.c_float, _c_fp, _c_fp]),
    "mp_batch_trajectory_host_f32": (ctypes.c_int, [_vp, _vp, _c_fp, _c_fp, _i64, _i64, ctypes.c_double, ctypes.c_int, _c_fp, _c_fp, _c_fp]),
    "mp_id_trajectory_host_f32": (ctypes.c_int, [_vp, _vp, _c_fp, _c_fp, _c_fp, _i64, _c_dp, _c_dp, _c_fp]),
    "mp_id_trajectory_host_f64": (ctypes.c_int, [_vp, _vp, _c_dp, _c_dp, _c_dp, _i64, _c_dp, _c_dp, _c_dp]),
    "mp_traj_id_fused_host_f32": (ctypes.c_int, [_vp, _vp, _c_fp, _c_fp, _i64, _i64, ctypes.c_double, ctypes.c_int, _c_dp, _c_dp, _c_fp]),
    "mp_fk_jac_id_host_f64": (ctypes.c_int, [_vp, _vp, _c_dp, _c_dp, _c_dp, _i64, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp]),
    "mp_pd_regulation_host_f64": (ctypes.c_int, [_vp, _vp, _c_dp, _c_dp, _c_dp, _c_dp, _i64, _c_dp, ctypes.c_double, ctypes.c_int, _c_dp, _vp]),
    "mp_pd_regulation_cpu_f64": (ctypes.c_int, [_vp, _c_dp, _c_dp, _c_dp, _c_dp, _i64, _c_dp, ctypes.c_double, ctypes.c_int, _c_dp, _vp, ctypes.c_int]),
    "mp_cpu_threads": (ctypes.c_int, [_i64]),
    "mp_id_trajectory_cpu_f32": (ctypes.c_int, [_vp, _c_fp, _c_fp, _c_fp, _i64, _c_dp, _c_dp, _c_fp, ctypes.c_int]),
    "mp_id_row_precision_cpu_f32": (ctypes.c_int, [_vp, _c_fp, _c_fp, _c_fp, _i64, _c_dp, _c_dp, _vp, ctypes.c_int]),
    "mp_id_trajectory_cpu_f64": (ctypes.c_int, [_vp, _c_dp, _c_dp, _c_dp, _i64, _c_dp, _c_dp, _c_dp, ctypes.c_int]),
    "mp_fk_jac_id_cpu_f64": (ctypes.c_int, [_vp, _c_dp, _c_dp, _c_dp, _i64, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, ctypes.c_int]),
    "mp_mass_matrix_cpu_f64": (ctypes.c_int, [_vp, _c_dp, _i64, _c_dp, ctypes.c_int]),
    "mp_forward_dynamics_cpu_f64": (ctypes.c_int, [_vp, _c_dp, _c_dp, _c_dp, _i64, _c_dp, _c_dp, _c_dp, ctypes.c_int]),
    "mp_fd_trajectory_cpu_f32": (ctypes.c_int, [_vp, _c_fp, _c_fp, _c_fp, _c_fp, _i64, _i64, _c_dp, ctypes.c_double, ctypes.c_int, _c_fp, _c_fp, _c_fp, ctypes.c_int]),
    "mp_fd_trajectory_cpu_f64": (ctypes.c_int, [_vp, _c_dp, _c_dp, _c_dp, _c_dp, _i64, _i64, _c_dp, ctypes.c_double, ctypes.c_int, _c_fp, _c_fp, _c_fp, ctypes.c_int]),
    "mp_inverse_kinematics_cpu_f64": (ctypes.c_int, [_vp, _c_dp, _c_dp, ctypes.c_int64, _c_dp, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, _c_dp, _vp, _vp, _vp, ctypes.c_int]),
    "mp_cartesian_trajectory_cpu_f32": (ctypes.c_int, [_c_dp, _c_dp, _i64, _i64, ctypes.c_double, ctypes.c_int, _c_fp, _c_fp, _c_fp, _c_fp, ctypes.c_int]),
    "mp_comm_unique_id": (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint8)]),
    "mp_comm_create": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint8), ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp)]),
    "mp_comm_destroy": (ctypes.c_int, [_vp]),
    "mp_comm_allgather": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t]),
    "mp_comm_exchange_chunk": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t]),
    "mp_comm_join": (ctypes.c_int, [_vp]),
    "mp_comm_allgatherv": (ctypes.c_int, [_vp, _vp, _vp, ctypes.POINTER(ctypes.c_size_t)]),
    "mp_comm_exchange_chunk_v": (ctypes.c_int, [_vp, _vp, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
}

_lib = None
_lib_lock = threading.Lock()


def lib_path() -> str:
    return os.environ.get(LIB_ENV, DEFAULT_LIB)


def load_library():
    """Load (once) and type the shared library.  Raises HipUnavailableError if it is not built."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        path = lib_path()
        if not os.path.exists(path) and path == DEFAULT_LIB:
            # a fresh checkout: compile the native library in-tree (hipcc cross-compiles without a GPU)
            try:
                from .build import build as _build

                _build(verbose=False)
            except Exception as exc:
                raise HipUnavailableError(
                    f"{path} not found and building it failed ({exc}); run `python -m manipulapy_amd.build` "
                    "(needs hipcc) - there is no CPU fallback for the HIP backend") from exc
        if not os.path.exists(path):
            raise HipUnavailableError(
                f"{path} not found: build it with `python -m manipulapy_amd.build` (needs hipcc); "
                "there is no CPU fallback for the HIP backend")
        try:
            lib = ctypes.CDLL(path)
        except OSError as exc:
            raise HipUnavailableError(f"cannot load {path}: {exc}") from exc
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def _check(rc: int) -> None:
    if rc != MP_OK:
        msg = load_library().mp_last_error()
        raise HipError(rc, msg.decode("utf-8", "replace") if msg else "unknown error")


def device_count() -> int:
    """Number of visible GPUs (0 if none).  Raises only when the library itself is missing."""
    n = ctypes.c_int(0)
    rc = load_library().mp_device_count(ctypes.byref(n))
    if rc != MP_OK:
        return 0
    return int(n.value)


def _dptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_c_dp)


def _fptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_c_fp)


def _as_c(a, dtype, shape=None, name="array") -> np.ndarray:
    arr = np.ascontiguousarray(a, dtype=dtype)
    if shape is not None and tuple(arr.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(arr.shape)}")
    return arr


def _vec_or_none(v, k, name):
    if v is None:
        return None
    return _as_c(v, np.float64, (k,), name)


class DeviceBuffer:
    """A pooled device allocation (mp_malloc / mp_free)."""

    def __init__(self, ctx: "HipContext", nbytes: int):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = _vp()
        _check(ctx.lib.mp_malloc(ctx.handle, ctypes.c_size_t(max(self.nbytes, 1)), ctypes.byref(p)))
        self.ptr = p

    def upload(self, host: np.ndarray) -> "DeviceBuffer":
        host = np.ascontiguousarray(host)
        if host.nbytes > self.nbytes:
            raise ValueError("upload larger than the device buffer")
        _check(self.ctx.lib.mp_memcpy_h2d(self.ctx.handle, self.ptr, host.ctypes.data_as(_vp), ctypes.c_size_t(host.nbytes)))
        return self

    def download(self, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        if out.nbytes > self.nbytes:
            raise ValueError("download larger than the device buffer")
        _check(self.ctx.lib.mp_memcpy_d2h(self.ctx.handle, out.ctypes.data_as(_vp), self.ptr, ctypes.c_size_t(out.nbytes)))
        return out

    def free(self) -> None:
        if self.ptr is not None and self.ctx.handle is not None:
            self.ctx.lib.mp_free(self.ctx.handle, self.ptr)
        self.ptr = None

    def offset(self, nbytes: int) -> ctypes.c_void_p:
        return _vp(self.ptr.value + int(nbytes))


class HipEvent:
    def __init__(self, ctx: "HipContext"):
        self.ctx = ctx
        p = _vp()
        _check(ctx.lib.mp_event_create(ctx.handle, ctypes.byref(p)))
        self.handle = p

    def record(self) -> None:
        _check(self.ctx.lib.mp_event_record(self.ctx.handle, self.handle))

    def elapsed_ms_since(self, start: "HipEvent") -> float:
        ms = ctypes.c_float(0)
        _check(self.ctx.lib.mp_event_elapsed_ms(start.handle, self.handle, ctypes.byref(ms)))
        return float(ms.value)

    def destroy(self) -> None:
        if self.handle is not None:
            self.ctx.lib.mp_event_destroy(self.handle)
            self.handle = None


class PinnedBuffer:
    """Page-locked host memory (mp_host_alloc).  ``np.asarray(buf)`` / ``buf.array(shape, dtype)`` give NumPy views that
    keep the allocation alive; on such arrays the *_host entry points overlap upload, kernels and download."""

    def __init__(self, ctx: "HipContext", nbytes: int):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = _vp()
        _check(ctx.lib.mp_host_alloc(ctx.handle, ctypes.c_size_t(self.nbytes), ctypes.byref(p)))
        self.ptr = p.value
        self.__array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 3}

    def array(self, shape, dtype) -> np.ndarray:
        a = np.asarray(self).view(np.dtype(dtype))
        n = int(np.prod(shape))
        if n > a.size:
            raise ValueError("shape larger than the pinned buffer")
        return a[:n].reshape(shape)

    def __del__(self):
        try:
            if self.ptr:  # valid with or without the allocating context: the buffer does not belong to it
                self.ctx.lib.mp_host_free(None, _vp(self.ptr))
        except Exception:
            pass
        self.ptr = None


class HipGraph:
    """A captured sequence of device-pointer launches (mp_graph_*), replayed with one submission."""

    def __init__(self, ctx: "HipContext", handle):
        self.ctx, self.handle = ctx, handle

    def launch(self) -> None:
        _check(self.ctx.lib.mp_graph_launch(self.ctx.handle, self.handle))

    def destroy(self) -> None:
        if self.handle is not None:
            self.ctx.lib.mp_graph_destroy(self.handle)
            self.handle = None


class _Capture:
    def __init__(self, ctx: "HipContext"):
        self.ctx, self.graph = ctx, None

    def __enter__(self):
        _check(self.ctx.lib.mp_graph_begin(self.ctx.handle))
        return self

    def __exit__(self, exc_type, exc, tb):
        p = _vp()
        rc = self.ctx.lib.mp_graph_end(self.ctx.handle, ctypes.byref(p))
        if exc_type is None:
            _check(rc)
            self.graph = HipGraph(self.ctx, p)
        elif rc == 0:
            self.ctx.lib.mp_graph_destroy(p)
        return False


class HipModel:
    """Compiled robot model (mp_model_create).  Host-only object: no GPU needed to build one."""

    def __init__(self, S_list, Mlist_per_link, Glist, M_ee, joint_limits=None, torque_limits=None):
        self.lib = load_library()
        S = _as_c(S_list, np.float64, name="S_list")
        if S.ndim != 2 or S.shape[0] != 6:
            raise ValueError(f"S_list must be (6, n), got {S.shape}")
        n = S.shape[1]
        Mc = _as_c(Mlist_per_link, np.float64, (n, 4, 4), "Mlist_per_link")
        G = _as_c(Glist, np.float64, (n, 6, 6), "Glist")
        Me = _as_c(M_ee, np.float64, (4, 4), "M_list")
        jl = None if joint_limits is None else _as_c(joint_limits, np.float64, (n, 2), "joint_limits")
        tl = None if torque_limits is None else _as_c(torque_limits, np.float64, (n, 2), "torque_limits")
        p = _vp()
        _check(self.lib.mp_model_create(n, _dptr(S), _dptr(Mc), _dptr(G), _dptr(Me), _dptr(jl), _dptr(tl), ctypes.byref(p)))
        self.handle = p
        self.n = n

    def params(self) -> np.ndarray:
        out = np.zeros((self.n, 16))
        _check(self.lib.mp_model_params(self.handle, _dptr(out)))
        return out

    def specialize_source(self, part: int = 0) -> str:
        """One of the two translation units the run-time specialiser compiles for this robot (part 1: the one-row-per-lane float32
        inverse dynamics, built with the max-ILP scheduling strategy)."""
        n = ctypes.c_size_t(0)
        _check(self.lib.mp_model_specialize_source(self.handle, None, ctypes.byref(n)))
        buf = ctypes.create_string_buffer(n.value)
        _check(self.lib.mp_model_specialize_source(self.handle, buf, ctypes.byref(n)))
        both = buf.value.decode().split("\n// ==== second program", 1)
        return both[0] if part == 0 else "// ==== second program" + both[1]

    def specialize_compile(self):
        """hiprtc-compile the specialised kernels (no GPU needed): (code bytes, came from the disk cache)."""
        nb, cached = ctypes.c_size_t(0), ctypes.c_int(0)
        _check(self.lib.mp_model_specialize_compile(self.handle, ctypes.byref(nb), ctypes.byref(cached)))
        return int(nb.value), bool(cached.value)

    def blob(self, dtype=np.float32) -> dict:
        """The compiled model as the kernels see it, unpacked by field (csrc/mp_model.h layout)."""
        f64 = np.dtype(dtype) == np.float64
        nb = ctypes.c_size_t(0)
        _check(self.lib.mp_model_blob(self.handle, int(f64), None, ctypes.byref(nb)))
        raw = np.zeros(nb.value, dtype=np.uint8)
        _check(self.lib.mp_model_blob(self.handle, int(f64), raw.ctypes.data_as(_vp), ctypes.byref(nb)))
        w = 8 if f64 else 4
        head = 16 if not f64 else 16  # int n + 3 pad ints
        vals = raw[head:].view(np.float64 if f64 else np.float32)
        o = 0
        def take(k):
            nonlocal o
            v = vals[o:o + k].copy(); o += k
            return v
        d = {"n": int(raw[:4].view(np.int32)[0]), "base_R": take(9), "base_p": take(3), "tool_R": take(9), "tool_p": take(3)}
        F = 18                                           # MP_JOINT_FIELDS (csrc/mp_model.h): the 16 of params() + cos / sin of the offset
        cap = (nb.value - head - 24 * w) // ((F + 4) * w)   # MP_MAX_DOF, or MP_BIG_DOF for the looped kernels' model
        d["joints"] = take(F * cap).reshape(cap, F)
        for k in ("qmin", "qmax", "taumin", "taumax"):
            d[k] = take(cap)
        assert o * w + head == nb.value, (o * w + head, nb.value)
        return d

    def joint_limits_f32(self) -> np.ndarray:
        """(n, 2) float32 joint limits exactly as the kernels clip against them (+-inf where the model is unbounded)."""
        b = self.blob(np.float32)
        return np.stack([b["qmin"][:self.n], b["qmax"][:self.n]], axis=1).astype(np.float32)

    def fk_host(self, q) -> np.ndarray:
        q = _as_c(q, np.float64, (self.n,), "q")
        T = np.zeros((4, 4))
        _check(self.lib.mp_model_fk_host(self.handle, _dptr(q), _dptr(T)))
        return T

    def destroy(self) -> None:
        if getattr(self, "handle", None) is not None:
            self.lib.mp_model_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class HipContext:
    """One device context (streams + pooled device memory).  Fails loudly without a GPU."""

    def __init__(self, device_id: int = 0):
        self.lib = load_library()
        self.handle = None
        n = ctypes.c_int(0)
        rc = self.lib.mp_device_count(ctypes.byref(n))
        if rc != MP_OK or n.value <= 0:
            raise HipUnavailableError("no HIP device visible: the HIP backend has no CPU fallback")
        p = _vp()
        _check(self.lib.mp_ctx_create(int(device_id), ctypes.byref(p)))
        self.handle = p
        self.device_id = int(device_id)

    # ---- plumbing
    def synchronize(self) -> None:
        _check(self.lib.mp_ctx_synchronize(self.handle))

    def selftest(self) -> None:
        _check(self.lib.mp_selftest(self.handle))

    def stream_bandwidth(self, bytes_per_array: int, reads: int = 1, reps: int = 20, nontemporal: bool = False) -> float:
        """GB/s of a device copy (reads = 1) or of the 3-reads-1-write mix of the inverse-dynamics kernels (reads = 3), with plain
        or non-temporal accesses."""
        out = ctypes.c_double(0.0)
        _check(self.lib.mp_stream_bandwidth(self.handle, ctypes.c_size_t(int(bytes_per_array)), int(reads) + (10 if nontemporal else 0), int(reps),
                                            ctypes.byref(out)))
        return float(out.value)

    def stream_bandwidth_mix(self, bytes_per_array: int, reads: int, writes: int, reps: int = 10, nontemporal: bool = True) -> float:
        """GB/s of a streaming kernel that reads `reads` and writes `writes` arrays of bytes_per_array bytes (a kernel's own byte mix)."""
        out = ctypes.c_double(0.0)
        _check(self.lib.mp_stream_bandwidth_mix(self.handle, ctypes.c_size_t(int(bytes_per_array)), int(reads), int(writes),
                                                1 if nontemporal else 0, int(reps), ctypes.byref(out)))
        return float(out.value)

    def clock_sample_begin(self, duration_ms: float) -> None:
        """Start the bounded shader-clock sampler beside whatever is launched next (mp_clock_sample_begin)."""
        _check(self.lib.mp_clock_sample_begin(self.handle, ctypes.c_double(float(duration_ms))))

    def clock_sample_end(self):
        """(clock in Hz held while the sampler ran, milliseconds its stamps span)."""
        hz, ms = ctypes.c_double(0.0), ctypes.c_double(0.0)
        _check(self.lib.mp_clock_sample_end(self.handle, ctypes.byref(hz), ctypes.byref(ms)))
        return float(hz.value), float(ms.value)

    def properties(self) -> dict:
        name = ctypes.create_string_buffer(256)
        cu = ctypes.c_int(0)
        mem = ctypes.c_uint64(0)
        _check(self.lib.mp_ctx_properties(self.handle, name, 256, ctypes.byref(cu), ctypes.byref(mem)))
        # key names follow the reference's get_gpu_properties() (cuda_kernels/registry.py:335-356)
        return {"name": name.value.decode(), "multiprocessor_count": int(cu.value), "total_memory": int(mem.value),
                "warp_size": 64}

    def specialize(self, model: "HipModel") -> None:
        """Load (compiling if needed) float32 kernels specialised for `model` on this device."""
        _check(self.lib.mp_model_specialize(self.handle, model.handle))

    def is_specialized(self, model: "HipModel") -> bool:
        yes = ctypes.c_int(0)
        _check(self.lib.mp_model_is_specialized(self.handle, model.handle, ctypes.byref(yes)))
        return bool(yes.value)

    def set_profiling(self, on: bool = True) -> None:
        """Timed HIP event pair + roctx range around every entry point's launches (mp_ctx_set_profiling)."""
        _check(self.lib.mp_ctx_set_profiling(self.handle, int(bool(on))))

    def profile(self, reset: bool = False) -> dict:
        """{"kernel_ms_total", "timed_calls", "kernel_ms_last"} since profiling was switched on (or the last reset);
        waits for the launches recorded so far."""
        tot, last, calls = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int64(0)
        _check(self.lib.mp_ctx_profile(self.handle, ctypes.byref(tot), ctypes.byref(calls), ctypes.byref(last), int(bool(reset))))
        return {"kernel_ms_total": float(tot.value), "timed_calls": int(calls.value), "kernel_ms_last": float(last.value)}

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def to_device(self, host: np.ndarray) -> DeviceBuffer:
        host = np.ascontiguousarray(host)
        return DeviceBuffer(self, host.nbytes).upload(host)

    def memset(self, buf, value: int, nbytes: int) -> None:
        """Asynchronous byte fill on the compute stream."""
        _check(self.lib.mp_memset(self.handle, _p(buf), int(value), ctypes.c_size_t(int(nbytes))))

    def event(self) -> HipEvent:
        return HipEvent(self)

    def pinned_empty(self, shape, dtype=np.float32) -> np.ndarray:
        """Uninitialised page-locked NumPy array (freed with its last view, or with the context)."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape))
        return PinnedBuffer(self, max(1, n) * dtype.itemsize).array(shape, dtype)

    def stream(self) -> int:
        """The compute stream (a hipStream_t as an integer) for callers that order their own HIP work behind the launches."""
        p = _vp()
        _check(self.lib.mp_ctx_get_stream(self.handle, ctypes.byref(p)))
        return p.value or 0

    def capture(self) -> _Capture:
        """``with ctx.capture() as cap: <device-pointer launches>`` -> ``cap.graph`` (HipGraph)."""
        return _Capture(self)

    def trim_pool(self) -> None:
        _check(self.lib.mp_pool_trim(self.handle))

    def destroy(self) -> None:
        if self.handle is not None:
            self.lib.mp_ctx_destroy(self.handle)
            self.handle = None

    # ---- hot path on device buffers (asynchronous)
    def id_trajectory(self, model: HipModel, d_q, d_qd, d_qdd, rows: int, d_tau, g=None, Ftip=None, dtype=np.float32):
        fn = self.lib.mp_id_trajectory_f32 if np.dtype(dtype) == np.float32 else self.lib.mp_id_trajectory_f64
        g = _vec_or_none(g, 3, "g")
        F = _vec_or_none(Ftip, 6, "Ftip")
        _check(fn(self.handle, model.handle, _p(d_q), _p(d_qd), _p(d_qdd), int(rows), _dptr(g), _dptr(F), _p(d_tau)))

    def batch_trajectory(self, model, d_start, d_end, B, N, Tf, method, d_pos, d_vel, d_acc):
        _check(self.lib.mp_batch_trajectory_f32(self.handle, model.handle, _p(d_start), _p(d_end), int(B), int(N),
                                                float(Tf), int(method), _p(d_pos), _p(d_vel), _p(d_acc)))

    def traj_id_fused(self, model, d_start, d_end, B, N, Tf, method, d_tau, g=None, Ftip=None):
        g = _vec_or_none(g, 3, "g")
        F = _vec_or_none(Ftip, 6, "Ftip")
        _check(self.lib.mp_traj_id_fused_f32(self.handle, model.handle, _p(d_start), _p(d_end), int(B), int(N),
                                             float(Tf), int(method), _dptr(g), _dptr(F), _p(d_tau)))

    def fk_jac_id(self, model, d_q, d_qd, d_qdd, rows, d_T, d_J, d_tau, g=None, Ftip=None, dtype=np.float64):
        fn = self.lib.mp_fk_jac_id_f64 if np.dtype(dtype) == np.float64 else self.lib.mp_fk_jac_id_f32
        g = _vec_or_none(g, 3, "g")
        F = _vec_or_none(Ftip, 6, "Ftip")
        _check(fn(self.handle, model.handle, _p(d_q), _p(d_qd), _p(d_qdd), int(rows), _dptr(g), _dptr(F), _p(d_T), _p(d_J), _p(d_tau)))

    def mass_matrix(self, model, d_q, rows, d_M, dtype=np.float64):
        fn = self.lib.mp_mass_matrix_f64 if np.dtype(dtype) == np.float64 else self.lib.mp_mass_matrix_f32
        _check(fn(self.handle, model.handle, _p(d_q), int(rows), _p(d_M)))

    def forward_dynamics(self, model, d_q, d_qd, d_tau, rows, d_qdd, g=None, Ftip=None, dtype=np.float64):
        fn = self.lib.mp_forward_dynamics_f64 if np.dtype(dtype) == np.float64 else self.lib.mp_forward_dynamics_f32
        g = _vec_or_none(g, 3, "g")
        F = _vec_or_none(Ftip, 6, "Ftip")
        _check(fn(self.handle, model.handle, _p(d_q), _p(d_qd), _p(d_tau), int(rows), _dptr(g), _dptr(F), _p(d_qdd)))

    def cartesian_trajectory(self, d_Xstart, d_Xend, B, N, Tf, method, d_pos, d_vel, d_acc, d_orient):
        _check(self.lib.mp_cartesian_trajectory_f32(self.handle, _p(d_Xstart), _p(d_Xend), int(B), int(N), float(Tf), int(method),
                                                    _p(d_pos), _p(d_vel), _p(d_acc), _p(d_orient)))

    def potential_field(self, d_positions, goal, d_obstacles, P, O, influence_distance, d_potential, d_gradient):
        goal = np.ascontiguousarray(goal, dtype=np.float32).reshape(3)
        _check(self.lib.mp_potential_field_f32(self.handle, _p(d_positions), _fptr(goal), _p(d_obstacles), int(P), int(O),
                                               float(influence_distance), _p(d_potential), _p(d_gradient)))

    # ---- hot path on host arrays (what the registry's gpu launchers call; synchronous)
    def id_trajectory_host(self, model: HipModel, q, qd, qdd, g=None, Ftip=None, dtype=np.float32, out=None) -> np.ndarray:
        """tau (rows, n) for host arrays.  `out`: optional C-contiguous array of q's shape and dtype to write into (a
        reused or page-locked buffer, see `pinned_empty`); a fresh array is allocated otherwise."""
        dtype = np.dtype(dtype)
        q = _as_c(q, dtype, name="q")
        if q.ndim != 2 or q.shape[1] != model.n:
            raise ValueError(f"q must be (rows, {model.n}), got {q.shape}")
        qd = _as_c(qd, dtype, q.shape, "qd")
        qdd = _as_c(qdd, dtype, q.shape, "qdd")
        tau = _out_or_new(out, q.shape, dtype)
        g = _vec_or_none(g, 3, "g")
        F = _vec_or_none(Ftip, 6, "Ftip")
        if dtype == np.float32:
            _check(self.lib.mp_id_trajectory_host_f32(self.handle, model.handle, _fptr(q), _fptr(qd), _fptr(qdd),
                                                      q.shape[0], _dptr(g), _dptr(F), _fptr(tau)))
        else:
            _check(self.lib.mp_id_trajectory_host_f64(self.handle, model.handle, _dptr(q), _dptr(qd), _dptr(qdd),
                                                      q.shape[0], _dptr(g), _dptr(F), _dptr(tau)))
        return tau

    def batch_trajectory_host(self, model: HipModel, start, end, Tf, N, method):
        start = _as_c(start, np.float32, name="thetastart_batch")
        if start.ndim != 2 or start.shape[1] != model.n:
            raise ValueError(f"thetastart_batch must be (B, {model.n}), got {start.shape}")
        end = _as_c(end, np.float32, start.shape, "thetaend_batch")
        B = start.shape[0]
        out = [np.zeros((B, int(N), model.n), dtype=np.float32) for _ in range(3)]
        _check(self.lib.mp_batch_trajectory_host_f32(self.handle, model.handle, _fptr(start), _fptr(end), B, int(N),
                                                     float(Tf), int(method), _fptr(out[0]), _fptr(out[1]), _fptr(out[2])))
        return tuple(out)

    def traj_id_fused_host(self, model: HipModel, start, end, Tf, N, method, g=None, Ftip=None, out=None) -> np.ndarray:
        start = _as_c(start, np.float32, name="thetastart_batch")
        if start.ndim != 2 or start.shape[1] != model.n:
            raise ValueError(f"thetastart_batch must be (B, {model.n}), got {start.shape}")
        end = _as_c(end, np.float32, start.shape, "thetaend_batch")
        B = start.shape[0]
        tau = _out_or_new(out, (B, int(N), model.n), np.dtype(np.float32))
        g = _vec_or_none(g, 3, "g")
        F = _vec_or_none(Ftip, 6, "Ftip")
        _check(self.lib.mp_traj_id_fused_host_f32(self.handle, model.handle, _fptr(start), _fptr(end), B, int(N),
                                                  float(Tf), int(method), _dptr(g), _dptr(F), _fptr(tau)))
        return tau

    def fk_jac_id_host(self, model: HipModel, q, qd=None, qdd=None, g=None, Ftip=None, want_T=True, want_J=True, out_T=None,
                       out_J=None, out_tau=None):
        """(T (rows,4,4), J (rows,6,n), tau (rows,n)) float64; an output that was not asked for is None.  out_*: arrays to
        write into (reused or page-locked, see `pinned_empty`): with page-locked inputs and outputs a large call is chunked
        and its upload, kernels and download overlap."""
        q = _as_c(q, np.float64, name="q")
        if q.ndim != 2 or q.shape[1] != model.n:
            raise ValueError(f"q must be (rows, {model.n}), got {q.shape}")
        rows = q.shape[0]
        want_tau = qd is not None and qdd is not None
        qd = _as_c(qd, np.float64, q.shape, "qd") if want_tau else None
        qdd = _as_c(qdd, np.float64, q.shape, "qdd") if want_tau else None
        f64 = np.dtype(np.float64)
        T = _out_or_new(out_T, (rows, 4, 4), f64) if want_T else None
        J = _out_or_new(out_J, (rows, 6, model.n), f64) if want_J else None
        tau = _out_or_new(out_tau, (rows, model.n), f64) if want_tau else None
        g = _vec_or_none(g, 3, "g")
        F = _vec_or_none(Ftip, 6, "Ftip")
        _check(self.lib.mp_fk_jac_id_host_f64(self.handle, model.handle, _dptr(q), _dptr(qd), _dptr(qdd), rows,
                                              _dptr(g), _dptr(F), _dptr(T), _dptr(J), _dptr(tau)))
        return T, J, tau

    def cartesian_trajectory_host(self, Xstart, Xend, Tf, N, method):
        """B pose pairs (B,4,4) -> positions / velocities / accelerations (B,N,3), orientations (B,N,3,3), float32."""
        Xs = _as_c(Xstart, np.float64, name="Xstart")
        if Xs.ndim != 3 or Xs.shape[1:] != (4, 4):
            raise ValueError(f"Xstart must be (B, 4, 4), got {Xs.shape}")
        Xe = _as_c(Xend, np.float64, Xs.shape, "Xend")
        B, N = Xs.shape[0], int(N)
        pos, vel, acc = (np.zeros((B, N, 3), dtype=np.float32) for _ in range(3))
        ori = np.zeros((B, N, 3, 3), dtype=np.float32)
        _check(self.lib.mp_cartesian_trajectory_host_f32(self.handle, _dptr(Xs), _dptr(Xe), B, N, float(Tf), int(method),
                                                         _fptr(pos), _fptr(vel), _fptr(acc), _fptr(ori)))
        return pos, vel, acc, ori

    def potential_field_host(self, positions, goal, obstacles, influence_distance):
        """(potential (P,), gradient (P,3)) float32 for positions (P,3), goal (3,), obstacles (O,3)."""
        pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        goal = np.ascontiguousarray(goal, dtype=np.float32).reshape(3)
        obs = np.ascontiguousarray(obstacles, dtype=np.float32).reshape(-1, 3)
        P, O = pos.shape[0], obs.shape[0]
        pot, grad = np.zeros(P, dtype=np.float32), np.zeros((P, 3), dtype=np.float32)
        _check(self.lib.mp_potential_field_host_f32(self.handle, _fptr(pos), _fptr(goal), _fptr(obs) if O else None, P, O,
                                                    float(influence_distance), _fptr(pot), _fptr(grad)))
        return pot, grad

    def inverse_kinematics_host(self, model: HipModel, T_desired, theta0, joint_limits=None, eomg=1e-6, ev=1e-6,
                                max_iterations=10000, damping=2e-2, step_cap=0.3, weight_orientation=1.0, weight_position=1.0,
                                adaptive_tuning=False, backtracking=False, seed=1234):
        """B damped-least-squares IK problems in one launch: T_desired (B,4,4), theta0 (B,n) ->
        (theta (B,n) float64, success (B,) bool, iterations (B,) int32, restarts (B,) int32).
        joint_limits: (n,2) with +-inf / None for open ends, or None for no limits."""
        T = _as_c(T_desired, np.float64, name="T_desired")
        if T.ndim != 3 or T.shape[1:] != (4, 4):
            raise ValueError(f"T_desired must be (B, 4, 4), got {T.shape}")
        B = T.shape[0]
        th0 = _as_c(theta0, np.float64, (B, model.n), "thetalist0")
        lim = None
        if joint_limits is not None:
            lim = np.array([[-np.inf if lo is None else lo, np.inf if hi is None else hi] for lo, hi in joint_limits], dtype=np.float64)
            if lim.shape != (model.n, 2):
                raise ValueError(f"joint_limits must be ({model.n}, 2), got {lim.shape}")
        th = np.zeros((B, model.n))
        ok, it, rs = (np.zeros(B, dtype=np.int32) for _ in range(3))
        _check(self.lib.mp_inverse_kinematics_host_f64(
            self.handle, model.handle, _dptr(T), _dptr(th0), B, _dptr(lim), float(eomg), float(ev), int(max_iterations), float(damping),
            float(step_cap), float(weight_orientation), float(weight_position), int(bool(adaptive_tuning)), int(bool(backtracking)),
            int(seed) & 0xFFFFFFFF, _dptr(th),
            ok.ctypes.data_as(_vp), it.ctypes.data_as(_vp), rs.ctypes.data_as(_vp)))
        return th, ok.astype(bool), it, rs

    def pd_regulation_host(self, model: HipModel, theta0, theta_des, Kp, Kd, g, dt, steps):
        """K closed-loop PD regulation runs in one launch (mp_pd_regulation_host_f64): theta0 / theta_des (K,n), Kp / Kd (K,) ->
        (errors (K,steps) float64 with NaN past each run's end, count (K,) int32)."""
        args = _pd_regulation_args(model, theta0, theta_des, Kp, Kd, g, steps)
        th0, des, kp, kd, gv, err, cnt = args
        _check(self.lib.mp_pd_regulation_host_f64(self.handle, model.handle, _dptr(th0), _dptr(des), _dptr(kp), _dptr(kd), th0.shape[0], _dptr(gv),
                                                  float(dt), int(steps), _dptr(err), cnt.ctypes.data_as(_vp)))
        return err, cnt

    def inverse_kinematics(self, model, d_T_desired, d_theta0, B, d_theta, d_success, d_iterations, d_restarts, joint_limits=None,
                           eomg=1e-6, ev=1e-6, max_iterations=10000, damping=2e-2, step_cap=0.3, weight_orientation=1.0,
                           weight_position=1.0, adaptive_tuning=False, backtracking=False, seed=1234):
        lim = None if joint_limits is None else np.ascontiguousarray(joint_limits, dtype=np.float64)
        _check(self.lib.mp_inverse_kinematics_f64(
            self.handle, model.handle, _p(d_T_desired), _p(d_theta0), int(B), _dptr(lim), float(eomg), float(ev), int(max_iterations),
            float(damping), float(step_cap), float(weight_orientation), float(weight_position), int(bool(adaptive_tuning)),
            int(bool(backtracking)), int(seed) & 0xFFFFFFFF, _p(d_theta),
            _p(d_success), _p(d_iterations), _p(d_restarts)))

    def mass_matrix_host(self, model: HipModel, q) -> np.ndarray:
        q = _as_c(q, np.float64, name="q")
        if q.ndim != 2 or q.shape[1] != model.n:
            raise ValueError(f"q must be (rows, {model.n}), got {q.shape}")
        M = np.zeros((q.shape[0], model.n, model.n))
        _check(self.lib.mp_mass_matrix_host_f64(self.handle, model.handle, _dptr(q), q.shape[0], _dptr(M)))
        return M

    def forward_dynamics_host(self, model: HipModel, q, qd, tau, g=None, Ftip=None) -> np.ndarray:
        q = _as_c(q, np.float64, name="q")
        if q.ndim != 2 or q.shape[1] != model.n:
            raise ValueError(f"q must be (rows, {model.n}), got {q.shape}")
        qd = _as_c(qd, np.float64, q.shape, "qd")
        tau = _as_c(tau, np.float64, q.shape, "tau")
        out = np.zeros_like(q)
        g = _vec_or_none(g, 3, "g")
        F = _vec_or_none(Ftip, 6, "Ftip")
        _check(self.lib.mp_forward_dynamics_host_f64(self.handle, model.handle, _dptr(q), _dptr(qd), _dptr(tau), q.shape[0],
                                                     _dptr(g), _dptr(F), _dptr(out)))
        return out

    def fd_trajectory_host(self, model: HipModel, theta0, dtheta0, taumat, g, Ftipmat, dt, intRes, dtype=np.float64,
                           layout: str = "batch_major", device_layout: str | None = None, out=None):
        """B trajectories: theta0/dtheta0 (B,n), taumat (B,N,n), Ftipmat (B,N,6) or None -> 3 x (B,N,n) float32.

        `out`: optional triple of C-contiguous float32 arrays of the result's shape to write into.  With page-locked inputs AND
        outputs (`pinned_empty`) a batch-major call is cut into chunks of whole trajectories whose upload, roll-out and download
        overlap (csrc/mp_capi.cpp, fdtraj_host_impl).

        layout="time_major": the HOST arrays are (N,B,n) / (N,B,6) and so are the results.  device_layout selects the kernel
        ("batch_major": 4-step LDS tiles on (B,N,n); "time_major": whole lines per step on (N,B,n)); when it differs from the
        host layout the arrays are converted on the device (mp_transpose_rows).  Default: the host layout's own kernel."""
        if layout not in ("batch_major", "time_major") or device_layout not in (None, "batch_major", "time_major"):
            raise ValueError("layout / device_layout must be 'batch_major' or 'time_major'")
        dtype = np.dtype(dtype)
        th = _as_c(theta0, dtype, name="thetalist")
        if th.ndim != 2 or th.shape[1] != model.n:
            raise ValueError(f"thetalist must be (B, {model.n}), got {th.shape}")
        B, n = th.shape
        dth = _as_c(dtheta0, dtype, th.shape, "dthetalist")
        tm = _as_c(taumat, dtype, name="taumat")
        bax = 0 if layout == "batch_major" else 1
        if tm.ndim != 3 or tm.shape[bax] != B or tm.shape[2] != n:
            raise ValueError(f"taumat must be {'(B, N, %d)' % n if bax == 0 else '(N, B, %d)' % n}, got {tm.shape}")
        N = tm.shape[1 - bax]
        Fm = None if Ftipmat is None else _as_c(Ftipmat, dtype, tm.shape[:2] + (6,), "Ftipmat")
        g = _vec_or_none(g, 3, "g")
        if out is None:
            out = [np.zeros(tm.shape[:2] + (n,), dtype=np.float32) for _ in range(3)]
        else:
            out = [_out_or_new(o, tm.shape[:2] + (n,), np.dtype(np.float32)) for o in out]
            if len(out) != 3:
                raise ValueError("out must hold three arrays (positions, velocities, accelerations)")
        if layout == "batch_major" and device_layout in (None, "batch_major"):
            if dtype == np.float32:
                fn, ptr = self.lib.mp_fd_trajectory_host_f32, _fptr
            else:
                fn, ptr = self.lib.mp_fd_trajectory_host_f64, _dptr
            _check(fn(self.handle, model.handle, ptr(th), ptr(dth), ptr(tm), ptr(Fm), B, N, _dptr(g), float(dt), int(intRes),
                      _fptr(out[0]), _fptr(out[1]), _fptr(out[2])))
            return tuple(out)
        if B == 0 or N == 0:
            return tuple(out)
        dev_tm = (device_layout or layout) == "time_major"
        convert = dev_tm != (layout == "time_major")
        outer, inner = tm.shape[0], tm.shape[1]      # of the host arrays
        bufs = []
        try:
            def up(a):
                bufs.append(self.to_device(a))
                return bufs[-1]

            def flip(d, row_bytes, o, i):
                bufs.append(self.alloc(o * i * row_bytes))
                self.transpose_rows(d, o, i, row_bytes, bufs[-1])
                return bufs[-1]

            d_th, d_dth, d_tau = up(th), up(dth), up(tm)
            d_F = up(Fm) if Fm is not None else None
            if convert:
                d_tau = flip(d_tau, n * dtype.itemsize, outer, inner)
                d_F = flip(d_F, 6 * dtype.itemsize, outer, inner) if d_F is not None else None
            d_out = [self.alloc(B * N * n * 4) for _ in range(3)]
            bufs.extend(d_out)
            self.fd_trajectory(model, d_th, d_dth, d_tau, d_F, B, N, g, dt, intRes, *d_out, dtype=dtype, time_major=dev_tm)
            for k in range(3):
                src = flip(d_out[k], n * 4, inner, outer) if convert else d_out[k]
                _check(self.lib.mp_memcpy_d2h(self.handle, out[k].ctypes.data, _p(src), out[k].nbytes))
        finally:
            for b in bufs:
                b.free()
        return tuple(out)

    def fd_trajectory(self, model, d_theta0, d_dtheta0, d_taumat, d_Ftipmat, B, N, g, dt, intRes, d_pos, d_vel, d_acc,
                      dtype=np.float32, time_major: bool = False):
        """Device pointers.  time_major=False: taumat (B,N,n), Ftipmat (B,N,6), outputs (B,N,n); True: (N,B,*) throughout."""
        f32 = np.dtype(dtype) == np.float32
        if time_major:
            fn = self.lib.mp_fd_trajectory_tm_f32 if f32 else self.lib.mp_fd_trajectory_tm_f64
        else:
            fn = self.lib.mp_fd_trajectory_f32 if f32 else self.lib.mp_fd_trajectory_f64
        g = _vec_or_none(g, 3, "g")
        _check(fn(self.handle, model.handle, _p(d_theta0), _p(d_dtheta0), _p(d_taumat), _p(d_Ftipmat), int(B), int(N), _dptr(g),
                  float(dt), int(intRes), _p(d_pos), _p(d_vel), _p(d_acc)))

    def transpose_rows(self, d_src, outer, inner, row_bytes, d_dst):
        """d_dst (inner, outer, row_bytes) <- d_src (outer, inner, row_bytes), on the device."""
        _check(self.lib.mp_transpose_rows(self.handle, _p(d_src), int(outer), int(inner), int(row_bytes), _p(d_dst)))

    # ---- RCCL
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = (ctypes.c_uint8 * UNIQUE_ID_BYTES)()
        _check(load_library().mp_comm_unique_id(buf))
        return bytes(buf)

    def comm_create(self, unique_id: bytes, nranks: int, rank: int) -> "HipComm":
        return HipComm(self, unique_id, nranks, rank)


# --------------------------------------------------------------------------- CPU twins (csrc/mp_cpu.cpp)
# Host arrays in, host arrays out, no context: the registry's cpu_launchers (reference cuda_kernels/registry.py:85-89).
def _ptr(a, dtype):
    if a is None:
        return None
    return a.ctypes.data_as(_c_fp if dtype == np.float32 else _c_dp)


def cpu_id_trajectory(model: "HipModel", q, qd, qdd, g=None, Ftip=None, dtype=np.float32, nthreads: int = 0) -> np.ndarray:
    lib = load_library()
    q = _as_c(q, dtype, name="q")
    if q.ndim != 2 or q.shape[1] != model.n:
        raise ValueError(f"q must be (rows, {model.n}); got {q.shape}")
    qd, qdd = _as_c(qd, dtype, q.shape, "qd"), _as_c(qdd, dtype, q.shape, "qdd")
    tau = np.empty_like(q)
    fn = lib.mp_id_trajectory_cpu_f32 if dtype == np.float32 else lib.mp_id_trajectory_cpu_f64
    _check(fn(model.handle, _ptr(q, dtype), _ptr(qd, dtype), _ptr(qdd, dtype), q.shape[0], _dptr(_vec_or_none(g, 3, "g")),
              _dptr(_vec_or_none(Ftip, 6, "Ftip")), _ptr(tau, dtype), int(nthreads)))
    return tau


def cpu_id_row_precision(model: "HipModel", q, qd, qdd, g=None, Ftip=None, nthreads: int = 0) -> np.ndarray:
    """(rows,) bool: the rows the float32 inverse-dynamics kernels evaluate in float64 (ill-conditioned rows, csrc/mp_core.h)."""
    lib = load_library()
    q = _as_c(q, np.float32, name="q")
    if q.ndim != 2 or q.shape[1] != model.n:
        raise ValueError(f"q must be (rows, {model.n}); got {q.shape}")
    rows = q.shape[0]
    qd, qdd = _as_c(qd, np.float32, q.shape, "qd"), _as_c(qdd, np.float32, q.shape, "qdd")
    out = np.zeros(rows, dtype=np.uint8)
    _check(lib.mp_id_row_precision_cpu_f32(model.handle, _ptr(q, np.float32), _ptr(qd, np.float32), _ptr(qdd, np.float32), rows,
                                           _dptr(_vec_or_none(g, 3, "g")), _dptr(_vec_or_none(Ftip, 6, "Ftip")),
                                           out.ctypes.data_as(_vp), nthreads))
    return out.astype(bool)


def cpu_fk_jac_id(model: "HipModel", q, qd=None, qdd=None, g=None, Ftip=None, want_T=True, want_J=True, nthreads: int = 0):
    lib = load_library()
    q = _as_c(q, np.float64, name="q")
    if q.ndim != 2 or q.shape[1] != model.n:
        raise ValueError(f"q must be (rows, {model.n}); got {q.shape}")
    rows, n = q.shape
    want_tau = qd is not None and qdd is not None
    qd = _as_c(qd, np.float64, q.shape, "qd") if want_tau else None
    qdd = _as_c(qdd, np.float64, q.shape, "qdd") if want_tau else None
    T = np.empty((rows, 4, 4)) if want_T else None
    J = np.empty((rows, 6, n)) if want_J else None
    tau = np.empty((rows, n)) if want_tau else None
    if not (want_T or want_J or want_tau):
        raise ValueError("at least one output is required")
    _check(lib.mp_fk_jac_id_cpu_f64(model.handle, _dptr(q), _dptr(qd), _dptr(qdd), rows, _dptr(_vec_or_none(g, 3, "g")),
                                    _dptr(_vec_or_none(Ftip, 6, "Ftip")), _dptr(T), _dptr(J), _dptr(tau), int(nthreads)))
    return T, J, tau


def cpu_mass_matrix(model: "HipModel", q, nthreads: int = 0) -> np.ndarray:
    q = _as_c(q, np.float64, name="q")
    if q.ndim != 2 or q.shape[1] != model.n:
        raise ValueError(f"q must be (rows, {model.n}); got {q.shape}")
    M = np.empty((q.shape[0], model.n, model.n))
    _check(load_library().mp_mass_matrix_cpu_f64(model.handle, _dptr(q), q.shape[0], _dptr(M), int(nthreads)))
    return M


def cpu_forward_dynamics(model: "HipModel", q, qd, tau, g=None, Ftip=None, nthreads: int = 0) -> np.ndarray:
    q = _as_c(q, np.float64, name="q")
    if q.ndim != 2 or q.shape[1] != model.n:
        raise ValueError(f"q must be (rows, {model.n}); got {q.shape}")
    qd, tau = _as_c(qd, np.float64, q.shape, "qd"), _as_c(tau, np.float64, q.shape, "tau")
    out = np.empty_like(q)
    _check(load_library().mp_forward_dynamics_cpu_f64(model.handle, _dptr(q), _dptr(qd), _dptr(tau), q.shape[0],
                                                      _dptr(_vec_or_none(g, 3, "g")), _dptr(_vec_or_none(Ftip, 6, "Ftip")), _dptr(out),
                                                      int(nthreads)))
    return out


def _pd_regulation_args(model, theta0, theta_des, Kp, Kd, g, steps):
    th0 = _as_c(theta0, np.float64, name="theta0")
    if th0.ndim != 2 or th0.shape[1] != model.n:
        raise ValueError(f"theta0 must be (K, {model.n}), got {th0.shape}")
    K = th0.shape[0]
    des = _as_c(theta_des, np.float64, (K, model.n), "theta_des")
    kp, kd = _as_c(Kp, np.float64, (K,), "Kp"), _as_c(Kd, np.float64, (K,), "Kd")
    gv = None if g is None else _as_c(g, np.float64, (3,), "g")
    if int(steps) < 0:
        raise ValueError("steps must be non-negative")
    return th0, des, kp, kd, gv, np.full((K, int(steps)), np.nan), np.zeros(K, dtype=np.int32)


def cpu_pd_regulation(model: "HipModel", theta0, theta_des, Kp, Kd, g, dt, steps, nthreads: int = 0):
    """HipContext.pd_regulation_host on the host cores (mp_pd_regulation_cpu_f64): same arguments, same results."""
    th0, des, kp, kd, gv, err, cnt = _pd_regulation_args(model, theta0, theta_des, Kp, Kd, g, steps)
    _check(load_library().mp_pd_regulation_cpu_f64(model.handle, _dptr(th0), _dptr(des), _dptr(kp), _dptr(kd), th0.shape[0], _dptr(gv), float(dt),
                                                   int(steps), _dptr(err), cnt.ctypes.data_as(_vp), int(nthreads)))
    return err, cnt


def cpu_inverse_kinematics(model: "HipModel", T_desired, theta0, joint_limits=None, eomg=1e-6, ev=1e-6, max_iterations=10000,
                           damping=2e-2, step_cap=0.3, weight_orientation=1.0, weight_position=1.0, adaptive_tuning=False,
                           backtracking=False, seed=1234, nthreads: int = 0):
    """HipContext.inverse_kinematics_host on the host cores (mp_inverse_kinematics_cpu_f64): same arguments, same results."""
    T = _as_c(T_desired, np.float64, name="T_desired")
    if T.ndim != 3 or T.shape[1:] != (4, 4):
        raise ValueError(f"T_desired must be (B, 4, 4), got {T.shape}")
    B = T.shape[0]
    th0 = _as_c(theta0, np.float64, (B, model.n), "thetalist0")
    lim = None
    if joint_limits is not None:
        lim = np.array([[-np.inf if lo is None else lo, np.inf if hi is None else hi] for lo, hi in joint_limits], dtype=np.float64)
        if lim.shape != (model.n, 2):
            raise ValueError(f"joint_limits must be ({model.n}, 2), got {lim.shape}")
    th = np.zeros((B, model.n))
    ok, it, rs = (np.zeros(B, dtype=np.int32) for _ in range(3))
    _check(load_library().mp_inverse_kinematics_cpu_f64(
        model.handle, _dptr(T), _dptr(th0), B, _dptr(lim), float(eomg), float(ev), int(max_iterations), float(damping), float(step_cap),
        float(weight_orientation), float(weight_position), int(bool(adaptive_tuning)), int(bool(backtracking)), int(seed) & 0xFFFFFFFF,
        _dptr(th), ok.ctypes.data_as(_vp), it.ctypes.data_as(_vp), rs.ctypes.data_as(_vp), int(nthreads)))
    return th, ok.astype(bool), it, rs


def cpu_fd_trajectory(model: "HipModel", theta0, dtheta0, taumat, g, Ftipmat, dt, intRes, dtype=np.float64, nthreads: int = 0):
    lib = load_library()
    tm = _as_c(taumat, dtype, name="taumat")
    if tm.ndim != 3 or tm.shape[2] != model.n:
        raise ValueError(f"taumat must be (B, N, {model.n}); got {tm.shape}")
    B, N, n = tm.shape
    th, dth = _as_c(theta0, dtype, (B, n), "theta0"), _as_c(dtheta0, dtype, (B, n), "dtheta0")
    Fm = None if Ftipmat is None else _as_c(Ftipmat, dtype, (B, N, 6), "Ftipmat")
    out = [np.empty((B, N, n), dtype=np.float32) for _ in range(3)]
    fn = lib.mp_fd_trajectory_cpu_f32 if dtype == np.float32 else lib.mp_fd_trajectory_cpu_f64
    _check(fn(model.handle, _ptr(th, dtype), _ptr(dth, dtype), _ptr(tm, dtype), _ptr(Fm, dtype), B, N, _dptr(_vec_or_none(g, 3, "g")),
              float(dt), int(intRes), _fptr(out[0]), _fptr(out[1]), _fptr(out[2]), int(nthreads)))
    return out[0], out[1], out[2]


def cpu_cartesian_trajectory(Xstart, Xend, Tf, N, method, nthreads: int = 0):
    Xs = _as_c(Xstart, np.float64, name="Xstart")
    if Xs.ndim != 3 or Xs.shape[1:] != (4, 4):
        raise ValueError(f"Xstart must be (B, 4, 4); got {Xs.shape}")
    Xe = _as_c(Xend, np.float64, Xs.shape, "Xend")
    B, N = Xs.shape[0], int(N)
    out = [np.empty((B, N, 3), dtype=np.float32) for _ in range(3)] + [np.empty((B, N, 3, 3), dtype=np.float32)]
    _check(load_library().mp_cartesian_trajectory_cpu_f32(_dptr(Xs), _dptr(Xe), B, N, float(Tf), int(method), *[_fptr(o) for o in out],
                                                          int(nthreads)))
    return tuple(out)


def cpu_threads(items: int = 1 << 30) -> int:
    return int(load_library().mp_cpu_threads(int(items)))


def _out_or_new(out, shape, dtype) -> np.ndarray:
    if out is None:
        return np.empty(shape, dtype=dtype)
    if not isinstance(out, np.ndarray) or out.shape != tuple(shape) or out.dtype != dtype or not out.flags.c_contiguous \
            or not out.flags.writeable:
        raise ValueError(f"out must be a writeable C-contiguous {np.dtype(dtype).name} array of shape {tuple(shape)}")
    return out


def _p(buf):
    if buf is None:
        return None
    if isinstance(buf, DeviceBuffer):
        return buf.ptr
    return buf  # already a c_void_p


class HipComm:
    def __init__(self, ctx: HipContext, unique_id: bytes, nranks: int, rank: int):
        if len(unique_id) != UNIQUE_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        self.ctx = ctx
        self.nranks, self.rank = int(nranks), int(rank)
        buf = (ctypes.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        p = _vp()
        _check(ctx.lib.mp_comm_create(ctx.handle, buf, self.nranks, self.rank, ctypes.byref(p)))
        self.handle = p

    def allgather(self, d_send, d_recv, bytes_per_rank: int) -> None:
        _check(self.ctx.lib.mp_comm_allgather(self.handle, _p(d_send), _p(d_recv), ctypes.c_size_t(int(bytes_per_rank))))

    def exchange_chunk(self, d_all, bytes_per_rank: int, offset: int, nbytes: int) -> None:
        """Bytes [offset, offset + nbytes) of this rank's slot of `d_all` (just written on the compute stream) go to every
        peer, the peers' same range arrives in their slots, on the communicator's own stream (overlaps later kernels)."""
        _check(self.ctx.lib.mp_comm_exchange_chunk(self.handle, _p(d_all), ctypes.c_size_t(int(bytes_per_rank)),
                                                   ctypes.c_size_t(int(offset)), ctypes.c_size_t(int(nbytes))))

    def _sizes(self, values):
        if len(values) != self.nranks:
            raise ValueError(f"expected {self.nranks} per-rank values, got {len(values)}")
        return (ctypes.c_size_t * self.nranks)(*[int(v) for v in values])

    def allgatherv(self, d_send, d_recv, bytes_of_rank) -> None:
        """Uneven shards: rank r contributes bytes_of_rank[r] bytes; d_recv gets them back to back in rank order."""
        _check(self.ctx.lib.mp_comm_allgatherv(self.handle, _p(d_send), _p(d_recv), self._sizes(bytes_of_rank)))

    def exchange_chunk_v(self, d_all, slot_offset, chunk_offset, chunk_bytes) -> None:
        _check(self.ctx.lib.mp_comm_exchange_chunk_v(self.handle, _p(d_all), self._sizes(slot_offset), self._sizes(chunk_offset),
                                                     self._sizes(chunk_bytes)))

    def join(self) -> None:
        """The compute stream waits for every exchange issued so far."""
        _check(self.ctx.lib.mp_comm_join(self.handle))

    def destroy(self) -> None:
        if self.handle is not None:
            self.ctx.lib.mp_comm_destroy(self.handle)
            self.handle = None
