"""Backend-dispatch seam, mirroring the reference's registry semantics.

Reference: ManipulaPy/backend/__init__.py:60-237 (one process-wide active backend under a single
RLock; `register` refuses duplicates; unknown names raise ValueError listing the known ones) and
ManipulaPy/backend/base.py:39-203 (the ArrayBackend contract: `float32/float64`, `is_concrete`,
`gpu_capable`, `cache_token()` and NumPy-shaped primitives that never mutate their inputs).

`gpu_capable` is the single predicate that routes a call to the GPU launchers
(base.py:59-64).  The "hip" backend registered here is a host-array backend — callers pass and
receive NumPy arrays exactly as with the reference's CuPy/Numba path ("callers pass /
receive host NumPy arrays; the library owns device memory") — whose only difference from "numpy"
is `gpu_capable = True`; the device work happens inside the kernel registry's gpu launchers
(manipulapy_amd/registry.py) through the C ABI.
"""
from __future__ import annotations

import threading
from contextlib import contextmanager
from typing import Any, Dict, Iterator, Optional

import numpy as np

__all__ = ["ArrayBackend", "NumpyBackend", "HipBackend", "register", "set_backend", "use_backend",
           "get_backend", "get_registered"]


class ArrayBackend:
    """Array-primitive provider.  Subclasses set the class attributes and `xp` (a NumPy-like module)."""

    float32: Any = np.float32
    float64: Any = np.float64
    is_concrete: bool = True   # arrays are concrete host values -> value-keyed caches are valid
    gpu_capable: bool = False  # the dispatch-boundary predicate (reference base.py:59-64)
    xp: Any = np

    def cache_token(self) -> Any:
        return self

    # construction
    def array(self, obj, dtype=None): return self.xp.array(obj, dtype=dtype)
    def asarray(self, obj, dtype=None): return self.xp.asarray(obj, dtype=dtype)
    def zeros(self, shape, dtype=None): return self.xp.zeros(shape, dtype=dtype)
    def eye(self, n, dtype=None): return self.xp.eye(n, dtype=dtype)
    def stack(self, arrays, axis=0): return self.xp.stack(arrays, axis=axis)
    def concatenate(self, arrays, axis=0): return self.xp.concatenate(arrays, axis=axis)
    def diag(self, v): return self.xp.diag(v)
    # linear algebra
    def svd(self, a, full_matrices=False): return self.xp.linalg.svd(a, full_matrices=full_matrices)
    def svdvals(self, a): return self.xp.linalg.svd(a, compute_uv=False)
    def inv(self, a): return self.xp.linalg.inv(a)
    def pinv(self, a): return self.xp.linalg.pinv(a)
    def solve(self, a, b): return self.xp.linalg.solve(a, b)
    def norm(self, x, ord=None, axis=None): return self.xp.linalg.norm(x, ord=ord, axis=axis)
    def trace(self, a): return self.xp.trace(a)
    # elementwise
    def sin(self, x): return self.xp.sin(x)
    def cos(self, x): return self.xp.cos(x)
    def sqrt(self, x): return self.xp.sqrt(x)
    def arccos(self, x): return self.xp.arccos(x)
    def arctan2(self, y, x): return self.xp.arctan2(y, x)
    def abs(self, x): return self.xp.abs(x)
    def clip(self, x, a_min, a_max): return self.xp.clip(x, a_min, a_max)
    def maximum(self, x1, x2): return self.xp.maximum(x1, x2)
    def minimum(self, x1, x2): return self.xp.minimum(x1, x2)
    def where(self, condition, x, y): return self.xp.where(condition, x, y)
    def cross(self, a, b): return self.xp.cross(a, b)
    def matmul(self, a, b): return self.xp.matmul(a, b)
    # reductions
    def sum(self, x, axis=None): return self.xp.sum(x, axis=axis)
    def amax(self, x, axis=None): return self.xp.amax(x, axis=axis)
    def amin(self, x, axis=None): return self.xp.amin(x, axis=axis)
    def mean(self, x, axis=None): return self.xp.mean(x, axis=axis)
    def argmax(self, x, axis=None): return self.xp.argmax(x, axis=axis)
    def all(self, x, axis=None): return self.xp.all(x, axis=axis)
    def any(self, x, axis=None): return self.xp.any(x, axis=axis)
    def isfinite(self, x): return self.xp.isfinite(x)
    # host / device movement
    def to_device(self, x): return self.xp.asarray(x)
    def to_numpy(self, x): return np.asarray(x)
    def ascontiguous(self, x): return np.ascontiguousarray(x)
    def is_backend_array(self, x) -> bool: return isinstance(x, np.ndarray)


class NumpyBackend(ArrayBackend):
    """Host-CPU backend (the process default, reference backend/numpy_backend.py:36-43)."""

    gpu_capable = False


class HipBackend(NumpyBackend):
    """Host arrays + `gpu_capable = True`: routes registered operations to the HIP launchers.

    Same shape as the reference tests' `_GpuCapableBackend(NumpyBackend)` double
    (reference tests/test_cuda_kernels_cpu.py:31-47).
    """

    gpu_capable = True


_LOCK = threading.RLock()
_REGISTRY: Dict[str, ArrayBackend] = {}
_active: Optional[ArrayBackend] = None


def register(name: str, backend: ArrayBackend) -> None:
    """Register `backend` under `name`; names are never overwritten (reference backend/__init__.py:65-75)."""
    with _LOCK:
        if name in _REGISTRY:
            raise ValueError(f"Backend {name!r} is already registered")
        _REGISTRY[name] = backend


def get_registered(name: str) -> ArrayBackend:
    """Return the backend registered under `name` (reference backend/__init__.py:172-198)."""
    if name in ("cupy", "torch", "jax") and name not in _REGISTRY:
        # the reference lazily imports these (backend/__init__.py:78-169); this build ships none of them
        raise ImportError(
            f"The {name!r} backend is not part of manipulapy_amd (MI355X build: use 'hip' for the GPU path "
            "or 'numpy').")
    with _LOCK:
        if name not in _REGISTRY:
            known = ", ".join(sorted(_REGISTRY)) or "<none>"
            raise ValueError(f"Unknown backend {name!r}. Registered backends: {known}")
        return _REGISTRY[name]


def set_backend(name: str) -> None:
    """Switch the active backend process-wide; explicit opt-in only (reference backend/__init__.py:201-215)."""
    backend = get_registered(name)
    global _active
    with _LOCK:
        _active = backend


def get_backend() -> ArrayBackend:
    with _LOCK:
        return _active


@contextmanager
def use_backend(name: str) -> Iterator[ArrayBackend]:
    """Temporarily activate `name`; the previous backend is restored even if the body raises."""
    previous = get_backend()
    set_backend(name)
    try:
        yield get_backend()
    finally:
        global _active
        with _LOCK:
            _active = previous


register("numpy", NumpyBackend())
register("hip", HipBackend())
set_backend("numpy")
