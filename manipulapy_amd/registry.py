"""Kernel registry + routing predicate, mirroring ManipulaPy/cuda_kernels/registry.py.

Reference semantics kept:
  * `KernelRegistration` is a frozen record (name, implementation, launch_config, cpu_fallback,
    gpu_launcher, cpu_launcher, metadata) with read-only metadata        (registry.py:46-60)
  * `KernelRegistry.register` refuses to replace a name (ValueError)      (:69-73)
  * `KernelRegistry.get` raises KeyError naming the available kernels     (:75-83)
  * `KernelRegistry.execute` picks gpu_launcher iff the physical probe succeeded AND the ACTIVE
    backend is `gpu_capable`, read live at call time                      (:85-89, :729-732)

Deliberate difference: the reference's GPU launchers swallow every exception and return the CPU
result (trajectory_kernels.py:1083-1086).  Here a GPU launcher that fails RAISES — a silent CPU
recompute would make every parity/throughput claim about the HIP path meaningless.  The CPU
launchers (NumPy for trajectory generation and the potential field, the C ABI's *_cpu twins for the
dynamics / kinematics) are reached only through the routing rule above, i.e. with the NumPy backend
active; with the "hip" backend active and no usable device `execute` raises instead of computing on
the host.  Batched IK has no CPU launcher (`BackendNotSupportedError`).
"""
from __future__ import annotations

import logging
import os
import threading
from dataclasses import dataclass
from types import MappingProxyType
from typing import Any, Callable, Dict, Mapping, Optional, Tuple

import numpy as np

from . import _hip
from .backend import get_backend

__all__ = ["KernelRegistration", "KernelRegistry", "execute_registered_kernel", "get_registered_kernel",
           "check_hip_availability", "get_context", "get_gpu_properties", "BackendNotSupportedError",
           "trajectory_cpu", "potential_field_cpu", "HIP_DEVICE_ENV", "FALLBACK_ENV", "fallback_enabled", "fallback_stats",
           "run_gpu_launcher"]

HIP_DEVICE_ENV = "MANIPULAPY_HIP_DEVICE"


class BackendNotSupportedError(NotImplementedError):
    """The requested operation only exists on the HIP path in this build (no CPU twin)."""


@dataclass(frozen=True)
class KernelRegistration:
    name: str
    implementation: Any            # C-ABI symbol name(s) behind the gpu launcher
    launch_config: Callable[..., Any]
    cpu_fallback: Optional[Callable[..., Any]]
    gpu_launcher: Callable[..., Any]
    cpu_launcher: Callable[..., Any]
    metadata: Mapping[str, Any]

    def __post_init__(self) -> None:
        object.__setattr__(self, "metadata", MappingProxyType(dict(self.metadata)))


class KernelRegistry:
    """Fail-closed name -> operation table."""

    def __init__(self) -> None:
        self._entries: Dict[str, KernelRegistration] = {}

    def register(self, entry: KernelRegistration) -> None:
        if entry.name in self._entries:
            raise ValueError(f"HIP kernel '{entry.name}' is already registered")
        self._entries[entry.name] = entry

    def get(self, name: str) -> KernelRegistration:
        try:
            return self._entries[name]
        except KeyError:
            available = ", ".join(sorted(self._entries))
            raise KeyError(f"Unknown HIP kernel '{name}'. Available kernels: {available}") from None

    def names(self):
        return sorted(self._entries)

    def execute(self, name: str, *args: Any, **kwargs: Any) -> Any:
        entry = self.get(name)
        if _hip_routing_enabled():
            return run_gpu_launcher(entry, *args, **kwargs)[0]
        _refuse_silent_cpu(name)
        return entry.cpu_launcher(*args, **kwargs)


# ------------------------------------------------------------------------ reference failure semantics
# The reference wraps every GPU path in try / except and recomputes on the CPU (planning/trajectory.py:270-274,
# planning/trajectory_dynamics.py:292-302, cuda_kernels/trajectory_kernels.py:1083-1086).  Here a failing launch RAISES by default -
# a silent recompute would void every parity and throughput claim - and MANIPULAPY_HIP_FALLBACK=1 opts into the reference's
# behaviour for drop-in users: the HipError is logged with the reference's wording, the operation's registered cpu_launcher runs
# (the C ABI's *_cpu launchers or the NumPy ones - never the test suite's checker), and the event is counted (fallback_stats, and the planner's
# performance_stats["cpu_calls"]).  bench.py and the GPU test suite never set it.
FALLBACK_ENV = "MANIPULAPY_HIP_FALLBACK"
fallback_stats: Dict[str, int] = {"calls": 0}
logger = logging.getLogger("manipulapy_amd")


def fallback_enabled() -> bool:
    return os.environ.get(FALLBACK_ENV) == "1"


def run_gpu_launcher(entry: KernelRegistration, *args: Any, **kwargs: Any):
    """(result, "gpu" | "cpu"): the entry's GPU launcher; with MANIPULAPY_HIP_FALLBACK=1 a HipError / HipUnavailableError from it
    is logged and answered by the entry's CPU launcher, as the reference does."""
    try:
        return entry.gpu_launcher(*args, **kwargs), "gpu"
    except (_hip.HipError, _hip.HipUnavailableError) as exc:
        if not fallback_enabled():
            raise
        logger.warning("GPU %s failed: %s, falling back to CPU", entry.name, exc)
        fallback_stats["calls"] += 1
        return entry.cpu_launcher(*args, **kwargs), "cpu"


# ------------------------------------------------------------------------------ device probe / ctx
_probe_lock = threading.Lock()
_probe_result: Optional[bool] = None
_ctx: Optional[_hip.HipContext] = None


def check_hip_availability() -> bool:
    """Physical probe: the library loads and at least one GPU is visible.  Cached per process.

    Replaces check_cuda_availability (reference registry.py:92-137).  `MANIPULAPY_FORCE_CPU=1`
    pins it to False (reference tests/conftest.py:85).
    """
    global _probe_result
    with _probe_lock:
        if _probe_result is None:
            if os.environ.get("MANIPULAPY_FORCE_CPU") == "1":
                _probe_result = False
            else:
                try:
                    _probe_result = _hip.device_count() > 0
                except _hip.HipUnavailableError:
                    _probe_result = False
        return _probe_result


def _refuse_silent_cpu(name: str) -> None:
    """The CPU launchers serve the NumPy backend.  With the "hip" backend ACTIVE and no usable device the reference's rule
    would quietly hand the work to the CPU; here that is an error (a GPU box whose HIP path is broken must not pass as
    working), unless the caller pinned the CPU on purpose with MANIPULAPY_FORCE_CPU=1 (reference tests/conftest.py:85)."""
    if getattr(get_backend(), "gpu_capable", False) and os.environ.get("MANIPULAPY_FORCE_CPU") != "1":
        raise _hip.HipUnavailableError(
            f"'{name}': the 'hip' backend is active but no MI355X is usable in this process; refusing to compute on the CPU "
            "silently (select the NumPy backend, or set MANIPULAPY_FORCE_CPU=1, to run the CPU launchers on purpose)")


def _reset_probe_for_tests(value: Optional[bool] = None) -> None:
    global _probe_result
    with _probe_lock:
        _probe_result = value


def _hip_routing_enabled(hip_available: Optional[bool] = None) -> bool:
    """GPU launchers run iff a GPU is physically there AND the active backend is gpu_capable
    (reference `_cuda_routing_enabled`, registry.py:729-732)."""
    physical = check_hip_availability() if hip_available is None else hip_available
    return bool(physical and getattr(get_backend(), "gpu_capable", False))


def default_device_id() -> int:
    for key in (HIP_DEVICE_ENV, "LOCAL_RANK"):
        v = os.environ.get(key)
        if v is not None and v.strip() != "":
            return int(v)
    return 0


def get_context() -> _hip.HipContext:
    """The process-wide device context (one process drives one GPU).  Raises without a GPU."""
    global _ctx
    with _probe_lock:
        if _ctx is None:
            _ctx = _hip.HipContext(default_device_id())
        return _ctx


def get_gpu_properties() -> Optional[Dict[str, Any]]:
    """reference registry.py:335-356; None when no GPU."""
    if not check_hip_availability():
        return None
    return get_context().properties()


# ------------------------------------------------------------------------------ host-side NumPy path
def trajectory_cpu(thetastart, thetaend, Tf: float, N: int, method: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Time-scaled point-to-point trajectory on the host (NumPy backend plumbing, BASELINE config 0).

    Same arithmetic as the HIP kernel and the reference's numba loop (planning/trajectory.py:45-73):
    float32 endpoints, float64 polynomial in tau = idx / (N - 1), float32 store; cubic / quintic,
    any other `method` -> zeros.  No joint-limit clip here (the planner applies it).
    """
    a = np.asarray(thetastart, dtype=np.float32)
    b = np.asarray(thetaend, dtype=np.float32)
    n = a.shape[-1]
    N = int(N)
    if N <= 0:
        z = np.zeros(a.shape[:-1] + (0, n), dtype=np.float32)
        return z, z.copy(), z.copy()
    k = np.arange(N, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        tau = (k * (float(Tf) / (N - 1)) if N > 1 else k * np.inf) / float(Tf)
    if method == 3:
        s = 3.0 * tau * tau - 2.0 * tau * tau * tau
        sd = 6.0 * tau * (1.0 - tau) / Tf
        sdd = 6.0 / (Tf * Tf) * (1.0 - 2.0 * tau)
    elif method == 5:
        t2 = tau * tau
        t3 = t2 * tau
        t4 = t2 * t2
        t5 = t4 * tau
        s = 10.0 * t3 - 15.0 * t4 + 6.0 * t5
        sd = (30.0 * t2 - 60.0 * t3 + 30.0 * t4) / Tf
        sdd = (60.0 * tau - 180.0 * t2 + 120.0 * t3) / (Tf * Tf)
    else:
        s = sd = sdd = np.zeros(N)
    d = (b - a).astype(np.float64)[..., None, :]
    a64 = a.astype(np.float64)[..., None, :]
    pos = (s[:, None] * d + a64).astype(np.float32)
    vel = (sd[:, None] * d).astype(np.float32)
    acc = (sdd[:, None] * d).astype(np.float32)
    return pos, vel, acc


def potential_field_cpu(positions, goal, obstacles, influence_distance: float):
    """Fused attractive + repulsive potential field on the host (NumPy launcher of "potential_field.fused").

    U = 1/2 |p - goal|^2 + sum over obstacles with 0 < d < d0 of 1/2 (1/d - 1/d0)^2, and its gradient; float32
    arithmetic, zero-distance obstacles ignored (reference cuda_kernels/field_kernels.py:113-161)."""
    p = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
    g = np.ascontiguousarray(goal, dtype=np.float32).reshape(3)
    o = np.ascontiguousarray(obstacles, dtype=np.float32).reshape(-1, 3)
    diff = p - g
    pot = np.float32(0.5) * np.einsum("ij,ij->i", diff, diff)
    grad = diff.copy()
    if o.shape[0]:
        inv_d0 = np.float32(1.0 / influence_distance) if influence_distance > 0.0 else np.float32(0.0)
        d0sq = np.float32(influence_distance * influence_distance)
        rel = p[:, None, :] - o[None, :, :]                      # (P, O, 3)
        d2 = np.einsum("poj,poj->po", rel, rel)
        hit = (d2 > 0.0) & (d2 < d0sq)
        inv = np.zeros_like(d2)
        inv[hit] = np.float32(1.0) / np.sqrt(d2[hit])
        t = np.where(hit, inv - inv_d0, np.float32(0.0)).astype(np.float32)
        pot = pot + np.float32(0.5) * np.sum(t * t, axis=1, dtype=np.float32)
        grad = grad + np.einsum("po,poj->pj", -t * inv * inv * inv, rel).astype(np.float32)
    return pot.astype(np.float32), grad.astype(np.float32)


# ------------------------------------------------------------------------------ launchers
def _no_cpu(name: str) -> Callable[..., Any]:
    def launcher(*_a: Any, **_k: Any) -> Any:
        raise BackendNotSupportedError(
            f"'{name}' exists only on the HIP path of manipulapy_amd: it needs set_backend('hip') AND a "
            "visible MI355X (no CPU twin is shipped for this operation)")
    return launcher


# CPU launchers of the dynamics / kinematics operations: the C ABI's *_cpu twins (csrc/mp_cpu.cpp - the same per-row
# templates the kernels instantiate, on host threads).  The registry picks them by the reference's rule (NumPy backend
# active, registry.py:85-89); they are not a fallback of a failing GPU launch unless MANIPULAPY_HIP_FALLBACK=1 asks for the reference's behaviour.
def _launch_id_cpu(model, q, qd, qdd, g=None, Ftip=None, dtype=np.float32):
    return _hip.cpu_id_trajectory(model, q, qd, qdd, g, Ftip, dtype=dtype)


def _launch_fused_cpu(model, start_batch, end_batch, Tf, N, method, g=None, Ftip=None):
    """joint_trajectory -> inverse_dynamics_trajectory for B start / end pairs: NumPy generation (the reference's own
    CPU arithmetic, float32 rows, positions clipped to the model's float32 joint limits) then the float32 CPU twin."""
    sb = np.asarray(start_batch, dtype=np.float32)
    pos, vel, acc = trajectory_cpu(sb, np.asarray(end_batch, dtype=np.float32), float(Tf), int(N), int(method))
    lim = model.joint_limits_f32()
    pos = np.clip(pos, lim[:, 0], lim[:, 1])
    n = sb.shape[1]
    tau = _hip.cpu_id_trajectory(model, pos.reshape(-1, n), vel.reshape(-1, n), acc.reshape(-1, n), g, Ftip, dtype=np.float32)
    return tau.reshape(sb.shape[0], int(N), n)


def _launch_fk_jac_cpu(model, q, qd=None, qdd=None, g=None, Ftip=None, want_T=True, want_J=True):
    return _hip.cpu_fk_jac_id(model, q, qd, qdd, g, Ftip, want_T, want_J)


def _launch_mass_matrix_cpu(model, q):
    return _hip.cpu_mass_matrix(model, q)


def _launch_forward_dynamics_cpu(model, q, qd, tau, g=None, Ftip=None):
    return _hip.cpu_forward_dynamics(model, q, qd, tau, g, Ftip)


def _launch_fd_trajectory_cpu(model, theta0, dtheta0, taumat, g, Ftipmat, dt, intRes, dtype=np.float64, layout="batch_major",
                              device_layout=None):
    del device_layout   # (a device-side choice; the host rows are walked in whatever order they come)
    if layout == "time_major":   # host arrays (N, B, *): the CPU rows are batch-major
        sw = lambda a: None if a is None else np.ascontiguousarray(np.swapaxes(np.asarray(a), 0, 1))
        out = _hip.cpu_fd_trajectory(model, theta0, dtheta0, sw(taumat), g, sw(Ftipmat), dt, intRes, dtype=dtype)
        return tuple(sw(o) for o in out)
    return _hip.cpu_fd_trajectory(model, theta0, dtheta0, taumat, g, Ftipmat, dt, intRes, dtype=dtype)


def _launch_cartesian_cpu(Xstart, Xend, Tf, N, method):
    return _hip.cpu_cartesian_trajectory(Xstart, Xend, Tf, N, method)


def _launch_trajectory_gpu(model, thetastart, thetaend, Tf, N, method, use_pinned=True, *, variant="auto",
                           enable_monitoring=True):
    """(pos, vel, acc) host float32 (N, n), positions clipped to the model's joint limits.
    Signature follows reference registry.py:828-851 with the compiled model prepended."""
    del use_pinned, variant, enable_monitoring
    s = np.asarray(thetastart, dtype=np.float32)[None, :]
    e = np.asarray(thetaend, dtype=np.float32)[None, :]
    p, v, a = get_context().batch_trajectory_host(model, s, e, Tf, N, method)
    return p[0], v[0], a[0]


def _launch_trajectory_cpu(model, thetastart, thetaend, Tf, N, method, use_pinned=True, *, variant="auto",
                           enable_monitoring=True):
    del model, use_pinned, variant, enable_monitoring
    return trajectory_cpu(thetastart, thetaend, Tf, N, method)


def _launch_batch_trajectory_gpu(model, start_batch, end_batch, Tf, N, method):
    return get_context().batch_trajectory_host(model, start_batch, end_batch, Tf, N, method)


def _launch_batch_trajectory_cpu(model, start_batch, end_batch, Tf, N, method):
    del model
    return trajectory_cpu(start_batch, end_batch, Tf, N, method)


def _launch_id_gpu(model, q, qd, qdd, g=None, Ftip=None, dtype=np.float32):
    return get_context().id_trajectory_host(model, q, qd, qdd, g, Ftip, dtype=dtype)


def _launch_fused_gpu(model, start_batch, end_batch, Tf, N, method, g=None, Ftip=None):
    return get_context().traj_id_fused_host(model, start_batch, end_batch, Tf, N, method, g, Ftip)


def _launch_fk_jac_gpu(model, q, qd=None, qdd=None, g=None, Ftip=None, want_T=True, want_J=True):
    return get_context().fk_jac_id_host(model, q, qd, qdd, g, Ftip, want_T, want_J)


# ---- profiling is a flag of the shared context: planners that asked for it hold a reference, the last one out switches it off
_profiling_users = 0
_profiling_lock = threading.Lock()


def acquire_profiling() -> None:
    global _profiling_users
    with _profiling_lock:
        _profiling_users += 1
        if _profiling_users == 1:
            get_context().set_profiling(True)


def release_profiling() -> None:
    global _profiling_users
    with _profiling_lock:
        if _profiling_users == 0:
            return
        _profiling_users -= 1
        if _profiling_users == 0 and _ctx is not None and getattr(_ctx, "handle", None) is not None:
            try:
                _ctx.set_profiling(False)
            except Exception:  # interpreter shutdown: the context may already be gone
                pass


def _launch_ik_gpu(model, T_desired, theta0, **kw):
    return get_context().inverse_kinematics_host(model, T_desired, theta0, **kw)


def _launch_ik_cpu(model, T_desired, theta0, **kw):
    return _hip.cpu_inverse_kinematics(model, T_desired, theta0, **kw)


def _launch_pd_regulation_gpu(model, theta0, theta_des, Kp, Kd, g, dt, steps):
    return get_context().pd_regulation_host(model, theta0, theta_des, Kp, Kd, g, dt, steps)


def _launch_pd_regulation_cpu(model, theta0, theta_des, Kp, Kd, g, dt, steps):
    return _hip.cpu_pd_regulation(model, theta0, theta_des, Kp, Kd, g, dt, steps)


def _launch_mass_matrix_gpu(model, q):
    return get_context().mass_matrix_host(model, q)


def _launch_forward_dynamics_gpu(model, q, qd, tau, g=None, Ftip=None):
    return get_context().forward_dynamics_host(model, q, qd, tau, g, Ftip)


def _launch_fd_trajectory_gpu(model, theta0, dtheta0, taumat, g, Ftipmat, dt, intRes, dtype=np.float64, layout="batch_major",
                              device_layout=None):
    return get_context().fd_trajectory_host(model, theta0, dtheta0, taumat, g, Ftipmat, dt, intRes, dtype=dtype, layout=layout,
                                            device_layout=device_layout)


def _launch_cartesian_gpu(Xstart, Xend, Tf, N, method):
    return get_context().cartesian_trajectory_host(Xstart, Xend, Tf, N, method)


def _launch_potential_field_gpu(positions, goal, obstacles, influence_distance, use_pinned=True):
    del use_pinned
    return get_context().potential_field_host(positions, goal, obstacles, influence_distance)


def _launch_potential_field_cpu(positions, goal, obstacles, influence_distance, use_pinned=True):
    del use_pinned
    return potential_field_cpu(positions, goal, obstacles, influence_distance)


def _grid_1d(rows: int, block: int = 256):
    """Launch shape every kernel uses: one thread per (trajectory, timestep) row, 256-thread blocks
    (4 wavefronts of 64).  Replaces the CUDA block heuristics of reference registry.py:409-515."""
    rows = int(rows)
    return ((max(rows, 0) + block - 1) // block,), (block,)


_TRAJECTORY_VARIANTS = ("auto", "auto_tune", "standard", "vectorized", "memory_optimized", "warp_optimized",
                        "cache_friendly")


def _bind_variant(function: Callable[..., Any], variant: str) -> Callable[..., Any]:
    def bound(*args: Any, **kwargs: Any) -> Any:
        kwargs.setdefault("variant", variant)
        return function(*args, **kwargs)
    return bound


def _build_kernel_registry() -> KernelRegistry:
    reg = KernelRegistry()
    # the reference's seven trajectory variant names stay valid; on gfx950 they are one kernel
    for variant in _TRAJECTORY_VARIANTS:
        reg.register(KernelRegistration(
            name=f"trajectory.{variant}", implementation="mp_batch_trajectory_host_f32",
            launch_config=_grid_1d, cpu_fallback=trajectory_cpu,
            gpu_launcher=_bind_variant(_launch_trajectory_gpu, variant),
            cpu_launcher=_bind_variant(_launch_trajectory_cpu, variant),
            metadata={"family": "trajectory", "variant": variant, "dimensions": 1}))
    reg.register(KernelRegistration(
        name="trajectory.batch", implementation="mp_batch_trajectory_host_f32", launch_config=_grid_1d,
        cpu_fallback=trajectory_cpu, gpu_launcher=_launch_batch_trajectory_gpu,
        cpu_launcher=_launch_batch_trajectory_cpu,
        metadata={"family": "trajectory", "variant": "batch", "dimensions": 1}))
    reg.register(KernelRegistration(
        name="potential_field.fused", implementation="mp_potential_field_host_f32", launch_config=_grid_1d,
        cpu_fallback=potential_field_cpu, gpu_launcher=_launch_potential_field_gpu, cpu_launcher=_launch_potential_field_cpu,
        metadata={"family": "potential_field", "variant": "fused", "dimensions": 1}))
    for name, impl, gpu, cpu in (
        ("dynamics.inverse_trajectory", "mp_id_trajectory_host_f32 / _f64", _launch_id_gpu, _launch_id_cpu),
        ("dynamics.fused_trajectory_inverse", "mp_traj_id_fused_host_f32", _launch_fused_gpu, _launch_fused_cpu),
        ("kinematics.fk_jacobian", "mp_fk_jac_id_host_f64", _launch_fk_jac_gpu, _launch_fk_jac_cpu),
        ("kinematics.inverse", "mp_inverse_kinematics_host_f64", _launch_ik_gpu, _launch_ik_cpu),
        ("dynamics.mass_matrix", "mp_mass_matrix_host_f64", _launch_mass_matrix_gpu, _launch_mass_matrix_cpu),
        ("dynamics.forward", "mp_forward_dynamics_host_f64", _launch_forward_dynamics_gpu, _launch_forward_dynamics_cpu),
        ("dynamics.forward_trajectory", "mp_fd_trajectory_host_f32 / _f64", _launch_fd_trajectory_gpu, _launch_fd_trajectory_cpu),
        ("trajectory.cartesian", "mp_cartesian_trajectory_host_f32", _launch_cartesian_gpu, _launch_cartesian_cpu),
        ("control.pd_regulation", "mp_pd_regulation_host_f64", _launch_pd_regulation_gpu, _launch_pd_regulation_cpu),
    ):
        reg.register(KernelRegistration(
            name=name, implementation=impl, launch_config=_grid_1d, cpu_fallback=cpu, gpu_launcher=gpu,
            cpu_launcher=cpu or _no_cpu(name), metadata={"family": name.split(".")[0], "variant": "hip", "dimensions": 1}))
    return reg


_KERNEL_REGISTRY = _build_kernel_registry()


def get_registered_kernel(name: str) -> KernelRegistration:
    return _KERNEL_REGISTRY.get(name)


def execute_registered_kernel(name: str, *args: Any, **kwargs: Any) -> Any:
    """Run a registered operation on the GPU (hip backend + device present) or through its explicit
    CPU launcher (reference registry.py:963-965)."""
    return _KERNEL_REGISTRY.execute(name, *args, **kwargs)
