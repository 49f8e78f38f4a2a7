"""OptimizedTrajectoryPlanning — host-side mirror of the reference planner facade for the hot path.

Reference: ManipulaPy/planning/trajectory_planning.py:116-399 (constructor, routing),
planning/trajectory.py:103-502 (joint / batch trajectories), planning/trajectory_dynamics.py:31-90,
:308-380 (inverse_dynamics_trajectory), :710-735 (calculate_derivatives).

What is kept: the constructor signature (positional + keyword-only arguments), float32 limits,
`performance_stats`, the `_should_use_gpu` routing rule (forced CPU pin -> live routing predicate ->
work threshold), result dtypes / shapes (float32 (N, n) / (B, N, n)), joint-limit and torque-limit
clipping, default gravity / zero wrench.

What is different, on purpose:
  * a failing GPU launch raises instead of silently recomputing on the CPU
    (reference planning/trajectory.py:270-274, trajectory_dynamics.py:292-302);
  * with the NumPy backend active (or use_cuda=False) every operation runs its registered CPU launcher
    (NumPy for trajectory generation, the C ABI's *_cpu twins for the dynamics), as the reference's
    planner runs its _*_cpu methods; with the "hip" backend active and no usable GPU it raises instead
    of quietly computing on the host;
  * with the "hip" backend the dynamics always go to the device (the reference applies its work
    threshold to them too; a launch costs microseconds here, so the threshold only steers trajectory
    generation, whose NumPy path is BASELINE config 0);
  * collision avoidance: the mesh-less checker (never a collision, like the reference's without mesh files) and `plan_trajectory`'s
    potential-field waypoint push are in; mesh loading is not;
  * `batch_inverse_dynamics_trajectory` is new: joint_trajectory -> inverse_dynamics_trajectory fused
    on the device for B start/end pairs; `batch_forward_dynamics_trajectory` is new: B roll-outs of
    forward_dynamics_trajectory (planning/trajectory_dynamics.py:580-708) in one launch.
"""
from __future__ import annotations

import logging
import os
import time
import warnings
from typing import Dict, Optional, Tuple

import numpy as np

from . import _hip
from . import registry as _reg
from .backend import get_backend

__all__ = ["OptimizedTrajectoryPlanning", "TrajectoryPlanning"]

logger = logging.getLogger("ManipulaPy.planning.trajectory_planning")  # name kept


class OptimizedTrajectoryPlanning:
    def __init__(self, serial_manipulator, urdf_path, dynamics, joint_limits, torque_limits=None, *,
                 use_cuda: Optional[bool] = None, cuda_threshold: int = 10, memory_pool_size_mb: Optional[int] = None,
                 enable_profiling: bool = False, auto_optimize: bool = True, kernel_type: str = "auto",
                 target_speedup: float = 40.0) -> None:
        self.serial_manipulator = serial_manipulator
        self.dynamics = dynamics
        self.urdf_path = urdf_path
        self.joint_limits = np.asarray(joint_limits, dtype=np.float32)
        self.torque_limits = (np.asarray(torque_limits, dtype=np.float32) if torque_limits is not None
                              else np.array([[-np.inf, np.inf]] * len(joint_limits), dtype=np.float32))
        self.kernel_type = kernel_type if kernel_type is not None else "auto"
        self.target_speedup = target_speedup if target_speedup is not None else 40.0
        self.enable_profiling = bool(enable_profiling)
        # collision helpers as the reference sets them up (planning/trajectory_planning.py:229-238): both, or neither when the
        # URDF cannot be read.  The checker is the mesh-less one (potential_field.py): it never reports a collision.
        try:
            from .potential_field import CollisionChecker, PotentialField

            self.collision_checker = CollisionChecker(urdf_path)
            self.potential_field = PotentialField()
        except Exception as exc:
            logger.warning("Could not initialise collision checker: %s", exc)
            self.collision_checker = None
            self.potential_field = None
        self._last_cpu_time = 0.0
        self.performance_stats = {"gpu_calls": 0, "cpu_calls": 0, "total_gpu_time": 0.0, "total_cpu_time": 0.0,
                                  "memory_transfers": 0, "kernel_launches": 0, "speedup_achieved": 0.0,
                                  "best_kernel_used": "none"}
        self.performance_stats.update({"gpu_kernel_ms_total": 0.0, "gpu_kernel_ms_last": 0.0, "gpu_timed_calls": 0})
        del auto_optimize, memory_pool_size_mb  # CUDA-environment knobs with no HIP counterpart

        physical = _reg.check_hip_availability()
        detected = _reg._hip_routing_enabled(physical)
        self._physical_cuda = physical
        self._forced_cpu = use_cuda is False
        if use_cuda is None:
            self.cuda_available = detected
        elif use_cuda and not detected:
            raise RuntimeError("use_cuda=True requested but no GPU-capable backend with a HIP device is active. "
                               "Select the 'hip' backend on a machine with an MI355X.")
        else:
            self.cuda_available = bool(use_cuda)
        self.gpu_properties = _reg.get_gpu_properties() if self.cuda_available else None
        if self.cuda_available and self.gpu_properties:
            cus = self.gpu_properties["multiprocessor_count"]
            per_cu = 1000 if self.target_speedup >= 40 else 500
            self.cpu_threshold = max(int(cuda_threshold), int(cus * per_cu / len(joint_limits)))
        else:
            self.cpu_threshold = int(cuda_threshold)
        self._model = None
        # reference planning/trajectory_planning.py:295-296 (profile_start): here a timed HIP event pair and an roctx
        # range around every launch of the context, read back into performance_stats after each GPU call
        self._profiling_held = False
        self._prof_base = {"kernel_ms_total": 0.0, "timed_calls": 0}
        if self.enable_profiling and self.cuda_available and self._gpu_routed():
            _reg.acquire_profiling()          # released by close() / __del__: the flag belongs to the shared context
            self._profiling_held = True
            self._prof_base = _reg.get_context().profile()   # this planner reports what was timed SINCE, not the context's totals

    # ------------------------------------------------------------------ plumbing
    def _hip_model(self):
        """Dynamics tables + THIS planner's float32 joint / torque limits, compiled once."""
        if self._model is None:
            self._model = self.dynamics.hip_model(self.joint_limits.astype(np.float64), self.torque_limits.astype(np.float64))
            if self._gpu_routed() and os.environ.get("MANIPULAPY_HIP_SPECIALIZE", "1") != "0" and self._model.n <= _hip.MP_MAX_DOF:
                # float32 kernels with this robot's constants baked in (hiprtc, ~1.5 s once, cached on disk);
                # purely an optimisation: the generic kernels compute the same values
                try:
                    _reg.get_context().specialize(self._model)
                except Exception as exc:  # pragma: no cover - depends on the hiprtc installation
                    logger.warning("kernel specialisation unavailable (%s); using the generic kernels", exc)
        return self._model

    def _should_use_gpu(self, N: int, num_joints: int) -> bool:
        """reference planning/trajectory_planning.py:356-399."""
        if getattr(self, "_forced_cpu", False):
            return False
        physical = getattr(self, "_physical_cuda", self.cuda_available)
        if not _reg._hip_routing_enabled(physical):
            return False
        return N * num_joints >= self.cpu_threshold

    def _gpu_routed(self) -> bool:
        """Routing for operations that exist only on the device (no work threshold)."""
        if getattr(self, "_forced_cpu", False):
            return False
        return _reg._hip_routing_enabled(getattr(self, "_physical_cuda", self.cuda_available))

    def _count(self, kind: str, t0: float) -> None:
        dt = time.time() - t0
        self.performance_stats[f"{kind}_calls"] += 1
        self.performance_stats[f"total_{kind}_time"] += dt
        if kind == "gpu":
            self.performance_stats["kernel_launches"] += 1
            self.performance_stats["best_kernel_used"] = "hip"
            if self.enable_profiling and self._profiling_held:
                p = _reg.get_context().profile()
                self.performance_stats.update({"gpu_kernel_ms_total": p["kernel_ms_total"] - self._prof_base["kernel_ms_total"],
                                               "gpu_kernel_ms_last": p["kernel_ms_last"],
                                               "gpu_timed_calls": p["timed_calls"] - self._prof_base["timed_calls"]})
        else:
            self._last_cpu_time = dt

    def _dispatch(self, name: str, *args, **kwargs):
        """Run a registered operation where the routing rule sends it: the GPU launcher when the "hip" backend is active
        and a device is there, otherwise the CPU launcher (NumPy backend, or this planner pinned to the CPU with
        use_cuda=False).  "hip" backend without a usable device: raises (registry._refuse_silent_cpu)."""
        entry = _reg.get_registered_kernel(name)
        t0 = time.time()
        if self._gpu_routed():
            out, route = _reg.run_gpu_launcher(entry, *args, **kwargs)   # "cpu" only under MANIPULAPY_HIP_FALLBACK=1 after a failed launch
            self._count(route, t0)
        else:
            if not self._forced_cpu:
                _reg._refuse_silent_cpu(name)
            out = entry.cpu_launcher(*args, **kwargs)
            self._count("cpu", t0)
        return out

    def _clip_positions(self, pos: np.ndarray) -> np.ndarray:
        return np.clip(pos, self.joint_limits[:, 0], self.joint_limits[:, 1])

    # ------------------------------------------------------------------ trajectories
    def joint_trajectory(self, thetastart, thetaend, Tf, N, method, kernel_type=None, enable_monitoring=None) -> Dict[str, np.ndarray]:
        """positions / velocities / accelerations, each (N, n) float32 (reference planning/trajectory.py:103-169)."""
        del enable_monitoring
        backend = get_backend()
        start = np.array(backend.to_numpy(backend.asarray(thetastart)), dtype=np.float32)
        end = np.array(backend.to_numpy(backend.asarray(thetaend)), dtype=np.float32)
        t0 = time.time()
        if self._should_use_gpu(int(N), len(start)):
            variant = kernel_type or self.kernel_type or "auto"
            entry = _reg.get_registered_kernel(f"trajectory.{variant}")  # fail-closed on unknown names
            (pos, vel, acc), route = _reg.run_gpu_launcher(entry, self._hip_model(), start, end, Tf, int(N), int(method))
            self._count(route, t0)
        else:
            pos, vel, acc = _reg.trajectory_cpu(start, end, float(Tf), int(N), int(method))
            pos = self._clip_positions(pos)
            self._count("cpu", t0)
        return {"positions": backend.asarray(pos), "velocities": backend.asarray(vel), "accelerations": backend.asarray(acc)}

    def batch_joint_trajectory(self, thetastart_batch, thetaend_batch, Tf, N, method, kernel_type=None) -> Dict[str, np.ndarray]:
        """(B, N, n) float32 arrays (reference planning/trajectory.py:335-502)."""
        del kernel_type
        backend = get_backend()
        sb = np.asarray(backend.to_numpy(backend.asarray(thetastart_batch)), dtype=np.float32)
        eb = np.asarray(backend.to_numpy(backend.asarray(thetaend_batch)), dtype=np.float32)
        if sb.ndim != 2 or sb.shape != eb.shape:
            raise ValueError(f"start/end batches must both be (B, n); got {sb.shape} and {eb.shape}")
        B, n = sb.shape
        t0 = time.time()
        if B == 0:
            z = np.zeros((0, int(N), n), dtype=np.float32)
            return {"positions": z, "velocities": z.copy(), "accelerations": z.copy()}
        if self._gpu_routed():
            (pos, vel, acc), route = _reg.run_gpu_launcher(_reg.get_registered_kernel("trajectory.batch"), self._hip_model(), sb, eb, Tf,
                                                           int(N), int(method))
            self._count(route, t0)
        else:
            pos, vel, acc = _reg.trajectory_cpu(sb, eb, float(Tf), int(N), int(method))
            pos = self._clip_positions(pos)
            self._count("cpu", t0)
        return {"positions": backend.asarray(pos), "velocities": backend.asarray(vel), "accelerations": backend.asarray(acc)}

    def cartesian_trajectory(self, Xstart, Xend, Tf, N, method) -> Dict[str, np.ndarray]:
        """Straight-line Cartesian trajectory between two SE(3) poses (reference planning/trajectory.py:504-594):
        positions / velocities / accelerations (N, 3) and orientations (N, 3, 3), float32."""
        Xs, Xe = np.asarray(Xstart, dtype=np.float64), np.asarray(Xend, dtype=np.float64)
        N = int(N)
        if N < 0:
            raise ValueError("negative dimensions are not allowed")
        if N == 0:  # the reference's empty shapes (:555-559, :727-730)
            return {"positions": np.zeros((0,), np.float32), "velocities": np.zeros((0, 3), np.float32),
                    "accelerations": np.zeros((0, 3), np.float32), "orientations": np.zeros((0, 3, 3), np.float32)}
        if N == 1:
            raise ZeroDivisionError("float division by zero")  # timegap = Tf / (N - 1.0), as in the reference
        r = self.batch_cartesian_trajectory(Xs[None], Xe[None], Tf, N, method)
        return {k: v[0] for k, v in r.items()}

    def batch_cartesian_trajectory(self, Xstart_batch, Xend_batch, Tf, N, method) -> Dict[str, np.ndarray]:
        """B pose pairs (B, 4, 4) in one launch -> (B, N, 3) / (B, N, 3, 3) float32 arrays.  New."""
        pos, vel, acc, ori = self._dispatch("trajectory.cartesian", Xstart_batch, Xend_batch, Tf, int(N), int(method))
        b = get_backend()
        return {"positions": b.asarray(pos), "velocities": b.asarray(vel), "accelerations": b.asarray(acc),
                "orientations": b.asarray(ori)}

    # ------------------------------------------------------------------ dynamics over trajectories
    def inverse_dynamics_trajectory(self, thetalist_trajectory, dthetalist_trajectory, ddthetalist_trajectory,
                                    gravity_vector=None, Ftip=None) -> np.ndarray:
        """(rows, n) float32 torques, clipped to the torque limits
        (reference planning/trajectory_dynamics.py:31-90, :308-380).  Rows are independent, so a
        flattened (B*N, n) history is a valid input.  float64 inputs are evaluated in float64 and
        then stored float32 exactly as the reference does (:354); float32 inputs run the float32 kernel."""
        if gravity_vector is None:
            gravity_vector = np.array([0.0, 0.0, -9.81])
        if Ftip is None:
            Ftip = [0, 0, 0, 0, 0, 0]
        q = np.asarray(thetalist_trajectory)
        if q.ndim != 2:
            raise ValueError(f"trajectories must be (N, n); got {q.shape}")
        if q.shape[0] == 0:
            return np.zeros(q.shape, dtype=np.float32)
        if getattr(self.dynamics, "_legacy", False):
            return get_backend().asarray(self._legacy_inverse_dynamics(q, np.asarray(dthetalist_trajectory),
                                                                       np.asarray(ddthetalist_trajectory), gravity_vector, Ftip))
        dtype = np.float64 if q.dtype == np.float64 else np.float32
        tau = self._dispatch("dynamics.inverse_trajectory", self._hip_model(), q, dthetalist_trajectory,
                             ddthetalist_trajectory, gravity_vector, Ftip, dtype=dtype)
        return get_backend().asarray(tau.astype(np.float32, copy=False))

    def batch_inverse_dynamics_trajectory(self, thetastart_batch, thetaend_batch, Tf, N, method, gravity_vector=None,
                                          Ftip=None) -> np.ndarray:
        """(B, N, n) float32 torques of the time-scaled trajectories between B start/end pairs, i.e.
        inverse_dynamics_trajectory(**batch_joint_trajectory(...)) without materialising the histories."""
        sb = np.asarray(thetastart_batch, dtype=np.float32)
        eb = np.asarray(thetaend_batch, dtype=np.float32)
        if sb.ndim != 2 or sb.shape != eb.shape:
            raise ValueError(f"start/end batches must both be (B, n); got {sb.shape} and {eb.shape}")
        if sb.shape[0] == 0:
            return np.zeros((0, int(N), sb.shape[1]), dtype=np.float32)
        tau = self._dispatch("dynamics.fused_trajectory_inverse", self._hip_model(), sb, eb, Tf, int(N), int(method),
                             gravity_vector, Ftip)
        return get_backend().asarray(tau)

    def forward_dynamics_trajectory(self, thetalist, dthetalist, taumat, g, Ftipmat, dt, intRes) -> Dict[str, np.ndarray]:
        """Semi-implicit Euler roll-out of ONE trajectory (reference planning/trajectory_dynamics.py:382-423,
        :580-708): positions / velocities / accelerations, each (N, n) float32.  The state is integrated in the
        dtype of `thetalist` (float32 stays float32, everything else float64), as the reference does."""
        th = np.asarray(thetalist)
        tm = np.asarray(taumat)
        if tm.ndim != 2:
            raise ValueError(f"taumat must be (N, n); got {tm.shape}")
        if tm.shape[0] == 0:  # the reference seeds row 0 unconditionally (:614-617)
            raise IndexError("index 0 is out of bounds for axis 0 with size 0")
        if getattr(self.dynamics, "_legacy", False):
            b = get_backend()
            return {k: b.asarray(v) for k, v in self._legacy_forward_dynamics(th, np.asarray(dthetalist), tm, g, Ftipmat, dt, intRes).items()}
        Fm = None if Ftipmat is None else np.asarray(Ftipmat)[None]
        r = self.batch_forward_dynamics_trajectory(th[None], np.asarray(dthetalist)[None], tm[None], g, Fm, dt, intRes)
        return {k: v[0] for k, v in r.items()}

    def batch_forward_dynamics_trajectory(self, theta0_batch, dtheta0_batch, taumat_batch, g, Ftipmat_batch, dt, intRes,
                                          layout: str = "batch_major", device_layout: Optional[str] = None) -> Dict[str, np.ndarray]:
        """B independent roll-outs in one launch: theta0 / dtheta0 (B, n), taumat (B, N, n), Ftipmat (B, N, 6) or
        None -> (B, N, n) float32 arrays.  New (the reference integrates one trajectory per call).

        layout="time_major": taumat / Ftipmat and the three results are (N, B, *) - the layout of the faster device kernel
        (mp_fd_trajectory_tm_*: every step touches whole cache lines), for callers that build their histories step by step.
        device_layout ("batch_major" / "time_major") picks the kernel independently of the host layout (converted on the
        device); from host arrays the call is PCIe-bound either way (profiles/r03_time_ops.jsonl), so the default is the
        host layout's own kernel."""
        if layout not in ("batch_major", "time_major"):
            raise ValueError("layout must be 'batch_major' or 'time_major'")
        th = np.asarray(theta0_batch)
        if th.ndim != 2:
            raise ValueError(f"initial states must be (B, n); got {th.shape}")
        dtype = np.float32 if th.dtype == np.float32 else np.float64
        if int(intRes) == 0:
            raise ZeroDivisionError("float division by zero")  # dt_step = dt / intRes, as in the reference (:627)
        if int(intRes) < 0:
            raise ValueError("intRes must be positive")
        if g is None:
            g = np.array([0.0, 0.0, -9.81])
        pos, vel, acc = self._dispatch("dynamics.forward_trajectory", self._hip_model(), th, dtheta0_batch, taumat_batch, g,
                                       Ftipmat_batch, dt, int(intRes), dtype=dtype, layout=layout, device_layout=device_layout)
        b = get_backend()
        return {"positions": b.asarray(pos), "velocities": b.asarray(vel), "accelerations": b.asarray(acc)}

    # ------------------------------------------------------------------ legacy dynamics objects (Mlist_per_link=None)
    # The reference's approximation for such objects is not rigid-body dynamics (dynamics/mass_matrix.py:101-132), so there
    # is no compiled model and no kernel for it: the planner walks the rows on the host exactly as the reference's CPU
    # paths do (planning/trajectory_dynamics.py:308-380, :580-708), under every backend.
    def _legacy_inverse_dynamics(self, q, qd, qdd, g, Ftip) -> np.ndarray:
        t0 = time.time()
        rows = []
        with warnings.catch_warnings():   # one warning for the call instead of one per row and term
            warnings.simplefilter("ignore")
            for i in range(q.shape[0]):
                try:
                    rows.append(np.asarray(self.dynamics.inverse_dynamics(q[i], qd[i], qdd[i], g, Ftip), dtype=np.float32))
                except Exception as exc:  # the reference's per-row semantics: a row that raises becomes zeros (:345-358)
                    logger.warning("Error in inverse dynamics at point %d: %s", i, exc)
                    rows.append(np.zeros(q.shape[1], dtype=np.float32))
        warnings.warn("inverse_dynamics_trajectory on a ManipulatorDynamics without Mlist_per_link \u2014 using the reference's "
                      "legacy approximation on the host (incorrect for non-trivial robots).", stacklevel=3)
        tau = np.clip(np.stack(rows), self.torque_limits[:, 0], self.torque_limits[:, 1])
        self._count("cpu", t0)
        return tau

    def _legacy_forward_dynamics(self, theta0, dtheta0, taumat, g, Ftipmat, dt, intRes) -> Dict[str, np.ndarray]:
        t0 = time.time()
        N, n = taumat.shape[0], theta0.shape[0]
        th, dth = np.array(theta0), np.array(dtheta0)       # the state keeps the caller's dtype (:619-624)
        lo, hi = self.joint_limits[:, 0], self.joint_limits[:, 1]
        P, V, A = [th.astype(np.float32)], [dth.astype(np.float32)], [np.zeros(n, np.float32)]
        if int(intRes) == 0:
            raise ZeroDivisionError("float division by zero")
        h = dt / intRes
        Fm = np.zeros((N, 6)) if Ftipmat is None else np.asarray(Ftipmat)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for i in range(1, N):
                acc = np.zeros(n, np.float32)
                for _ in range(int(intRes)):
                    try:
                        dd = self.dynamics.forward_dynamics(th, dth, taumat[i], g, Fm[i])
                    except Exception as exc:  # only the dynamics call is tolerated (:640-650)
                        logger.warning("Error in forward dynamics at step %d: %s", i, exc)
                        acc = np.zeros(n)
                        continue
                    dth = (dth + dd * h).astype(dth.dtype)
                    th = np.clip((th + dth * h).astype(th.dtype), lo, hi)
                    acc = dd
                P.append(th.astype(np.float32)); V.append(dth.astype(np.float32)); A.append(np.asarray(acc, dtype=np.float32))
        warnings.warn("forward_dynamics_trajectory on a ManipulatorDynamics without Mlist_per_link \u2014 using the reference's "
                      "legacy approximation on the host (incorrect for non-trivial robots).", stacklevel=3)
        self._count("cpu", t0)
        return {"positions": np.stack(P), "velocities": np.stack(V), "accelerations": np.stack(A)}

    # ------------------------------------------------------------------ helpers kept from the reference
    def plan_trajectory(self, start_position, target_position, obstacle_points):
        """Six joint-space waypoints on the straight line start -> target; with obstacles (points in joint space) and a
        potential field each waypoint is pushed down the field's gradient, 0.01 per step, until the collision checker reports it
        free (at once, with the mesh-less checker) or ten steps have passed (reference planning/collision_host.py:90-152).
        Host arithmetic throughout, like the reference's.  Returns a list of joint lists."""
        start = np.asarray(start_position, dtype=np.float64)
        target = np.asarray(target_position, dtype=np.float64)
        logger.info("Planning trajectory from %d to %d DOF", len(start), len(target))
        num_waypoints = 5
        out = []
        for i in range(num_waypoints + 1):
            alpha = i / num_waypoints
            waypoint = (1 - alpha) * start + alpha * target
            if obstacle_points and self.potential_field:
                obstacles = [np.asarray(o, dtype=np.float64) for o in obstacle_points]
                for _ in range(10):
                    waypoint = waypoint - 0.01 * self.potential_field.compute_gradient(waypoint, target, obstacles)
                    if self.collision_checker and not self.collision_checker.check_collision(waypoint):
                        break
            out.append(waypoint.tolist())
        logger.info("Planned trajectory with %d waypoints", len(out))
        return out

    def calculate_derivatives(self, positions, dt) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """First differences (reference planning/trajectory_dynamics.py:710-735)."""
        p = np.asarray(positions)
        v = (p[1:] - p[:-1]) / dt
        a = (v[1:] - v[:-1]) / dt
        j = (a[1:] - a[:-1]) / dt
        return v, a, j

    # ---- the planner's own timing helpers (reference planning/trajectory_planning.py:526-658, :660-830), without the console tables
    def benchmark_all_kernels(self, N: int = 5000, num_joints: int = 6, num_runs: int = 5) -> Dict[str, Dict[str, object]]:
        """Times `joint_trajectory` under each of the reference's five trajectory-kernel names - on gfx950 they are one kernel, the
        names stay valid registry entries - and returns {name: mean / std / min / max / all_times / success_rate / trajectory_shape};
        {} when the GPU is not routed to (as the reference returns without CUDA)."""
        if not self._gpu_routed():
            logger.warning("GPU not available for benchmarking")
            return {}
        start = np.random.uniform(-1, 1, num_joints).astype(np.float32)
        end = np.random.uniform(-1, 1, num_joints).astype(np.float32)
        results: Dict[str, Dict[str, object]] = {}
        for kernel_type in ("standard", "vectorized", "memory_optimized", "warp_optimized", "cache_friendly"):
            self.reset_performance_stats()
            times, shape = [], None
            for _ in range(num_runs):
                t0 = time.time()
                try:
                    shape = self.joint_trajectory(start, end, 2.0, N, 5, kernel_type=kernel_type, enable_monitoring=False)["positions"].shape
                    times.append(time.time() - t0)
                except Exception as exc:
                    logger.warning("Kernel %s failed: %s", kernel_type, exc)
                    times.append(float("inf"))
            if times and min(times) < float("inf"):
                good = [t for t in times if t < float("inf")]
                results[kernel_type] = {"mean_time": float(np.mean(good)), "std_time": float(np.std(good)), "min_time": float(np.min(good)),
                                        "max_time": float(np.max(good)), "all_times": times, "success_rate": len(good) / len(times),
                                        "trajectory_shape": shape}
        return results

    def benchmark_performance(self, test_cases=None, include_cpu_comparison: bool = True) -> Dict[str, Dict[str, object]]:
        """Times `joint_trajectory` on a list of {"N", "joints", "name"} cases (three runs each; default: four sizes at this
        robot's joint count) and, when the GPU is routed to and asked for, the CPU launcher of the same call for a speed-up."""
        n = len(self.joint_limits)
        if test_cases is None:
            test_cases = [{"N": 100, "joints": n, "name": "Small"}, {"N": 1000, "joints": n, "name": "Medium"},
                          {"N": 5000, "joints": n, "name": "Large"}, {"N": 10000, "joints": n, "name": "Very Large"}]
        results: Dict[str, Dict[str, object]] = {}
        for case in test_cases:
            N, joints, name = case["N"], case["joints"], case["name"]
            start = np.random.uniform(-1, 1, joints).astype(np.float32)
            end = np.random.uniform(-1, 1, joints).astype(np.float32)
            self.reset_performance_stats()
            times = []
            for _ in range(3):
                t0 = time.time()
                traj = self.joint_trajectory(start, end, 2.0, N, 5)
                times.append(time.time() - t0)
            stats = self.get_performance_stats()
            mean = float(np.mean(times))
            results[name] = {"mean_time": mean, "std_time": float(np.std(times)), "min_time": min(times), "max_time": max(times), "N": N,
                             "joints": joints, "stats": stats, "used_gpu": stats["gpu_calls"] > 0,
                             "trajectory_shape": traj["positions"].shape, "speedup_achieved": stats.get("speedup_achieved", 0),
                             "kernel_used": stats.get("best_kernel_used", "unknown"), "elements_per_second": (N * joints) / mean if mean > 0 else 0.0}
            if include_cpu_comparison and results[name]["used_gpu"]:
                old = self.cpu_threshold
                self.cpu_threshold = float("inf")   # the routing rule then picks the CPU launcher
                try:
                    t0 = time.time()
                    self.joint_trajectory(start, end, 2.0, N, 5)
                    cpu_time = time.time() - t0
                finally:
                    self.cpu_threshold = old
                results[name]["cpu_time"] = cpu_time
                results[name]["actual_speedup"] = cpu_time / mean if mean > 0 else 0
            logger.info("%s benchmark: %.4fs, GPU: %s", name, mean, results[name]["used_gpu"])
        return results

    def reset_performance_stats(self) -> None:
        """reference planning/trajectory_planning.py:489-500."""
        self.performance_stats = {"gpu_calls": 0, "cpu_calls": 0, "total_gpu_time": 0.0, "total_cpu_time": 0.0,
                                  "memory_transfers": 0, "kernel_launches": 0, "speedup_achieved": 0.0, "best_kernel_used": "none",
                                  "gpu_kernel_ms_total": 0.0, "gpu_kernel_ms_last": 0.0, "gpu_timed_calls": 0}
        if self.enable_profiling and getattr(self, "_profiling_held", False):
            self._prof_base = _reg.get_context().profile()   # re-base this planner; other users of the context keep their totals

    def close(self) -> None:
        """Give back what this planner holds on the shared context (its profiling reference)."""
        if getattr(self, "_profiling_held", False):
            self._profiling_held = False
            _reg.release_profiling()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def cleanup_gpu_memory(self) -> None:
        """Return the context's pooled device buffers to the driver (reference planning/trajectory_planning.py:502-524)."""
        if self._gpu_routed():
            ctx = _reg.get_context()
            ctx.synchronize()
            ctx.trim_pool()

    def get_performance_stats(self) -> Dict[str, float]:
        """reference planning/trajectory_planning.py:440-487: the counters plus averages, GPU share, overall speed-up and
        the EWMA adaptation of the CPU / GPU work threshold (kept within the reference's [50, 5000] band)."""
        stats = dict(self.performance_stats)
        stats["avg_gpu_time"] = stats["total_gpu_time"] / stats["gpu_calls"] if stats["gpu_calls"] > 0 else 0.0
        stats["avg_cpu_time"] = stats["total_cpu_time"] / stats["cpu_calls"] if stats["cpu_calls"] > 0 else 0.0
        total = stats["gpu_calls"] + stats["cpu_calls"]
        stats["gpu_usage_percent"] = stats["gpu_calls"] / total * 100 if total > 0 else 0.0
        stats["overall_speedup"] = (stats["total_cpu_time"] / stats["total_gpu_time"]
                                    if stats["total_gpu_time"] > 0 and stats["total_cpu_time"] > 0 else 0.0)
        if stats["avg_gpu_time"] > 0 and stats["avg_cpu_time"] > 0:
            ratio = stats["avg_cpu_time"] / stats["avg_gpu_time"]
            self.cpu_threshold = int(0.9 * self.cpu_threshold + 0.1 * ratio * self.cpu_threshold)
            self.cpu_threshold = max(50, min(self.cpu_threshold, 5000))
        return stats


TrajectoryPlanning = OptimizedTrajectoryPlanning  # alias kept by the reference (planning/__init__.py)
