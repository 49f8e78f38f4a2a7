"""SerialManipulator — host-side mirror of ManipulaPy/kinematics/serial_manipulator.py for the hot path.

Constructor signature and the FK / Jacobian methods follow the reference
(kinematics/serial_manipulator.py:46-121, fk.py:39-86, jacobian.py:39-93); the arithmetic runs in
the HIP kernel `k_fk_jac_id` through the kernel registry ("kinematics.fk_jacobian").  IK solvers,
plotting and the other concerns of the reference class are out of scope.
"""
from __future__ import annotations

import logging
import os

from typing import List, Optional, Tuple

import numpy as np

from . import _hip
from .registry import execute_registered_kernel

__all__ = ["SerialManipulator"]


def _skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def _adjoint(T):
    """[[R, 0], [[p]R, R]] for twists ordered [w; v] (reference utils/se3.py:45-52)."""
    R, p = T[:3, :3], T[:3, 3]
    A = np.zeros((6, 6))
    A[:3, :3] = R
    A[3:, 3:] = R
    A[3:, :3] = _skew(p) @ R
    return A


def _screws_from_axes(omega_list, r_list) -> np.ndarray:
    """S_i = [w_i ; -w_i x r_i] (reference utils/screw.py extract_screw_list)."""
    w = np.asarray(omega_list, dtype=np.float64)
    r = np.asarray(r_list, dtype=np.float64)
    if w.shape[0] != 3:
        w = w.T
    if r.shape[0] != 3:
        r = r.T
    return np.vstack([w, np.cross(-w.T, r.T).T])


class SerialManipulator:
    """Kinematic model of a serial chain described by space screws and a home pose."""

    def __init__(self, M_list, omega_list, r_list=None, b_list=None, S_list=None, B_list=None, G_list=None,
                 joint_limits: Optional[List[Tuple[Optional[float], Optional[float]]]] = None) -> None:
        self.M_list = np.asarray(M_list, dtype=np.float64)
        self.G_list = G_list
        self.omega_list = omega_list
        if S_list is None:
            if r_list is None:
                raise ValueError("either S_list or (omega_list, r_list) is required")
            S_list = _screws_from_axes(omega_list, r_list)
        self.S_list = np.asarray(S_list, dtype=np.float64)
        self.r_list = r_list
        self.b_list = b_list
        M = self.M_list[-1] if self.M_list.ndim > 2 else self.M_list  # stacked poses: the last one (fk.py:69)
        self._M_ee = np.ascontiguousarray(M, dtype=np.float64)
        self.B_list = (np.asarray(B_list, dtype=np.float64) if B_list is not None
                       else _adjoint(np.linalg.inv(self._M_ee)) @ self.S_list)  # urdf/core.py:755-758
        n = self.S_list.shape[1]
        # body-frame results are derived from the space-frame kernel, valid iff B = Ad(M^-1) S
        self._B_consistent = bool(np.allclose(self.B_list, _adjoint(np.linalg.inv(self._M_ee)) @ self.S_list, atol=1e-8))
        self.joint_limits = joint_limits if joint_limits is not None else [(None, None)] * n
        self._hip_model: Optional[_hip.HipModel] = None

    # ---- compiled model (kinematics only needs screws + home pose; unit inertias are placeholders)
    def _kin_model(self) -> _hip.HipModel:
        if self._hip_model is None:
            n = self.S_list.shape[1]
            Mcom = np.tile(self._M_ee, (n, 1, 1))
            G = np.tile(np.eye(6), (n, 1, 1))
            self._hip_model = _hip.HipModel(self.S_list, Mcom, G, self._M_ee)
        return self._hip_model

    def _space_fk_jac(self, thetalist, want_T=True, want_J=True):
        q = np.atleast_2d(np.asarray(thetalist, dtype=np.float64))
        T, J, _ = execute_registered_kernel("kinematics.fk_jacobian", self._kin_model(), q, want_T=want_T, want_J=want_J)
        return T, J

    # ---- product of exponentials on the host, for what the compiled chain does not cover: a TRUNCATED joint vector
    #      (reference kinematics/fk.py:59-70 / jacobian.py:62-73 loop over `thetalist`, so theta[:k] gives
    #      prod_{j<k} exp([S_j] theta_j) . M - the reference's own mass matrix calls it that way, mass_matrix.py:71-75) and
    #      body-frame results of a model whose B_list is not Ad(M^-1) S_list (fk.py:72-80, jacobian.py:74-91)
    def _fk_poe(self, theta: np.ndarray, frame: str) -> np.ndarray:
        from .utils import transform_from_twist

        screws = self.S_list if frame == "space" else self.B_list
        if len(theta) > screws.shape[1]:
            raise IndexError(f"index {screws.shape[1]} is out of bounds for axis 1 with size {screws.shape[1]}")
        T = np.eye(4)
        for i, th in enumerate(theta):
            T = T @ transform_from_twist(screws[:, i], th)
        return T @ self._M_ee if frame == "space" else self._M_ee @ T

    def _jacobian_poe(self, theta: np.ndarray, frame: str) -> np.ndarray:
        from .utils import adjoint_transform, transform_from_twist

        k = len(theta)
        if k > self.S_list.shape[1]:
            raise IndexError(f"index {self.S_list.shape[1]} is out of bounds for axis 1 with size {self.S_list.shape[1]}")
        if k == 0:
            if frame == "space":
                return np.zeros((6, 0))
            raise IndexError("list assignment index out of range")  # the reference seeds columns[n - 1] (jacobian.py:79)
        T = np.eye(4)
        cols = [None] * k
        if frame == "space":
            for i in range(k):
                cols[i] = adjoint_transform(T) @ self.S_list[:, i]
                T = T @ transform_from_twist(self.S_list[:, i], theta[i])
        else:
            cols[k - 1] = self.B_list[:, k - 1]
            for i in range(k - 2, -1, -1):
                T = T @ transform_from_twist(self.B_list[:, i + 1], -theta[i + 1])
                cols[i] = adjoint_transform(T) @ self.B_list[:, i]
        return np.stack(cols, axis=1)

    def _needs_poe(self, thetalist, frame: str) -> bool:
        return np.shape(thetalist)[-1] != self.S_list.shape[1] or (frame == "body" and not self._B_consistent)

    def forward_kinematics(self, thetalist, frame: str = "space") -> np.ndarray:
        """End-effector pose (4, 4).  A 2-D `thetalist` (rows, n) returns (rows, 4, 4).  Fewer than n joint values give the
        pose of the chain truncated after them, as in the reference (kinematics/fk.py:59-70)."""
        if frame not in ("space", "body"):
            raise ValueError("Invalid frame specified. Choose 'space' or 'body'.")
        if self._needs_poe(thetalist, frame):
            q = np.asarray(thetalist, dtype=np.float64)
            return self._fk_poe(q, frame) if q.ndim == 1 else np.stack([self._fk_poe(r, frame) for r in q])
        # M . prod exp([B_i] q_i) == prod exp([S_i] q_i) . M whenever B = Ad(M^-1) S
        T, _ = self._space_fk_jac(thetalist, want_T=True, want_J=False)
        return T if np.ndim(thetalist) == 2 else T[0]

    def jacobian(self, thetalist, frame: str = "space") -> np.ndarray:
        """(6, n) Jacobian, or (rows, 6, n) for a 2-D `thetalist`; (6, k) for k < n joint values (jacobian.py:62-73)."""
        if frame not in ("space", "body"):
            raise ValueError("Invalid frame specified. Choose 'space' or 'body'.")
        if self._needs_poe(thetalist, frame):
            q = np.asarray(thetalist, dtype=np.float64)
            return self._jacobian_poe(q, frame) if q.ndim == 1 else np.stack([self._jacobian_poe(r, frame) for r in q])
        T, J = self._space_fk_jac(thetalist, want_T=(frame == "body"), want_J=True)
        if frame == "body":  # J_b = Ad(T_sb^-1) J_s = [[R^T, 0], [-R^T [p], R^T]] J_s, all rows at once
            Rt = np.swapaxes(T[:, :3, :3], 1, 2)
            Jw, Jv = J[:, :3, :], J[:, 3:, :]
            pxJw = np.cross(T[:, :3, 3][:, :, None], Jw, axis=1)       # [p] Jw, column by column
            J = np.concatenate([Rt @ Jw, Rt @ (Jv - pxJw)], axis=1)
        return J if np.ndim(thetalist) == 2 else J[0]

    # ---- inverse kinematics (reference kinematics/ik.py:39-311)
    def batch_inverse_kinematics(self, T_desired_batch, thetalist0_batch, eomg: float = 1e-6, ev: float = 1e-6,
                                 max_iterations: int = 10000, damping: float = 2e-2, step_cap: float = 0.3,
                                 weight_orientation: float = 1.0, weight_position: float = 1.0, adaptive_tuning: bool = False,
                                 backtracking: bool = False, seed: int = 1234):
        """B pose targets (B,4,4) from B initial guesses (B,n), one kernel launch, one lane per target:
        (theta (B,n), success (B,) bool, iterations (B,) int).  Each problem runs the reference's damped-least-squares
        iteration (optionally with its adaptive damping and its five-scale line search); `self.joint_limits` is the
        projection box (None = open end)."""
        model = self._kin_model()
        if (np.shape(T_desired_batch)[0] >= 16384 and os.environ.get("MANIPULAPY_HIP_SPECIALIZE", "1") != "0"
                and model.n <= _hip.MP_MAX_DOF):
            from .registry import _hip_routing_enabled, get_context

            if _hip_routing_enabled():  # big batch: the ~2 s (first time; cached on disk) of baking this robot's constants in pays for itself
                try:
                    get_context().specialize(model)
                except _hip.HipError as exc:  # e.g. no hiprtc on this machine: the generic GPU kernel serves
                    logging.getLogger("ManipulaPy.kinematics").warning("kernel specialisation unavailable (%s); using the generic kernel", exc)
        theta, ok, it, _ = execute_registered_kernel(
            "kinematics.inverse", model, T_desired_batch, thetalist0_batch, joint_limits=self.joint_limits, eomg=eomg,
            ev=ev, max_iterations=max_iterations, damping=damping, step_cap=step_cap, weight_orientation=weight_orientation,
            weight_position=weight_position, adaptive_tuning=adaptive_tuning, backtracking=backtracking, seed=seed)
        return theta, ok, it

    def iterative_inverse_kinematics(self, T_desired, thetalist0, eomg: float = 1e-6, ev: float = 1e-6, max_iterations: int = 10000,
                                     plot_residuals: bool = False, damping: float = 2e-2, step_cap: float = 0.3,
                                     png_name: str = "ik_residuals.png", weight_orientation: float = 1.0,
                                     weight_position: float = 1.0, adaptive_tuning: bool = False, backtracking: bool = False):
        """(theta, success, iterations) for one target — the reference's signature; only the residual plot is missing."""
        if plot_residuals:
            raise NotImplementedError("plot_residuals is not part of the batched kernel")
        T = np.asarray(T_desired, dtype=np.float64)
        if T.shape != (4, 4):
            raise ValueError(f"T_desired must be (4, 4), got {T.shape}")
        th, ok, it = self.batch_inverse_kinematics(T[None], np.asarray(thetalist0, dtype=np.float64)[None], eomg, ev, max_iterations,
                                                   damping, step_cap, weight_orientation, weight_position, adaptive_tuning, backtracking)
        return th[0], bool(ok[0]), int(it[0])

    def trac_ik(self, T_desired, theta0=None, timeout: float = 0.2, eomg: float = 1e-4, ev: float = 1e-4, num_restarts: int = 5,
                use_parallel: bool = False):
        """TRAC-IK style multi-start solve: (theta, success, solve_time) (reference kinematics/ik.py:600-651).  All initial
        guesses are rows of one batched inverse-kinematics launch - see manipulapy_amd/trac_ik.py."""
        from .trac_ik import trac_ik_solve

        if theta0 is not None:
            theta0 = np.array(theta0, dtype=float)
        return trac_ik_solve(self, T_desired, theta0, timeout, eomg, ev, num_restarts, use_parallel)

    def smart_inverse_kinematics(self, T_desired, strategy: str = "workspace_heuristic", theta_current=None, T_current=None, cache=None,
                                 eomg: float = 1e-6, ev: float = 1e-6, max_iterations: int = 10000, plot_residuals: bool = False,
                                 damping: float = 2e-2, step_cap: float = 0.3, png_name: str = "ik_residuals.png",
                                 weight_orientation: float = 1.0, weight_position: float = 1.0, adaptive_tuning: bool = True,
                                 backtracking: bool = True, auto_fallback: bool = True):
        """IK from a strategy-chosen initial guess, with up to four fallback starts (midpoint, three random) when the first
        attempt fails (reference kinematics/ik.py:327-475).  Returns (theta, success, iterations summed over the attempts)."""
        from . import ik_helpers

        valid = ["workspace_heuristic", "extrapolate", "cached", "random", "midpoint"]
        if strategy not in valid:
            raise ValueError(f"Unknown strategy '{strategy}'. Choose from: {valid}")
        T = np.asarray(T_desired, dtype=np.float64)
        n = len(self.joint_limits)

        def guess(name):
            if name == "workspace_heuristic":
                return ik_helpers.workspace_heuristic_guess(T, n, self.joint_limits)
            if name == "extrapolate":
                if theta_current is None or T_current is None:
                    return None
                return ik_helpers.extrapolate_from_current(theta_current, T_current, T, lambda th: self.jacobian(th, frame="space"),
                                                           self.joint_limits, alpha=0.5)
            if name == "cached":
                return None if cache is None else cache.get_nearest(T, k=3, joint_limits=self.joint_limits)
            if name == "random":
                return ik_helpers.random_in_limits(self.joint_limits)
            return ik_helpers.midpoint_of_limits(self.joint_limits)

        def attempt(theta0):
            return self.iterative_inverse_kinematics(T, theta0, eomg, ev, max_iterations, plot_residuals, damping, step_cap, png_name,
                                                     weight_orientation, weight_position, adaptive_tuning, backtracking)

        def pose_error(theta):
            Tc = np.asarray(self.forward_kinematics(theta))
            tr = np.trace(Tc[:3, :3].T @ T[:3, :3])
            return float(np.linalg.norm(Tc[:3, 3] - T[:3, 3]) + np.arccos(np.clip((tr - 1) / 2, -1, 1)))

        theta0 = guess(strategy)
        if theta0 is None:
            theta0 = guess("workspace_heuristic")
        theta, ok, iters = attempt(theta0)
        if ok or not auto_fallback:
            return theta, ok, iters
        total, best_theta, best_err = iters, theta, pose_error(theta)
        for name in ("midpoint", "random", "random", "random"):
            th, ok, it = attempt(guess(name))
            total += it
            if ok:
                return th, True, total
            err = pose_error(th)
            if err < best_err:
                best_err, best_theta = err, th
        return best_theta, False, total

    # the (initial guess, damping, step cap) ladder of the reference's robust solver (kinematics/ik.py:505-516)
    _ROBUST_STRATEGIES = (("workspace_heuristic", 0.02, 0.3), ("midpoint", 0.02, 0.3), ("workspace_heuristic", 0.01, 0.4),
                          ("random", 0.02, 0.3), ("random", 0.03, 0.25), ("midpoint", 0.01, 0.4), ("random", 0.015, 0.35),
                          ("random", 0.025, 0.3), ("workspace_heuristic", 0.03, 0.25), ("random", 0.02, 0.35))

    def batch_robust_inverse_kinematics(self, T_desired_batch, max_attempts: int = 10, eomg: float = 2e-3, ev: float = 2e-3,
                                        max_iterations: int = 5000):
        """Multi-start IK for B targets (reference kinematics/ik.py:477-598, one target per call there): up to ten
        attempts per target, each with its own initial guess, damping and step cap, adaptive tuning and backtracking on.
        The reference runs the attempts one after the other and stops at the first success; here all attempts of all
        targets are rows of a few launches (one per distinct damping / step-cap pair) and the winner is picked afterwards
        in the same order, so the answer is the sequential one.  Random guesses are drawn from NumPy's global stream in
        attempt order, target by target.  Returns (theta (B,n), success (B,), total_iterations (B,), strategy list)."""
        from . import ik_helpers

        T = np.asarray(T_desired_batch, dtype=np.float64)
        if T.ndim != 3 or T.shape[1:] != (4, 4):
            raise ValueError(f"T_desired_batch must be (B, 4, 4), got {T.shape}")
        B, n = T.shape[0], self.S_list.shape[1]
        plan = self._ROBUST_STRATEGIES[:max(0, min(int(max_attempts), len(self._ROBUST_STRATEGIES)))]
        A = len(plan)
        guesses = np.zeros((B, A, n))
        heur, mid = ik_helpers.workspace_heuristic_guess(T, n, self.joint_limits), ik_helpers.midpoint_of_limits(self.joint_limits)
        for b in range(B):
            for a, (name, _, _) in enumerate(plan):
                guesses[b, a] = heur[b] if name == "workspace_heuristic" else mid if name == "midpoint" else \
                    ik_helpers.random_in_limits(self.joint_limits)
        theta_all, ok_all, it_all = np.zeros((B, A, n)), np.zeros((B, A), dtype=bool), np.zeros((B, A), dtype=np.int64)
        for damping, cap in sorted({(d, c) for _, d, c in plan}):
            cols = [a for a, (_, d, c) in enumerate(plan) if (d, c) == (damping, cap)]
            th, ok, it = self.batch_inverse_kinematics(np.repeat(T, len(cols), axis=0), guesses[:, cols].reshape(-1, n), eomg, ev,
                                                       max_iterations, damping, cap, adaptive_tuning=True, backtracking=True)
            theta_all[:, cols], ok_all[:, cols], it_all[:, cols] = th.reshape(B, len(cols), n), ok.reshape(B, -1), it.reshape(B, -1)
        theta, success, total, names = np.zeros((B, n)), np.zeros(B, dtype=bool), np.zeros(B, dtype=np.int64), []
        err = None
        for b in range(B):
            wins = np.flatnonzero(ok_all[b])
            if wins.size:
                a = int(wins[0])
                theta[b], success[b], total[b] = theta_all[b, a], True, it_all[b, :a + 1].sum()
                names.append(plan[a][0])
                continue
            if A == 0:
                theta[b] = mid
                names.append("none")
                continue
            if err is None:  # pose error of every failed attempt: one batched FK (reference :567-585)
                Tc = np.asarray(self.forward_kinematics(theta_all.reshape(-1, n))).reshape(B, A, 4, 4)
                pos = np.linalg.norm(Tc[..., :3, 3] - T[:, None, :3, 3], axis=-1)
                tr = np.einsum("baij,bij->ba", Tc[..., :3, :3], T[:, :3, :3])  # trace(Rc^T Rd)
                err = pos + np.arccos(np.clip((tr - 1) / 2, -1, 1))
            a = int(np.argmin(err[b]))  # first minimum == the reference's strict "<" update order
            theta[b], total[b] = theta_all[b, a], it_all[b].sum()
            names.append(plan[a][0])
        return theta, success, total, names

    def robust_inverse_kinematics(self, T_desired, max_attempts: int = 10, eomg: float = 2e-3, ev: float = 2e-3,
                                  max_iterations: int = 5000, verbose: bool = False):
        """(theta, success, total_iterations, winning_strategy) — the reference's signature for one target."""
        th, ok, it, names = self.batch_robust_inverse_kinematics(np.asarray(T_desired, dtype=np.float64)[None], max_attempts, eomg, ev,
                                                                 max_iterations)
        if verbose:
            print(f"robust IK: {'success' if ok[0] else 'failed'} with '{names[0]}' after {int(it[0])} iterations")
        return th[0], bool(ok[0]), int(it[0]), names[0]

    def end_effector_pose(self, thetalist) -> np.ndarray:
        """[x, y, z, roll, pitch, yaw]: position and ZYX Euler angles of the end effector (reference kinematics/fk.py:88-104,
        utils/so3.py:240-251); (rows, 6) for a 2-D `thetalist`."""
        T = np.asarray(self.forward_kinematics(np.atleast_2d(np.asarray(thetalist, dtype=np.float64))))
        R = T[:, :3, :3]
        sy = np.sqrt(R[:, 0, 0] ** 2 + R[:, 1, 0] ** 2)
        pitch = np.arctan2(-R[:, 2, 0], sy)
        regular = np.stack((np.arctan2(R[:, 2, 1], R[:, 2, 2]), pitch, np.arctan2(R[:, 1, 0], R[:, 0, 0])), axis=1)
        singular = np.stack((np.arctan2(-R[:, 1, 2], R[:, 1, 1]), pitch, sy * 0), axis=1)
        out = np.concatenate((T[:, :3, 3], np.where((sy < 1e-6)[:, None], singular, regular)), axis=1)
        return out if np.ndim(thetalist) == 2 else out[0]

    def joint_velocity(self, thetalist, V_ee, frame: str = "space") -> np.ndarray:
        """pinv(J) V_ee (reference kinematics/velocity.py:65-89); 2-D inputs batch."""
        if frame not in ("space", "body"):
            raise ValueError("Invalid frame specified. Choose 'space' or 'body'.")
        J = np.asarray(self.jacobian(thetalist, frame=frame))
        V = np.asarray(V_ee, dtype=np.float64)
        return np.einsum("...ij,...j->...i", np.linalg.pinv(J), V)

    def update_state(self, joint_positions, joint_velocities=None) -> None:
        """reference kinematics/serial_manipulator.py:125-145."""
        self.joint_positions = np.asarray(joint_positions)
        self.joint_velocities = (np.asarray(joint_velocities) if joint_velocities is not None
                                 else np.zeros(self.joint_positions.shape, dtype=self.joint_positions.dtype))

    def end_effector_velocity(self, thetalist, dthetalist, frame: str = "space") -> np.ndarray:
        """reference kinematics/velocity.py:35-60."""
        return self.jacobian(thetalist, frame=frame) @ np.asarray(dthetalist, dtype=np.float64)
