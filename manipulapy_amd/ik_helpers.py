"""Initial guesses for inverse kinematics, for one target or a batch (reference kinematics/ik_helpers.py).

`workspace_heuristic_guess` (:28-114), `random_in_limits` (:179-212), `midpoint_of_limits` (:215-246) with the same
formulas and the same use of NumPy's global random stream (one `np.random.uniform` per joint, in joint order), so that a
caller who seeds `np.random` gets the guesses the reference would draw.  The batched forms exist because every attempt
of a multi-start solve is just another row of one inverse-kinematics launch.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

Limits = Sequence[Tuple[Optional[float], Optional[float]]]

__all__ = ["workspace_heuristic_guess", "random_in_limits", "midpoint_of_limits", "clip_to_limits", "extrapolate_from_current",
           "IKInitialGuessCache", "se3_log_vector", "adaptive_multi_start_ik"]


def clip_to_limits(theta: np.ndarray, joint_limits: Limits) -> np.ndarray:
    """Project onto the joint-limit box; None leaves an end open (reference ik_helpers.py:407-446)."""
    theta = np.asarray(theta, dtype=np.float64)
    n = theta.shape[-1]
    lo = np.array([-np.inf if (i >= len(joint_limits) or joint_limits[i][0] is None) else joint_limits[i][0] for i in range(n)])
    hi = np.array([np.inf if (i >= len(joint_limits) or joint_limits[i][1] is None) else joint_limits[i][1] for i in range(n)])
    return np.minimum(np.maximum(theta, lo), hi)


def workspace_heuristic_guess(T_desired, n_joints: int, joint_limits: Limits) -> np.ndarray:
    """Geometric guess: joint 1 from the target's azimuth, joint 2 from its elevation, joint 3 = pi / 4, joints 4-6 from the
    ZYZ-like angles of the target rotation.  T_desired (4,4) -> (n,), or (B,4,4) -> (B,n)."""
    T = np.asarray(T_desired, dtype=np.float64)
    single = T.ndim == 2
    T = T[None] if single else T
    B = T.shape[0]
    th = np.zeros((B, n_joints))
    p, R = T[:, :3, 3], T[:, :3, :3]
    if n_joints >= 1:
        th[:, 0] = np.arctan2(p[:, 1], p[:, 0])
    if n_joints >= 2:
        r_xy = np.sqrt(p[:, 0] ** 2 + p[:, 1] ** 2)
        th[:, 1] = np.where(r_xy > 1e-6, np.arctan2(p[:, 2], r_xy), 0.0)
    if n_joints >= 3:
        th[:, 2] = np.pi / 4
    if n_joints > 3:
        generic = np.abs(R[:, 2, 2]) < 0.9999
        th[:, 3] = np.where(generic, np.arctan2(R[:, 1, 2], R[:, 0, 2]), np.arctan2(R[:, 1, 0], R[:, 0, 0]))
        if n_joints >= 5:
            th[:, 4] = np.where(generic, np.arccos(np.clip(R[:, 2, 2], -1, 1)), 0.0)
        if n_joints >= 6:
            th[:, 5] = np.where(generic, np.arctan2(R[:, 2, 1], -R[:, 2, 0]), 0.0)
    th = clip_to_limits(th, joint_limits)
    return th[0] if single else th


def random_in_limits(joint_limits: Limits, count: Optional[int] = None) -> np.ndarray:
    """Uniform inside the limits (open ends: a pi-wide band next to the closed end, or [-pi, pi]); draws from NumPy's
    global stream joint by joint, row by row.  count=None -> (n,), else (count, n)."""
    rows = 1 if count is None else int(count)
    out = np.empty((rows, len(joint_limits)))
    for r in range(rows):
        for j, (mn, mx) in enumerate(joint_limits):
            if mn is not None and mx is not None:
                out[r, j] = np.random.uniform(mn, mx)
            elif mn is not None:
                out[r, j] = mn + np.random.uniform(0, np.pi)
            elif mx is not None:
                out[r, j] = mx - np.random.uniform(0, np.pi)
            else:
                out[r, j] = np.random.uniform(-np.pi, np.pi)
    return out[0] if count is None else out


def midpoint_of_limits(joint_limits: Limits) -> np.ndarray:
    return np.array([(mn + mx) / 2.0 if mn is not None and mx is not None else 0.0 for mn, mx in joint_limits], dtype=np.float64)


def se3_log_vector(T) -> np.ndarray:
    """[rotation vector; theta G^-1 p] of a homogeneous transform: what the reference obtains from
    se3ToVec(MatrixLog6(T)) (utils/se3.py:55-166) — `manipulapy_amd.utils.logm`."""
    from .utils import logm

    return logm(T)


def extrapolate_from_current(theta_current, T_current, T_desired, jacobian_func, joint_limits: Limits, alpha: float = 0.5) -> np.ndarray:
    """theta + alpha pinv(J(theta)) log6(T_desired T_current^-1), clipped (reference kinematics/ik_helpers.py:116-176)."""
    theta = np.asarray(theta_current, dtype=np.float64)
    V = se3_log_vector(np.asarray(T_desired, dtype=np.float64) @ np.linalg.inv(np.asarray(T_current, dtype=np.float64)))
    J = np.asarray(jacobian_func(theta), dtype=np.float64)
    return clip_to_limits(theta + alpha * (np.linalg.pinv(J) @ V), joint_limits)


class IKInitialGuessCache:
    """(pose, solution, residual) triples of solved IK problems; `get_nearest` proposes an initial guess for a new target
    from the k closest cached poses (reference kinematics/ik_helpers.py:249-405: FIFO eviction, distance = position error +
    0.1 Frobenius rotation error + 0.2 residual, the best entry itself when its residual is below 1e-3, else the mean)."""

    def __init__(self, max_size: int = 100) -> None:
        self.cache: List[tuple] = []
        self.max_size = max_size

    def add(self, T, theta, residual: Optional[float] = None) -> None:
        self.cache.append((np.array(T, dtype=np.float64), np.array(theta, dtype=np.float64), None if residual is None else float(residual)))
        if len(self.cache) > self.max_size:
            self.cache.pop(0)

    def get_nearest(self, T_desired, k: int = 3, joint_limits: Optional[Limits] = None) -> Optional[np.ndarray]:
        if not self.cache:
            return None
        Td = np.asarray(T_desired, dtype=np.float64)
        scored = sorted(((self._pose_distance(Td, T) + 0.2 * (0.0 if r is None else r), 0.0 if r is None else r, th)
                         for T, th, r in self.cache), key=lambda x: x[0])
        near = scored[:min(k, len(scored))]
        best_quality, best_theta = near[0][1], near[0][2].copy()
        avg = best_theta.copy() if best_quality < 1e-3 else np.mean([e[2] for e in near], axis=0)
        if joint_limits is not None:
            avg, best_theta = clip_to_limits(avg, joint_limits), clip_to_limits(best_theta, joint_limits)
        return best_theta if np.linalg.norm(avg - best_theta) < 1e-6 else avg

    def clear(self) -> None:
        self.cache.clear()

    def size(self) -> int:
        return len(self.cache)

    @staticmethod
    def _pose_distance(T1, T2) -> float:
        return float(np.linalg.norm(T1[:3, 3] - T2[:3, 3]) + 0.1 * np.linalg.norm(T1[:3, :3] - T2[:3, :3], "fro"))



# (strategy, damping, step cap) ladder of the reference's adaptive multi-start (kinematics/ik_helpers.py:505-520)
_ADAPTIVE_LADDER = (("workspace_heuristic", 0.02, 0.3), ("midpoint", 0.03, 0.3), ("random", 0.02, 0.3), ("random", 0.03, 0.25),
                    ("random", 0.015, 0.35), ("random", 0.01, 0.4), ("random", 0.04, 0.2), ("workspace_heuristic", 0.01, 0.4),
                    ("random", 0.05, 0.15), ("midpoint", 0.01, 0.5))


def adaptive_multi_start_ik(ik_solver_func, T_desired, max_attempts: int = 10, eomg: float = 2e-3, ev: float = 2e-3,
                            max_iterations: int = 1500, verbose: bool = False):
    """Up to ten attempts of `ik_solver_func` (a robot's smart_inverse_kinematics) with the reference's ladder of initial-guess
    strategies and damping / step-cap pairs, first success wins (reference kinematics/ik_helpers.py:449-577): returns
    (theta, success, total iterations, winning strategy or "none (failed)"); on failure theta is the LAST attempt's, as in the
    reference; an attempt that raises is skipped.  For many targets at once see SerialManipulator.batch_robust_inverse_kinematics."""
    last, total = None, 0
    for attempt, (strategy, damping, step_cap) in enumerate(_ADAPTIVE_LADDER[:max_attempts]):
        if verbose:
            print(f"Attempt {attempt + 1}/{max_attempts}: strategy={strategy}, damping={damping}, step_cap={step_cap}")
        try:
            theta, ok, iters = ik_solver_func(T_desired, strategy=strategy, eomg=eomg, ev=ev, max_iterations=max_iterations,
                                              damping=damping, step_cap=step_cap)
        except Exception as exc:   # the reference skips a failing attempt
            if verbose:
                print(f"  exception: {exc}")
            continue
        total += iters
        if ok:
            return theta, True, total, strategy
        last = theta
    if last is None:
        last = midpoint_of_limits([])
    return last, False, total, "none (failed)"
