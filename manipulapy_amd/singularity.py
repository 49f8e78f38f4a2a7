"""Singularity / workspace analysis on top of the batched FK + Jacobian kernel ("next" row f-4: the batched-FK consumer).

Reference: ManipulaPy/singularity/singularity_analysis.py (`Singularity`): `singularity_analysis` (:52-73, smallest
singular value of the space Jacobian < 1e-4), `condition_number` (:246-292, sigma_max / sigma_min, NaN -> inf),
`near_singularity_detection` (:294-311, condition number > threshold) and the Monte-Carlo workspace estimate
(:165-245: uniform joint samples -> forward kinematics of every sample -> convex hull).  The reference evaluates one
configuration per call and, for the workspace, runs its FK in a Python loop over the samples; here every method
also takes a (rows, n) batch, and all Jacobians / poses of a call come from one `kinematics.fk_jacobian` launch.
The small dense SVDs stay on the host (NumPy batched LAPACK); plotting is out of scope.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np

__all__ = ["Singularity"]


class Singularity:
    def __init__(self, serial_manipulator) -> None:
        self.serial_manipulator = serial_manipulator

    # singular values of the space Jacobian, (n_sv,) or (rows, n_sv), descending
    def _singular_values(self, thetalist) -> np.ndarray:
        J = np.asarray(self.serial_manipulator.jacobian(thetalist, frame="space"), dtype=np.float64)
        with np.errstate(all="ignore"):
            return np.linalg.svd(J, compute_uv=False)

    def singularity_analysis(self, thetalist):
        """True where the smallest singular value of J_s is below 1e-4 (bool, or (rows,) bool for a batch)."""
        s = self._singular_values(thetalist)
        flag = s[..., -1] < 1e-4
        return bool(flag) if flag.ndim == 0 else flag

    def condition_number(self, thetalist):
        """sigma_max / sigma_min of J_s; a 0/0 or non-finite spectrum is reported as inf, like np.linalg.cond."""
        s = self._singular_values(thetalist)
        with np.errstate(all="ignore"):
            ratio = s[..., 0] / s[..., -1]
        ratio = np.where(np.isnan(ratio), np.inf, ratio)
        return ratio[()] if ratio.ndim == 0 else ratio

    def near_singularity_detection(self, thetalist, threshold: float = 1e-2):
        c = self.condition_number(thetalist)
        flag = np.asarray(c) > threshold
        return flag[()] if flag.ndim == 0 else flag   # numpy.bool_ for one configuration, as the reference returns (singularity_analysis.py:304)

    def manipulability(self, thetalist):
        """Yoshikawa measure sqrt(det(J J^T)) = product of the singular values (batched helper, not in the reference)."""
        s = self._singular_values(thetalist)
        m = np.prod(s, axis=-1)
        return m[()] if m.ndim == 0 else m

    def workspace_monte_carlo(self, joint_limits: Sequence[Tuple[float, float]], num_samples: int = 10000,
                              seed: Optional[int] = 1234, hull: bool = True) -> Dict[str, np.ndarray]:
        """Uniform joint samples inside `joint_limits` (float32, as the reference draws them), the end-effector position
        of every sample from ONE batched FK launch, and optionally the convex hull of the cloud.

        Returns {"joint_samples": (R, n) float32, "points": (R, 3) float64, "simplices": (F, 3) int, "volume": float};
        the random stream is NumPy's (the reference's xoroshiro128+ device stream is not a parity target: its own
        output is only ever plotted)."""
        lim = np.asarray(joint_limits, dtype=np.float32)
        if lim.ndim != 2 or lim.shape[1] != 2:
            raise ValueError("joint_limits must be a sequence of (low, high) pairs")
        if num_samples < 1:
            raise ValueError("num_samples must be positive")
        rng = np.random.default_rng(seed)
        u = rng.random((int(num_samples), lim.shape[0]), dtype=np.float32)
        samples = u * (lim[:, 1] - lim[:, 0]) + lim[:, 0]
        T = np.asarray(self.serial_manipulator.forward_kinematics(samples.astype(np.float64)))
        out = {"joint_samples": samples, "points": np.ascontiguousarray(T[:, :3, 3])}
        if hull:
            from scipy.spatial import ConvexHull

            h = ConvexHull(out["points"])
            out["simplices"], out["volume"] = h.simplices, float(h.volume)
        return out
