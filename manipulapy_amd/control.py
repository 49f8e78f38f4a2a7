"""Computed-torque / feedforward control on top of the batched dynamics kernels ("next" row f-3).

Reference: ManipulaPy/control/computed_torque.py:17-124 (`computed_torque_control`,
`feedforward_control`) inside `ManipulatorController` (control/manipulator_controller.py).  The
reference evaluates  tau = M(q) (Kp e + Ki int(e) + Kd de) + inverse_dynamics(q, qd, qdd_d, g, 0)  with a
mass matrix and an inverse-dynamics call per sample.  Inverse dynamics is affine in the acceleration
(tau = M qdd + bias), so the same torque is ONE inverse-dynamics evaluation with the commanded acceleration

        qdd_cmd = qdd_d + Kp e + Ki int(e) + Kd de,

which is what runs here — for a single sample or for a whole batch of (rows, n) samples in one launch.
The PID / adaptive / Kalman controllers of the reference are single-sample host loops and stay out of scope.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

__all__ = ["ManipulatorController"]


class ManipulatorController:
    def __init__(self, manipulator_dynamics) -> None:
        self.dynamics = manipulator_dynamics
        self.eint: Optional[np.ndarray] = None

    def computed_torque_control(self, thetalistd, dthetalistd, ddthetalistd, thetalist, dthetalist, g, dt, Kp, Ki, Kd,
                                i_clamp: Optional[float] = None) -> np.ndarray:
        """Torque command (n,), or (rows, n) when the states are 2-D (independent samples, one launch).
        The integral state `self.eint` follows the reference: reset on shape change, e*dt accumulation,
        optional symmetric clamp."""
        qd_des = np.asarray(thetalistd, dtype=np.float64)
        q = np.asarray(thetalist, dtype=np.float64)
        if i_clamp is not None and (not np.isfinite(i_clamp) or i_clamp <= 0):
            raise ValueError("i_clamp must be a positive finite number")
        if self.eint is None or self.eint.shape != q.shape:
            self.eint = np.zeros(q.shape)
        e = qd_des - q
        self.eint = self.eint + e * dt
        if i_clamp is not None:
            self.eint = np.clip(self.eint, -i_clamp, i_clamp)
        de = np.asarray(dthetalistd, dtype=np.float64) - np.asarray(dthetalist, dtype=np.float64)
        qdd_cmd = (np.asarray(ddthetalistd, dtype=np.float64) + np.asarray(Kp) * e + np.asarray(Ki) * self.eint
                   + np.asarray(Kd) * de)
        return self._id(q, np.asarray(dthetalist, dtype=np.float64), qdd_cmd, g, None)

    def feedforward_control(self, desired_position, desired_velocity, desired_acceleration, g, Ftip) -> np.ndarray:
        """inverse_dynamics at the desired state (reference control/computed_torque.py:93-124); 2-D inputs batch."""
        return self._id(np.asarray(desired_position, dtype=np.float64), np.asarray(desired_velocity, dtype=np.float64),
                        np.asarray(desired_acceleration, dtype=np.float64), g, Ftip)

    def _id(self, q, qd, qdd, g, Ftip):
        single = q.ndim == 1
        tau = self.dynamics._id(np.atleast_2d(q), np.atleast_2d(qd), np.atleast_2d(qdd), g, Ftip)
        return tau[0] if single else tau
