"""Computed-torque / feedforward control on top of the batched dynamics kernels ("next" row f-3).

Reference: ManipulaPy/control/computed_torque.py:17-124 (`computed_torque_control`,
`feedforward_control`) inside `ManipulatorController` (control/manipulator_controller.py).  The
reference evaluates  tau = M(q) (Kp e + Ki int(e) + Kd de) + inverse_dynamics(q, qd, qdd_d, g, 0)  with a
mass matrix and an inverse-dynamics call per sample.  Inverse dynamics is affine in the acceleration
(tau = M qdd + bias), so the same torque is ONE inverse-dynamics evaluation with the commanded acceleration

        qdd_cmd = qdd_d + Kp e + Ki int(e) + Kd de,

which is what runs here — for a single sample or for a whole batch of (rows, n) samples in one launch.
The reference's other model-based laws (PD / PID, PD + feed-forward, joint- and Cartesian-space PD, robust and adaptive
control, the Kalman filter on [q; qd]) are the same inverse- / forward-dynamics / FK / Jacobian calls plus host arithmetic and
are mirrored below, with the step-response metrics, the Ziegler-Nichols gain formulas and the closed-loop gain sweep
(`find_ultimate_gain_and_period`: every gain of the ladder is one lane of a single "control.pd_regulation" launch); the
response plot stays out of scope.
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

    # ---- the other model-based laws of the reference: each is inverse dynamics plus host arithmetic, so a (rows, n)
    #      batch of samples is one launch (control/pid.py, control/computed_torque.py, control/robust_adaptive.py)
    @staticmethod
    def pd_control(desired_position, desired_velocity, current_position, current_velocity, Kp, Kd) -> np.ndarray:
        """Kp e + Kd de (reference control/pid.py:17-52)."""
        e = np.asarray(desired_position, dtype=np.float64) - np.asarray(current_position, dtype=np.float64)
        de = np.asarray(desired_velocity, dtype=np.float64) - np.asarray(current_velocity, dtype=np.float64)
        return np.asarray(Kp) * e + np.asarray(Kd) * de

    def pid_control(self, thetalistd, dthetalistd, thetalist, dthetalist, dt, Kp, Ki, Kd, i_clamp: Optional[float] = None) -> np.ndarray:
        """Kp e + Ki int(e) + Kd de with the controller's integral state (reference control/pid.py:54-118)."""
        q = np.asarray(thetalist, dtype=np.float64)
        if i_clamp is not None and (not np.isfinite(i_clamp) or i_clamp <= 0):
            raise ValueError("i_clamp must be a positive finite number")
        if self.eint is None or self.eint.shape != q.shape:
            self.eint = np.zeros(q.shape)
        e = np.asarray(thetalistd, dtype=np.float64) - q
        self.eint = self.eint + e * dt
        if i_clamp is not None:
            self.eint = np.clip(self.eint, -i_clamp, i_clamp)
        de = np.asarray(dthetalistd, dtype=np.float64) - np.asarray(dthetalist, dtype=np.float64)
        return np.asarray(Kp) * e + np.asarray(Ki) * self.eint + np.asarray(Kd) * de

    def pd_feedforward_control(self, desired_position, desired_velocity, desired_acceleration, current_position, current_velocity,
                               Kp, Kd, g, Ftip) -> np.ndarray:
        """PD on the error + inverse dynamics at the desired state (reference control/computed_torque.py:134-178)."""
        return (self.pd_control(desired_position, desired_velocity, current_position, current_velocity, Kp, Kd)
                + self.feedforward_control(desired_position, desired_velocity, desired_acceleration, g, Ftip))

    @staticmethod
    def enforce_limits(thetalist, dthetalist, tau, joint_limits, torque_limits):
        """Clip angles and torques to their limits; velocities pass through (reference control/computed_torque.py:181-211)."""
        jl, tl = np.asarray(joint_limits, dtype=np.float64), np.asarray(torque_limits, dtype=np.float64)
        return (np.clip(np.asarray(thetalist, dtype=np.float64), jl[:, 0], jl[:, 1]), np.asarray(dthetalist, dtype=np.float64),
                np.clip(np.asarray(tau, dtype=np.float64), tl[:, 0], tl[:, 1]))

    @staticmethod
    def joint_space_control(desired_joint_angles, current_joint_angles, current_joint_velocities, Kp, Kd) -> np.ndarray:
        """Kp (qd - q) - Kd qdot (reference control/computed_torque.py:214-245)."""
        e = np.asarray(desired_joint_angles, dtype=np.float64) - np.asarray(current_joint_angles, dtype=np.float64)
        return np.asarray(Kp) * e - np.asarray(Kd) * np.asarray(current_joint_velocities, dtype=np.float64)

    def cartesian_space_control(self, desired_position, current_joint_angles, current_joint_velocities, Kp, Kd) -> np.ndarray:
        """J_v^T (Kp (x_d - x) - Kd J_v qdot), J_v = the linear rows of the space Jacobian as the reference slices them
        (control/computed_torque.py:248-296: rows 0..2).  2-D joint states batch: FK + Jacobian of all rows in one launch."""
        q = np.asarray(current_joint_angles, dtype=np.float64)
        single = q.ndim == 1
        q2, qd2 = np.atleast_2d(q), np.atleast_2d(np.asarray(current_joint_velocities, dtype=np.float64))
        xd = np.atleast_2d(np.asarray(desired_position, dtype=np.float64))
        T = np.asarray(self.dynamics.forward_kinematics(q2))
        Jv = np.asarray(self.dynamics.jacobian(q2))[:, :3, :]
        e = xd - T[:, :3, 3]
        xdot = np.einsum("rij,rj->ri", Jv, qd2)
        Kp, Kd = np.asarray(Kp, dtype=np.float64), np.asarray(Kd, dtype=np.float64)
        kp = e @ Kp.T if Kp.ndim == 2 else Kp * e
        kd = xdot @ Kd.T if Kd.ndim == 2 else Kd * xdot
        tau = np.einsum("rij,ri->rj", Jv, kp - kd)
        return tau[0] if single else tau

    def robust_control(self, thetalist, dthetalist, ddthetalist, g, Ftip, disturbance_estimate, adaptation_gain) -> np.ndarray:
        """M qdd + c + g + J^T Ftip + gain * disturbance estimate (reference control/robust_adaptive.py:17-66): the first four
        terms ARE inverse dynamics, so this is one launch plus an addition."""
        q = np.asarray(thetalist, dtype=np.float64)
        return (self._id(q, np.asarray(dthetalist, dtype=np.float64), np.asarray(ddthetalist, dtype=np.float64), g, Ftip)
                + np.asarray(adaptation_gain, dtype=np.float64) * np.asarray(disturbance_estimate, dtype=np.float64))

    def adaptive_control(self, thetalist, dthetalist, ddthetalist, g, Ftip, measurement_error, adaptation_gain) -> np.ndarray:
        """Inverse dynamics + the running parameter estimate, updated by gain * measurement error on every call
        (reference control/robust_adaptive.py:69-131; single sample, as there: the estimate is controller state)."""
        q = np.asarray(thetalist, dtype=np.float64)
        if q.ndim != 1:
            raise ValueError("adaptive_control keeps one parameter estimate: call it per sample")
        if getattr(self, "parameter_estimate", None) is None:
            self.parameter_estimate = np.zeros(q.shape[0])
        gamma = float(np.asarray(adaptation_gain, dtype=np.float64).reshape(-1)[0])
        self.parameter_estimate = self.parameter_estimate + gamma * np.asarray(measurement_error, dtype=np.float64).reshape(-1)
        return (self._id(q, np.asarray(dthetalist, dtype=np.float64), np.asarray(ddthetalist, dtype=np.float64), g, Ftip)
                + self.parameter_estimate)

    # ---- Kalman filter on the joint state [q; qd] (reference control/kalman.py): the prediction integrates forward dynamics
    def kalman_filter_predict(self, thetalist, dthetalist, taulist, g, Ftip, dt, Q) -> None:
        q, qd = np.asarray(thetalist, dtype=np.float64), np.asarray(dthetalist, dtype=np.float64)
        Q = np.asarray(Q, dtype=np.float64)
        x = getattr(self, "x_hat", None)
        n2 = x.shape[0] if x is not None else 2 * len(q)
        if Q.shape != (n2, n2):
            raise ValueError(f"Q must have shape ({n2}, {n2}), got {Q.shape}")
        if x is None:
            x = np.concatenate((q, qd))
        n = len(q)
        qdd = np.asarray(self.dynamics.forward_dynamics(x[:n], x[n:], np.asarray(taulist, dtype=np.float64), g, Ftip))
        x_pred = np.concatenate((x[:n] + x[n:] * dt, qdd * dt + x[n:]))
        P = getattr(self, "P", None)
        if P is None:
            P = np.eye(len(x_pred))
        self.P = P + Q  # F = I
        self.x_hat = x_pred

    def kalman_filter_update(self, z, R) -> None:
        x = getattr(self, "x_hat", None)
        if x is None:
            raise ValueError("kalman_filter_update called before kalman_filter_predict; x_hat has not been initialized")
        n = x.shape[0]
        P = getattr(self, "P", None)
        if P is None or getattr(P, "shape", None) != (n, n):
            raise ValueError(f"P must be initialized with shape ({n}, {n}) before update; got {None if P is None else P.shape}")
        z, R = np.asarray(z, dtype=np.float64), np.asarray(R, dtype=np.float64)
        if z.shape != (n,):
            raise ValueError(f"z must have shape ({n},) to match x_hat, got {z.shape}")
        if R.shape != (n, n):
            raise ValueError(f"R must have shape ({n}, {n}), got {R.shape}")
        K = P @ np.linalg.inv(P + R)  # H = I
        self.x_hat = x + K @ (z - x)
        self.P = (np.eye(n) - K) @ P

    def kalman_filter_control(self, thetalistd, dthetalistd, thetalist, dthetalist, taulist, g, Ftip, dt, Q, R):
        """One predict + update cycle on the measured state; returns the filtered (q, qd) (reference control/kalman.py:131-170)."""
        q, qd = np.asarray(thetalist, dtype=np.float64), np.asarray(dthetalist, dtype=np.float64)
        self.kalman_filter_predict(q, qd, taulist, g, Ftip, dt, Q)
        self.kalman_filter_update(np.concatenate((q, qd)), R)
        return self.x_hat[:len(q)], self.x_hat[len(q):]

    # ---- step-response metrics and Ziegler-Nichols gains (reference control/metrics.py; plotting and the gain sweep stay out)
    @staticmethod
    def calculate_rise_time(time, response, set_point: float) -> float:
        """Time between the first samples at or above 10 % and 90 % of the set point; inf if either is never reached."""
        time, response = np.asarray(time), np.asarray(response)
        lo, hi = response >= 0.1 * set_point, response >= 0.9 * set_point
        if not lo.any() or not hi.any():
            return float("inf")
        return float(time[int(np.argmax(hi))] - time[int(np.argmax(lo))])

    @staticmethod
    def calculate_percent_overshoot(response, set_point: float) -> float:
        if set_point == 0:
            return 0.0
        return float((np.amax(np.asarray(response)) - set_point) / set_point * 100)

    @staticmethod
    def calculate_settling_time(time, response, set_point: float, tolerance: float = 0.02) -> float:
        """First time after which the response stays inside |set_point| * tolerance of the set point; inf if it never does."""
        time, response = np.asarray(time), np.asarray(response)
        inside = np.abs(response - set_point) <= abs(set_point) * tolerance
        if not inside.any():
            return float("inf")
        outside = np.flatnonzero(~inside)
        last = int(outside[-1]) if outside.size else -1
        if last == len(inside) - 1:
            return float("inf")
        return float(time[last + 1])

    @staticmethod
    def calculate_steady_state_error(response, set_point: float) -> float:
        return float(np.asarray(response)[-1] - set_point)

    @staticmethod
    def ziegler_nichols_tuning(Ku, Tu, kind: str = "PID"):
        """(Kp, Ki, Kd) from the ultimate gain / period: P 0.5 Ku; PI 0.45 Ku, 1.2 Ku / Tu; PID 0.6 Ku, 2 Kp / Tu, Kp Tu / 8."""
        Ku = np.asarray(Ku, dtype=float)
        kind = kind.upper()
        if kind == "P":
            Kp, Ki, Kd = 0.50 * Ku, 0.0 * Ku, 0.0 * Ku
        else:
            Tu = np.asarray(Tu, dtype=float)
            if not np.all(np.isfinite(Tu)) or np.any(Tu <= 0):
                raise ValueError(f"Tu (ultimate period) must be positive and finite, got Tu={Tu!r}.")
            if kind == "PI":
                Kp, Ki, Kd = 0.45 * Ku, 1.2 * Ku / Tu, 0.0 * Ku
            elif kind == "PID":
                Kp = 0.60 * Ku
                Ki, Kd = 2.0 * Kp / Tu, 0.125 * Kp * Tu
            else:
                raise ValueError("kind must be 'P', 'PI' or 'PID'")
        if Ku.size == 1:
            return float(Kp), float(Ki), float(Kd)
        return Kp, Ki, Kd

    def tune_controller(self, Ku, Tu, kind: str = "PID"):
        return self.ziegler_nichols_tuning(Ku, Tu, kind)

    def find_ultimate_gain_and_period(self, thetalist, desired_joint_angles, dt, max_steps: int = 1000):
        """(ultimate_gain, ultimate_period, gain_history, error_history) by the reference's sweep (control/metrics.py:280-366): for
        Kp = 0.01, 0.011, ... (x 1.1, below 1000) simulate `max_steps` steps of the arm under tau = Kp (desired - theta) from rest
        (semi-implicit Euler on M^-1 (tau - c - g), g = [0, 0, -9.81]) and stop at the first gain whose error norm rose over the
        last step by less than 20 %.  The reference runs the gains one after the other; the runs are independent, so here the
        whole ladder (121 gains) is ONE registered "control.pd_regulation" launch, one lane per gain, and the sequential stopping
        rule is applied to its output - the same answer.  The integral state is reset like the reference's."""
        from .registry import execute_registered_kernel

        theta = np.asarray(thetalist, dtype=np.float64).copy()
        desired = np.asarray(desired_joint_angles, dtype=np.float64)
        steps = int(max_steps)
        gains, Kp = [], 0.01
        while Kp < 1000:        # the ladder the reference's `while not oscillation and Kp < 1000` can visit
            gains.append(Kp)
            Kp *= 1.1
        K = len(gains)
        self.eint = np.zeros_like(theta)
        if getattr(self.dynamics, "Mlist_per_link", None) is None:
            raise NotImplementedError("find_ultimate_gain_and_period needs a compiled model (ManipulatorDynamics with Mlist_per_link)")
        errors, count = execute_registered_kernel("control.pd_regulation", self.dynamics._model_for(K), np.tile(theta, (K, 1)),
                                                  np.tile(desired, (K, 1)), np.asarray(gains), np.zeros(K), np.array([0.0, 0.0, -9.81]),
                                                  float(dt), steps)
        gain_history, error_history, ultimate = [], [], gains[-1] * 1.1   # (ladder exhausted: the reference leaves with the next Kp)
        for k in range(K):
            e = errors[k, : count[k]].copy()
            gain_history.append(gains[k])
            error_history.append(e)
            if len(e) >= 2 and e[-2] < e[-1] < e[-2] * 1.2:
                ultimate = gains[k]
                break
        last = error_history[-1] if error_history else np.zeros(0)
        crossings = int(np.count_nonzero(np.diff(np.sign(last)))) // 2 if len(last) > 1 else 0
        return float(ultimate), float((steps * dt) / max(1, crossings)), gain_history, error_history

    def _id(self, q, qd, qdd, g, Ftip):
        single = q.ndim == 1
        tau = self.dynamics._id(np.atleast_2d(q), np.atleast_2d(qd), np.atleast_2d(qdd), g, Ftip)
        return tau[0] if single else tau
