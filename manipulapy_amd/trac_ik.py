"""TRAC-IK style multi-start inverse kinematics - host-side mirror of ManipulaPy/kinematics/trac_ik.py.

The reference (kinematics/trac_ik.py:110-280) tries a damped-least-squares solver on `num_restarts` initial guesses one
after the other under a wall-clock budget, then hands the best configuration to an SQP polish (scipy SLSQP,
:534-605), and returns `(theta, success, solve_time)`.  The same contract here (`TracIKSolver.solve`, `trac_ik_solve`,
`SerialManipulator.trac_ik`: parameter names, defaults and return types are the frozen ones of
tests/golden/api_contract_golden.json), re-cut for a batched solver:

* the initial guesses are the reference's (:282-312): the caller's `theta0` or the workspace heuristic, the midpoint
  of the limits, the zero configuration, the negated midpoint, then uniform draws inside the limits;
* built from a robot (`trac_ik_solve(robot, ...)`, `robot.trac_ik(...)`) ALL guesses are rows of ONE
  `batch_inverse_kinematics` launch - the registered "kinematics.inverse" kernel with the Levenberg-Marquardt style
  adaptive damping and the five-scale line search on (the reference's DLS settings, :346-360: damping 0.02, step cap
  0.3, at most 3000 iterations); the successful row with the smallest pose error wins.  `use_parallel` has nothing
  left to choose (the reference's thread pool exists to overlap the same attempts): it is accepted and ignored;
* built from bare `fk_func` / `jacobian_func` callables (the reference's constructor, :79-108) the guesses are tried one
  after the other with the host loop below, under the reference's time budgeting (:232-267);
* if nothing met the tolerances and time is left, SLSQP minimises the weighted squared pose error inside the joint
  limits from the best configuration found (reference :534-605), when scipy is importable.

`timeout` bounds the host phases; a launched batch runs to its iteration cap (a kernel is not preempted) - at ~2 us
per iteration and row that cap is a few milliseconds.  The random guesses come from NumPy's global stream, like the
reference's (`np.random.uniform` per joint).
"""
from __future__ import annotations

import time
from typing import Any, Callable, List, Optional, Tuple

import numpy as np

from . import ik_helpers

__all__ = ["TracIKSolver", "trac_ik_solve"]

_DLS_DAMPING, _DLS_STEP_CAP, _DLS_MAX_ITERS = 0.02, 0.3, 3000   # reference kinematics/trac_ik.py:346-360


def _pose_error(T_current, T_desired):
    """([omega_space; dp], rotation angle, translation norm) - reference kinematics/trac_ik.py:635-714."""
    Tc, Td = np.asarray(T_current, dtype=np.float64), np.asarray(T_desired, dtype=np.float64)
    dp = Td[:3, 3] - Tc[:3, 3]
    E = Tc[:3, :3].T @ Td[:3, :3]
    angle = float(np.arccos(np.clip(0.5 * (np.trace(E) - 1.0), -1.0, 1.0)))
    vee = np.array([E[2, 1] - E[1, 2], E[0, 2] - E[2, 0], E[1, 0] - E[0, 1]])
    if angle < 1e-6:
        w = 0.5 * vee
    elif abs(angle - np.pi) < 1e-6:
        # half turn: the reference's rule (kinematics/trac_ik.py:678-700) - component k = argmax diag is 1, every other component j
        # is E[k, j] / (1 + E[k, k]) (0 when that denominator degenerates), normalised
        k = int(np.argmax(np.diag(E)))
        denom = 1.0 + E[k, k]
        axis = np.array([1.0 if j == k else (E[k, j] / denom if abs(denom) > 1e-6 else 0.0) for j in range(3)])
        w = angle * axis / (np.linalg.norm(axis) + 1e-10)
    else:
        w = angle * vee / (2.0 * np.sin(angle) + 1e-10)
    return np.concatenate([Tc[:3, :3] @ w, dp]), angle, float(np.linalg.norm(dp))


class TracIKSolver:
    def __init__(self, fk_func: Callable, jacobian_func: Callable, joint_limits, n_joints: int,
                 error_func: Optional[Callable] = None, robot: Any = None) -> None:
        self.fk_func = fk_func
        self.jacobian_func = jacobian_func
        self.joint_limits = list(joint_limits)
        self.n_joints = int(n_joints)
        self.error_func = error_func or _pose_error
        self.robot = robot   # a SerialManipulator: the guesses then run as one batched launch
        self.bounds = [(-2 * np.pi if lo is None else lo, 2 * np.pi if hi is None else hi) for lo, hi in self.joint_limits]

    # ---- initial guesses (reference :282-312)
    def _initial_guesses(self, T_desired, theta0, num_restarts: int) -> List[np.ndarray]:
        mid = ik_helpers.midpoint_of_limits(self.joint_limits)
        first = (np.array(theta0, dtype=np.float64) if theta0 is not None
                 else ik_helpers.workspace_heuristic_guess(T_desired, self.n_joints, self.joint_limits))
        guesses = [first, mid, np.zeros(self.n_joints), ik_helpers.clip_to_limits(-mid, self.joint_limits)]
        for _ in range(max(0, int(num_restarts) - 4)):
            guesses.append(ik_helpers.random_in_limits(self.joint_limits))
        return guesses

    def _error_of(self, theta, T_desired) -> Tuple[float, float, float]:
        _, rot, tr = self.error_func(np.asarray(self.fk_func(theta)), T_desired)
        return float(rot) + float(tr), rot, tr

    # ---- host damped least squares on one guess (reference :314-504, without the oscillation heuristics)
    def _dls_host(self, T_desired, theta0, eomg, ev, budget, max_perturbations=3):
        lo = np.array([b[0] for b in self.bounds]); hi = np.array([b[1] for b in self.bounds])
        theta = np.array(theta0, dtype=np.float64)
        damping, nu, step_cap = _DLS_DAMPING, 2.0, _DLS_STEP_CAP
        best, best_err, prev_err, stall, perturbed = theta.copy(), np.inf, np.inf, 0, 0
        t0 = time.perf_counter()
        for _ in range(_DLS_MAX_ITERS):
            if time.perf_counter() - t0 > budget:
                break
            V, rot, tr = self.error_func(np.asarray(self.fk_func(theta)), T_desired)
            err = float(rot) + float(tr)
            if rot < eomg and tr < ev:
                return theta, True, err
            if err < best_err:
                best, best_err, stall = theta.copy(), err, 0
            else:
                stall += 1
            if stall > 20:
                perturbed += 1
                if perturbed > max_perturbations:
                    break
                theta = np.clip(best + 0.1 * np.random.randn(self.n_joints), lo, hi)
                damping, nu, stall = _DLS_DAMPING, 2.0, 0
                continue
            if err < 0.75 * prev_err:
                damping, step_cap, nu = max(1e-6, damping / 3.0), min(0.45, step_cap * 1.2), 2.0
            elif err > prev_err:
                damping, nu, step_cap = min(0.5, damping * nu), min(nu * 1.5, 8.0), max(0.01, step_cap * 0.7)
            prev_err = err
            J = np.asarray(self.jacobian_func(theta), dtype=np.float64)
            step = J.T @ np.linalg.solve(J @ J.T + (damping * damping + 1e-12) * np.eye(6), np.asarray(V, dtype=np.float64))
            norm = float(np.linalg.norm(step))
            if norm > step_cap:
                step *= step_cap / norm
            theta = np.clip(theta + step, lo, hi)
        return best, False, best_err

    # ---- SLSQP polish inside the limits (reference :534-605)
    def _sqp(self, T_desired, theta0, eomg, ev, budget):
        try:
            from scipy.optimize import minimize
        except ImportError:
            return None
        t0 = time.perf_counter()

        class _Stop(Exception):
            pass

        def cost(th):   # the reference's objective: rotation angle^2 + translation norm^2
            if time.perf_counter() - t0 > budget:
                raise _Stop
            _, rot, tr = self.error_func(np.asarray(self.fk_func(th)), T_desired)
            return float(rot) ** 2 + float(tr) ** 2

        # d/dtheta of |V|^2 with dV/dtheta = -J (space Jacobian, [omega; v] order).  DELIBERATE difference: the reference hands SLSQP
        # +2 J^T V (kinematics/trac_ik.py:570-577) - the gradient with the sign flipped, an ascent direction its line search has to
        # reject; only reached when no damped-least-squares row converged, and never changes what a converged solve returns
        def grad(th):
            V, _, _ = self.error_func(np.asarray(self.fk_func(th)), T_desired)
            return -2.0 * np.asarray(self.jacobian_func(th), dtype=np.float64).T @ np.asarray(V, dtype=np.float64)

        x = np.array(theta0, dtype=np.float64)
        try:
            res = minimize(cost, x, method="SLSQP", jac=grad, bounds=self.bounds, options={"ftol": 1e-8, "maxiter": 500, "disp": False})
            x = np.asarray(res.x, dtype=np.float64)
        except _Stop:
            pass
        except Exception:   # SLSQP failing is not an error of the solve: the starting configuration stands
            x = np.array(theta0, dtype=np.float64)
        _, rot, tr = self.error_func(np.asarray(self.fk_func(x)), T_desired)
        return x, bool(rot < eomg and tr < ev), float(rot) + float(tr)

    def solve(self, T_desired, theta0=None, timeout: float = 0.2, eomg: float = 1e-4, ev: float = 1e-4, num_restarts: int = 5,
              use_parallel: bool = False) -> Tuple[np.ndarray, bool, float]:
        """(theta (n,) float64 - the best found if unsuccessful, success, solve time in seconds)."""
        start = time.perf_counter()
        T = np.asarray(T_desired, dtype=np.float64)
        if T.shape != (4, 4):
            raise ValueError(f"T_desired must be (4, 4), got {T.shape}")
        guesses = self._initial_guesses(T, theta0, num_restarts)

        def remaining():
            return max(0.0, timeout - (time.perf_counter() - start))

        best_theta, best_ok, best_err = None, False, np.inf

        def offer(theta, ok, err):
            nonlocal best_theta, best_ok, best_err
            if (ok and (not best_ok or err < best_err)) or (not ok and not best_ok and err < best_err):
                best_theta, best_ok, best_err = np.asarray(theta, dtype=np.float64), bool(ok), float(err)

        if self.robot is not None:   # every guess is a row of one launch
            G = np.stack(guesses)
            th, ok, _ = self.robot.batch_inverse_kinematics(np.broadcast_to(T, (len(G), 4, 4)).copy(), G, eomg, ev, _DLS_MAX_ITERS,
                                                            _DLS_DAMPING, _DLS_STEP_CAP, 1.0, 1.0, True, True)
            for row, good in zip(th, ok):
                err, rot, tr = self._error_of(row, T)
                offer(row, bool(good) and rot < eomg and tr < ev, err)
        else:                        # the reference's sequential budgeting (:232-267)
            per_guess = timeout * 0.8 / max(len(guesses) - 1, 1)
            for g in guesses:
                if best_ok or remaining() < 0.005:
                    break
                offer(*self._dls_host(T, g, eomg, ev, min(per_guess, remaining() * 0.9)))
        if not best_ok and remaining() > 0.01:
            polished = self._sqp(T, best_theta if best_theta is not None else guesses[0], eomg, ev, remaining())
            if polished is not None:
                offer(*polished)
        if best_theta is None:
            best_theta = np.asarray(guesses[0], dtype=np.float64)
        return best_theta, best_ok, time.perf_counter() - start


def trac_ik_solve(robot: Any, T_desired, theta0=None, timeout: float = 0.2, eomg: float = 1e-4, ev: float = 1e-4,
                  num_restarts: int = 5, use_parallel: bool = False) -> Tuple[np.ndarray, bool, float]:
    """TRAC-IK for a SerialManipulator (reference kinematics/trac_ik.py:717-756); the guesses run as one batched launch."""
    solver = TracIKSolver(fk_func=lambda th: robot.forward_kinematics(th, frame="space"),
                          jacobian_func=lambda th: robot.jacobian(th, frame="space"),
                          joint_limits=robot.joint_limits, n_joints=len(robot.joint_limits),
                          robot=robot if hasattr(robot, "batch_inverse_kinematics") else None)
    return solver.solve(T_desired, theta0, timeout, eomg, ev, num_restarts, use_parallel)
