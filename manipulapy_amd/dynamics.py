"""ManipulatorDynamics — host-side mirror of ManipulaPy/dynamics/manipulator_dynamics.py.

Same constructor signature (dynamics/manipulator_dynamics.py:43-86) and the same public methods:
mass_matrix (mass_matrix.py:16-99), velocity_quadratic_forces / gravity_forces (forces.py:26-133),
inverse_dynamics / forward_dynamics (id_fd.py:16-83), partial_derivative (forces.py:16-24).
Every method runs a registered operation through the kernel registry (float64): its HIP kernel with the "hip" backend
active, its CPU launcher (the C ABI's *_cpu twin, same per-row code) with the NumPy backend active:

    inverse_dynamics(q, qd, qdd, g, F)      "dynamics.inverse_trajectory"   ID(q, qd, qdd, g, F)
    gravity_forces(q, g)                    "dynamics.inverse_trajectory"   ID(q, 0, 0, g, 0)
    velocity_quadratic_forces(q, qd)        "dynamics.inverse_trajectory"   ID(q, qd, 0, 0, 0)
    mass_matrix(q)                          "dynamics.mass_matrix"          columns ID(q, 0, e_j, 0, 0), symmetrised
    forward_dynamics(q, qd, tau, g, F)      "dynamics.forward"              solve(M, tau - ID(q, qd, 0, g, F))

which are exact identities of tau = M qdd + c + g + Js^T F.  2-D inputs (rows, n) evaluate all rows
in one launch.  There is no value-keyed cache (the
reference's caches exist to amortise its 1 + 2n mass-matrix evaluations per point, which the
analytic recursion does not need).

The legacy object (Mlist_per_link=None: what URDF.to_manipulator_dynamics() and hand-built models give,
reference urdf/core.py:795-817) is reproduced, not rejected: the reference evaluates it with an approximation it documents
as incorrect and warns about (dynamics/mass_matrix.py:45-57, :101-132, forces.py:81-95, :136-154).  That
approximation is not rigid-body dynamics, so it cannot be a compiled link-frame model; it runs on the host in NumPy under
every backend (same formulas, same warnings, pinned by tests/golden/legacy_dynamics.npz).
"""
from __future__ import annotations

import logging
import os
import warnings
from typing import Optional

import numpy as np

from . import _hip
from .kinematics import SerialManipulator
from .registry import execute_registered_kernel

__all__ = ["ManipulatorDynamics"]

logger = logging.getLogger("ManipulaPy.dynamics")

_ZERO3 = np.zeros(3)


class ManipulatorDynamics(SerialManipulator):
    def __init__(self, M_list, omega_list, r_list, b_list, S_list, B_list, Glist, Mlist_per_link=None) -> None:
        super().__init__(M_list, omega_list, r_list, b_list, S_list, B_list)
        self.Glist = Glist
        self.Mlist_per_link = Mlist_per_link
        self._dyn_model: Optional[_hip.HipModel] = None
        self._spec_tried = False

    # ---- compiled model
    def hip_model(self, joint_limits=None, torque_limits=None) -> _hip.HipModel:
        """Compile (once) the model the kernels consume.  Limits, if given, build a separate model
        (the planner passes its own float32 limits)."""
        if self.Mlist_per_link is None:
            raise NotImplementedError(
                "ManipulatorDynamics without Mlist_per_link has no compiled model: its legacy approximation "
                "(dynamics/mass_matrix.py:101-132) is evaluated on the host (mass_matrix / gravity_forces / inverse_dynamics / "
                "forward_dynamics and the planner's trajectory methods do that by themselves)")
        if joint_limits is None and torque_limits is None:
            if self._dyn_model is None:
                self._dyn_model = _hip.HipModel(self.S_list, np.asarray(self.Mlist_per_link), np.asarray(self.Glist), self._M_ee)
            return self._dyn_model
        return _hip.HipModel(self.S_list, np.asarray(self.Mlist_per_link), np.asarray(self.Glist), self._M_ee,
                             joint_limits, torque_limits)

    def _model_for(self, rows: int) -> _hip.HipModel:
        """The shared model; for big batches it is specialised first (~2 s once, cached on disk: the per-row forward
        dynamics kernel runs 3x faster with this robot's constants baked in)."""
        model = self.hip_model()
        if (rows >= 16384 and os.environ.get("MANIPULAPY_HIP_SPECIALIZE", "1") != "0" and not self._spec_tried
                and model.n <= _hip.MP_MAX_DOF):   # (9..32 joints run the looped generic kernels: nothing to specialise)
            from .registry import _hip_routing_enabled, get_context

            if _hip_routing_enabled():
                self._spec_tried = True
                try:
                    get_context().specialize(model)
                except _hip.HipError as exc:  # e.g. no hiprtc on this machine: the generic GPU kernels serve
                    logger.warning("kernel specialisation unavailable (%s); using the generic kernels", exc)
        return model

    def _id(self, q, qd, qdd, g, Ftip) -> np.ndarray:
        return execute_registered_kernel("dynamics.inverse_trajectory", self._model_for(np.shape(q)[0]), q, qd, qdd, g, Ftip,
                                         dtype=np.float64)

    # ---- the legacy (Mlist_per_link=None) approximation, on the host
    @property
    def _legacy(self) -> bool:
        return self.Mlist_per_link is None

    def _mass_matrix_legacy(self, q: np.ndarray) -> np.ndarray:
        """Row i = J_i^T (Ad_i^T G_i Ad_i) J_s with Ad_i = Ad(FK(q[:i + 1])), symmetrised
        (reference dynamics/mass_matrix.py:101-132; documented there as incorrect, kept for hand-built models)."""
        from .utils import adjoint_transform

        n = len(q)
        J = self.jacobian(q, frame="space")
        M = np.zeros((n, n))
        for i in range(n):
            Ad = adjoint_transform(self.forward_kinematics(q[: i + 1], frame="space"))
            M[i] = J[:, i] @ (Ad.T @ np.asarray(self.Glist[i], dtype=np.float64) @ Ad) @ J
        return 0.5 * (M + M.T)

    def _gravity_forces_legacy(self, q: np.ndarray, g: np.ndarray) -> np.ndarray:
        """(R_i^T g) . (column sums of G_i's inertia block), R_i from FK(q[:i + 1]) (reference dynamics/forces.py:136-154)."""
        out = np.zeros(len(q))
        for i in range(len(q)):
            R = self.forward_kinematics(q[: i + 1], "space")[:3, :3]
            out[i] = (R.T @ g[:3]) @ np.asarray(self.Glist[i], dtype=np.float64)[:3, :3].sum(axis=0)
        return out

    def _warn_legacy(self, what: str, fix: str) -> None:
        warnings.warn(f"{what} called without Mlist_per_link \u2014 using legacy approximation (incorrect for non-trivial "
                      f"robots). Construct ManipulatorDynamics via URDFToSerialManipulator to get accurate {fix}.", stacklevel=3)

    def _velocity_quadratic_legacy(self, q: np.ndarray, qd: np.ndarray, epsilon: float = 1e-6) -> np.ndarray:
        """Christoffel form on the central difference of the (legacy) mass matrix (reference dynamics/cache.py:23-56,
        forces.py:45-59)."""
        n = len(q)
        dM = np.zeros((n, n, n))
        for k in range(n):
            e = np.zeros(n)
            e[k] = epsilon
            dM[:, :, k] = (self.mass_matrix(q + e) - self.mass_matrix(q - e)) / (2.0 * epsilon)
        c = np.zeros(n)
        for i in range(n):
            gamma = 0.5 * (dM[i] + dM[i].T - dM[:, :, i])
            c[i] = qd @ gamma @ qd
        return c

    # ---- public API
    def mass_matrix(self, thetalist) -> np.ndarray:
        """(n, n) mass matrix, or (rows, n, n) for a 2-D `thetalist`."""
        if self._legacy:
            self._warn_legacy("mass_matrix", "mass matrix")
            q = np.asarray(thetalist, dtype=np.float64)
            return self._mass_matrix_legacy(q) if q.ndim == 1 else np.stack([self._mass_matrix_legacy(r) for r in q])
        q = np.atleast_2d(np.asarray(thetalist, dtype=np.float64))
        M = execute_registered_kernel("dynamics.mass_matrix", self._model_for(q.shape[0]), q)
        return M if np.ndim(thetalist) == 2 else M[0]

    def velocity_quadratic_forces(self, thetalist, dthetalist) -> np.ndarray:
        if self._legacy:
            self._warn_legacy("mass_matrix", "mass matrix")   # (the reference warns once per uncached evaluation: 2n times here)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                return self._velocity_quadratic_legacy(np.asarray(thetalist, dtype=np.float64), np.asarray(dthetalist, dtype=np.float64))
        q = np.asarray(thetalist, dtype=np.float64)[None, :]
        qd = np.asarray(dthetalist, dtype=np.float64)[None, :]
        return self._id(q, qd, np.zeros_like(q), _ZERO3, None)[0]

    def gravity_forces(self, thetalist, g=None) -> np.ndarray:
        g = [0.0, 0.0, -9.81] if g is None else g
        if self._legacy:
            self._warn_legacy("gravity_forces", "gravity compensation")
            return self._gravity_forces_legacy(np.asarray(thetalist, dtype=np.float64), np.asarray(g, dtype=np.float64))
        q = np.asarray(thetalist, dtype=np.float64)[None, :]
        z = np.zeros_like(q)
        return self._id(q, z, z, g, None)[0]

    def _legacy_terms(self, q, qd, g):
        """M, c, g-forces and Js^T of the legacy model (each warns as the reference's does)."""
        M = self.mass_matrix(q)
        c = self.velocity_quadratic_forces(q, qd)
        gf = self.gravity_forces(q, g)
        return M, c, gf, self.jacobian(q).T

    def inverse_dynamics(self, thetalist, dthetalist, ddthetalist, g, Ftip) -> np.ndarray:
        if self._legacy:   # M qdd + c + g + Js^T Ftip on the legacy terms (reference dynamics/id_fd.py:36-48)
            q, qd = np.asarray(thetalist, dtype=np.float64), np.asarray(dthetalist, dtype=np.float64)
            M, c, gf, Jt = self._legacy_terms(q, qd, g)
            return M @ np.asarray(ddthetalist, dtype=np.float64) + c + gf + Jt @ np.asarray(Ftip, dtype=np.float64)
        q = np.asarray(thetalist, dtype=np.float64)[None, :]
        qd = np.asarray(dthetalist, dtype=np.float64)[None, :]
        qdd = np.asarray(ddthetalist, dtype=np.float64)[None, :]
        return self._id(q, qd, qdd, g, Ftip)[0]

    def forward_dynamics(self, thetalist, dthetalist, taulist, g, Ftip) -> np.ndarray:
        """qdd (n,), or (rows, n) for 2-D inputs (one g / Ftip for all rows)."""
        if self._legacy:   # solve(M, tau - c - g - Js^T Ftip) (reference dynamics/id_fd.py:71-83)
            q, qd = np.asarray(thetalist, dtype=np.float64), np.asarray(dthetalist, dtype=np.float64)
            M, c, gf, Jt = self._legacy_terms(q, qd, g)
            return np.linalg.solve(M, np.asarray(taulist, dtype=np.float64) - c - gf - Jt @ np.asarray(Ftip, dtype=np.float64))
        q = np.atleast_2d(np.asarray(thetalist, dtype=np.float64))
        qd = np.atleast_2d(np.asarray(dthetalist, dtype=np.float64))
        tau = np.atleast_2d(np.asarray(taulist, dtype=np.float64))
        qdd = execute_registered_kernel("dynamics.forward", self._model_for(q.shape[0]), q, qd, tau, g, Ftip)
        return qdd if np.ndim(thetalist) == 2 else qdd[0]

    def partial_derivative(self, i: int, j: int, k: int, thetalist, epsilon: float = 1e-6) -> float:
        """dM[i, j] / dtheta_k by the reference's central difference (dynamics/cache.py:39-52)."""
        q = np.asarray(thetalist, dtype=np.float64)
        e = np.zeros_like(q)
        e[k] = epsilon
        return float((self.mass_matrix(q + e)[i, j] - self.mass_matrix(q - e)[i, j]) / (2.0 * epsilon))
