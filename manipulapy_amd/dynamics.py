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
analytic recursion does not need) and no legacy (Mlist_per_link=None) approximation.
"""
from __future__ import annotations

import logging
import os
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
                "ManipulatorDynamics without Mlist_per_link: the reference's legacy approximation "
                "(dynamics/mass_matrix.py:101-132) is documented as incorrect and is not reproduced; construct "
                "the dynamics with per-link CoM transforms (URDFToSerialManipulator does).")
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
        if rows >= 16384 and os.environ.get("MANIPULAPY_HIP_SPECIALIZE", "1") != "0" and not self._spec_tried:
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

    # ---- public API
    def mass_matrix(self, thetalist) -> np.ndarray:
        """(n, n) mass matrix, or (rows, n, n) for a 2-D `thetalist`."""
        q = np.atleast_2d(np.asarray(thetalist, dtype=np.float64))
        M = execute_registered_kernel("dynamics.mass_matrix", self._model_for(q.shape[0]), q)
        return M if np.ndim(thetalist) == 2 else M[0]

    def velocity_quadratic_forces(self, thetalist, dthetalist) -> np.ndarray:
        q = np.asarray(thetalist, dtype=np.float64)[None, :]
        qd = np.asarray(dthetalist, dtype=np.float64)[None, :]
        return self._id(q, qd, np.zeros_like(q), _ZERO3, None)[0]

    def gravity_forces(self, thetalist, g=None) -> np.ndarray:
        q = np.asarray(thetalist, dtype=np.float64)[None, :]
        g = [0.0, 0.0, -9.81] if g is None else g
        z = np.zeros_like(q)
        return self._id(q, z, z, g, None)[0]

    def inverse_dynamics(self, thetalist, dthetalist, ddthetalist, g, Ftip) -> np.ndarray:
        q = np.asarray(thetalist, dtype=np.float64)[None, :]
        qd = np.asarray(dthetalist, dtype=np.float64)[None, :]
        qdd = np.asarray(ddthetalist, dtype=np.float64)[None, :]
        return self._id(q, qd, qdd, g, Ftip)[0]

    def forward_dynamics(self, thetalist, dthetalist, taulist, g, Ftip) -> np.ndarray:
        """qdd (n,), or (rows, n) for 2-D inputs (one g / Ftip for all rows)."""
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
