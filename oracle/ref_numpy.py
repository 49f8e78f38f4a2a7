"""CPU ORACLE — test infrastructure, NOT product code.

A NumPy restatement of the reference's (boelnasr/ManipulaPy v1.4.1) NumPy-CPU algorithm for
the batched trajectory + rigid-body-dynamics hot path.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module; the
product path (manipulapy_amd/) never does and fails loudly without its HIP library.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function below against
  * the reference's own test data files tests/data/dynamics_golden_{ur5,panda}.npz
    (kept byte-for-byte in tests/golden/), with the reference's own tolerances
    (tests/test_dynamics_golden.py:77-83: rtol 1e-7, atol 1e-9 / 1e-8), and
  * fixtures produced by importing the reference in the build container
    (tests/golden/make_golden.py): model tables, FK, Jacobians, M, c, g, ID, FD for
    ur5 / iiwa14 / panda / xarm6, planner-level trajectory dumps.

Every function cites the reference file:line (relative to the reference repo root,
ManipulaPy/...) whose arithmetic it follows.  The algorithm is deliberately the
reference's O(n^2)-per-mass-matrix, (1+2n)-mass-matrices-per-point formulation
(tau = M qdd + c + g + Js^T F with a central-difference Christoffel c), not RNEA: it is the
definition of "correct" that the HIP kernels are compared against.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

FD_EPS = 1e-6  # dynamics/cache.py:24 (epsilon of the central difference)
G_DEFAULT = np.array([0.0, 0.0, -9.81])  # dynamics/forces.py:78, planning/trajectory_dynamics.py:54


# --------------------------------------------------------------------------- model tables
@dataclass
class RobotTables:
    """The constant tables the reference's hot path consumes (urdf/core.py:670-769)."""

    S: np.ndarray  # (6, n) space screws [w; v]
    M_ee: np.ndarray  # (4, 4) home pose of the end effector
    G: np.ndarray  # (n, 6, 6) spatial inertia per link, twist order [w; v]
    Mcom: np.ndarray  # (n, 4, 4) home pose of each link's CoM frame (Mlist_per_link)
    joint_limits: np.ndarray  # (n, 2)
    B: Optional[np.ndarray] = None  # (6, n) body screws
    name: str = ""

    @property
    def n(self) -> int:
        return self.S.shape[1]

    def body_screws(self) -> np.ndarray:
        # urdf/core.py:755-758: B = Ad(M^-1) S
        if self.B is not None:
            return self.B
        return adjoint(np.linalg.inv(self.M_ee)) @ self.S


def load_tables(path: str) -> RobotTables:
    z = np.load(path)
    return RobotTables(
        S=z["S_list"].astype(np.float64),
        M_ee=z["M_ee"].astype(np.float64),
        G=z["Glist"].astype(np.float64),
        Mcom=z["Mlist_per_link"].astype(np.float64),
        joint_limits=z["joint_limits"].astype(np.float64),
        B=z["B_list"].astype(np.float64) if "B_list" in z.files else None,
        name=str(z["ee_name"]) if "ee_name" in z.files else "",
    )


# --------------------------------------------------------------------------- se(3) pieces
def skew(v):
    """utils/so3.py:22-30."""
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]], dtype=np.result_type(v, np.float32))


def exp_twist(S, theta):
    """utils/se3.py:33-42 — se(3) exponential for a unit-|w| (or w=0) screw.

    R = I + sin(t)[w] + (1-cos(t))[w]^2 ;  p = (t I + (1-cos t)[w] + (t - sin t)[w]^2) v.
    The trig runs in the dtype of `theta`, as in the reference.
    """
    S = np.asarray(S)
    W = skew(S[:3])
    W2 = W @ W
    s, c = np.sin(theta), np.cos(theta)
    R = np.eye(3) + s * W + (1 - c) * W2
    Gm = np.eye(3) * theta + (1 - c) * W + (theta - s) * W2
    T = np.zeros((4, 4), dtype=R.dtype)
    T[:3, :3] = R
    T[:3, 3] = Gm @ S[3:]
    T[3, 3] = 1.0
    return T


def adjoint(T):
    """utils/se3.py:45-52 — [[R, 0], [[p]R, R]] for twists ordered [w; v]."""
    R, p = T[:3, :3], T[:3, 3]
    A = np.zeros((6, 6), dtype=R.dtype)
    A[:3, :3] = R
    A[3:, 3:] = R
    A[3:, :3] = skew(p) @ R
    return A


# --------------------------------------------------------------------------- kinematics
def _theta_array(theta):
    # kinematics/fk.py:55-58: backend arrays keep their dtype, everything else -> float64
    th = np.asarray(theta)
    if not isinstance(theta, np.ndarray) or th.dtype.kind in "biu":
        th = np.asarray(theta, dtype=np.float64)
    return th


def fk_space(tab: RobotTables, theta):
    """kinematics/fk.py:59-70 — T = prod_i exp([S_i] th_i) . M ; accepts a truncated theta."""
    th = _theta_array(theta)
    T = np.eye(4, dtype=th.dtype)
    for i in range(len(th)):
        T = T @ exp_twist(tab.S[:, i], th[i])
    return T @ tab.M_ee


def fk_body(tab: RobotTables, theta):
    """kinematics/fk.py:72-81 — T = M . prod_i exp([B_i] th_i)."""
    th = _theta_array(theta)
    B = tab.body_screws()
    T = np.eye(4, dtype=th.dtype)
    for i in range(len(th)):
        T = T @ exp_twist(B[:, i], th[i])
    return tab.M_ee @ T


def jacobian_space(tab: RobotTables, theta):
    """kinematics/jacobian.py:62-73 — column i = Ad(prod_{j<i} exp) S_i."""
    th = _theta_array(theta)
    if len(th) == 0:
        return np.zeros((6, 0), dtype=th.dtype)
    T = np.eye(4, dtype=th.dtype)
    cols = []
    for i in range(len(th)):
        cols.append(adjoint(T) @ tab.S[:, i])
        T = T @ exp_twist(tab.S[:, i], th[i])
    return np.stack(cols, axis=1)


def jacobian_body(tab: RobotTables, theta):
    """kinematics/jacobian.py:74-90 — last column B_n, earlier ones Ad(prod exp(-B_{j} th_{j})) B_i."""
    th = _theta_array(theta)
    B = tab.body_screws()
    n = len(th)
    T = np.eye(4, dtype=th.dtype)
    cols = [None] * n
    cols[n - 1] = B[:, n - 1]
    for i in range(n - 2, -1, -1):
        T = T @ exp_twist(B[:, i + 1], -th[i + 1])
        cols[i] = adjoint(T) @ B[:, i]
    return np.stack(cols, axis=1)


# --------------------------------------------------------------------------- dynamics
def _link_com_jacobians(tab: RobotTables, theta):
    """Shared by mass_matrix / gravity_forces (dynamics/mass_matrix.py:62-91, forces.py:100-120).

    For link k: T_k_com = FK(theta[:k+1]) . inv(FK(0_{k+1})) . Mcom_k, and
    J_k[:, :k+1] = Ad(inv(T_k_com)) . J_s[:, :k+1], zero for downstream joints.
    Yields (k, T_k_com, J_k).
    """
    n = len(theta)
    Js = jacobian_space(tab, theta)
    for k in range(n):
        T_k = fk_space(tab, theta[: k + 1])
        T_k0 = fk_space(tab, np.zeros(k + 1))
        T_k_com = T_k @ (np.linalg.inv(T_k0) @ tab.Mcom[k])
        Jk = np.zeros((6, n), dtype=np.float64)
        Jk[:, : k + 1] = adjoint(np.linalg.inv(T_k_com)) @ Js[:, : k + 1]
        yield k, T_k_com, Jk


def mass_matrix(tab: RobotTables, theta):
    """dynamics/mass_matrix.py:62-99 — M = sum_k J_k^T G_k J_k, then 0.5 (M + M^T)."""
    n = len(theta)
    M = np.zeros((n, n))
    for k, _, Jk in _link_com_jacobians(tab, theta):
        M = M + Jk.T @ tab.G[k] @ Jk
    return 0.5 * (M + M.T)


def mass_matrix_derivatives(tab: RobotTables, theta, eps: float = FD_EPS):
    """dynamics/cache.py:39-52 — dM[:, :, k] = (M(th + eps e_k) - M(th - eps e_k)) / (2 eps)."""
    theta = np.asarray(theta, dtype=np.float64)
    n = len(theta)
    dM = np.zeros((n, n, n))
    for k in range(n):
        e = np.zeros(n)
        e[k] = 1.0
        dM[:, :, k] = (mass_matrix(tab, theta + eps * e) - mass_matrix(tab, theta - eps * e)) / (2.0 * eps)
    return dM


def velocity_quadratic_forces(tab: RobotTables, theta, dtheta):
    """dynamics/forces.py:45-58 — c_i = dth^T Gamma_i dth, Gamma_i = 0.5 (dM_i + dM_i^T - dM[:, :, i])."""
    dth = np.asarray(dtheta, dtype=np.float64)
    dM = mass_matrix_derivatives(tab, theta)
    n = len(dth)
    c = np.zeros(n)
    for i in range(n):
        gamma = 0.5 * (dM[i] + dM[i].T - dM[:, :, i])
        c[i] = dth @ (gamma @ dth)
    return c


def gravity_forces(tab: RobotTables, theta, g=None):
    """dynamics/forces.py:100-133 — sum_k J_k^T [0; m_k R_k^T (-g)], m_k = G_k[3, 3]."""
    g = G_DEFAULT if g is None else np.asarray(g, dtype=np.float64)
    n = len(theta)
    out = np.zeros(n)
    for k, T_k_com, Jk in _link_com_jacobians(tab, theta):
        m_k = tab.G[k][3, 3]
        F = np.concatenate((np.zeros(3), m_k * (T_k_com[:3, :3].T @ (-g))))
        out = out + Jk.T @ F
    return out


def inverse_dynamics(tab: RobotTables, theta, dtheta, ddtheta, g, Ftip):
    """dynamics/id_fd.py:37-48 — tau = M qdd + c + g + J_s^T Ftip (Ftip is a SPACE-frame wrench)."""
    M = mass_matrix(tab, theta)
    c = velocity_quadratic_forces(tab, theta, dtheta)
    gf = gravity_forces(tab, theta, g)
    Jt = jacobian_space(tab, theta).T
    return M @ np.asarray(ddtheta) + c + gf + Jt @ np.asarray(Ftip)


def forward_dynamics(tab: RobotTables, theta, dtheta, tau, g, Ftip):
    """dynamics/id_fd.py:71-83 — qdd = solve(M, tau - c - g - J_s^T Ftip)."""
    M = mass_matrix(tab, theta)
    c = velocity_quadratic_forces(tab, theta, dtheta)
    gf = gravity_forces(tab, theta, g)
    Jt = jacobian_space(tab, theta).T
    rhs = np.asarray(tau) - c - gf - Jt @ np.asarray(Ftip)
    return np.linalg.solve(M, rhs)


# --------------------------------------------------------------------------- time scaling
def time_scaling(tau, Tf, method):
    """planning/trajectory.py:51-68 (numba path): cubic / quintic s, s', s''; anything else -> zeros.

    `tau` may be an array; all math in float64.
    """
    tau = np.asarray(tau, dtype=np.float64)
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
        s = sd = sdd = np.zeros_like(tau)
    return s, sd, sdd


def trajectory_points(thetastart, thetaend, Tf, N, method):
    """planning/trajectory.py:15-75 (`_trajectory_cpu_fallback`, numba semantics).

    float32 endpoints, float32 difference, float64 scalar math, float32 store.
    t = idx * (Tf / (N - 1)), tau = t / Tf.  (N = 1 divides by zero in the reference too.)
    Returns (pos, vel, acc), each (N, n) float32, WITHOUT the joint-limit clip.
    """
    a = np.asarray(thetastart, dtype=np.float32)
    b = np.asarray(thetaend, dtype=np.float32)
    idx = np.arange(N, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = idx * (Tf / (N - 1)) if N > 1 else idx * np.inf
        tau = t / Tf
    s, sd, sdd = time_scaling(tau, float(Tf), method)
    d = (b - a).astype(np.float64)  # float32 subtraction first, as numba types it
    pos = (s[:, None] * d[None, :] + a.astype(np.float64)[None, :]).astype(np.float32)
    vel = (sd[:, None] * d[None, :]).astype(np.float32)
    acc = (sdd[:, None] * d[None, :]).astype(np.float32)
    return pos, vel, acc


def trajectory_points_numpy_twin(thetastart, thetaend, Tf, N, method):
    """cuda_kernels/trajectory_kernels.py:20-88 (`trajectory_cpu_fallback`, the registry's CPU launcher).

    float32 linspace math; other `method` -> linear; N <= 1 or Tf <= 0 -> sit at start.
    """
    a = np.asarray(thetastart)
    b = np.asarray(thetaend)
    if N <= 1 or Tf <= 0.0:
        s = sd = sdd = np.zeros(N, dtype=np.float32)
    else:
        t = np.linspace(0, Tf, N, dtype=np.float32)
        tau = t / Tf
        if method == 3:
            s = 3.0 * tau**2 - 2.0 * tau**3
            sd = 6.0 * tau * (1.0 - tau) / Tf
            sdd = 6.0 * (1.0 - 2.0 * tau) / (Tf * Tf)
        elif method == 5:
            s = 10.0 * tau**3 - 15.0 * tau**4 + 6.0 * tau**5
            sd = (30.0 * tau**2 - 60.0 * tau**3 + 30.0 * tau**4) / Tf
            sdd = (60.0 * tau - 180.0 * tau**2 + 120.0 * tau**3) / (Tf * Tf)
        else:
            s, sd, sdd = tau, np.ones_like(tau) / Tf, np.zeros_like(tau)
    d = b - a
    pos = a[None, :] + s[:, None] * d[None, :]
    return (pos.astype(np.float32), (sd[:, None] * d[None, :]).astype(np.float32),
            (sdd[:, None] * d[None, :]).astype(np.float32))


# --------------------------------------------------------------------------- SO(3) log / exp, Cartesian path
def matrix_log3(R):
    """utils/so3.py:172-191 (+ :36-74 half-turn axis, :116-160 theta/sin theta): rotation VECTOR of log(R).

    theta = atan2(|vee(R - R^T)| / 2, (tr R - 1) / 2).  theta > pi - 1e-2: theta * n with n from the symmetric
    part (column 2 if sym[2,2] >= 1e-6, else column 1 if sym[1,1] >= 1e-6, else column 0; sign of the matching
    vee component, >= 0 -> +).  Otherwise 0.5 (theta / sin theta) vee, Taylor 1 + u/3 + 4u^2/45 in u = 1 - cos
    for cos > 1 - 5e-5."""
    R = np.asarray(R, dtype=np.float64)
    cos_t = np.clip((np.trace(R) - 1) / 2, -1.0, 1.0)
    vee = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    sin_t = np.sqrt(max(vee @ vee, 1e-300)) / 2
    theta = np.arctan2(sin_t, cos_t)
    if theta > np.pi - 1e-2:
        sym = 0.5 * (R + R.T) - cos_t * np.eye(3)
        j = 2 if sym[2, 2] >= 1e-6 else (1 if sym[1, 1] >= 1e-6 else 0)
        cand = sym[:, j]
        axis = cand / np.sqrt(max(cand @ cand, 1e-24))
        return theta * (1.0 if vee[j] >= 0 else -1.0) * axis
    u = 1.0 - cos_t
    if cos_t > 1 - 5e-5:
        coef = 1.0 + u / 3.0 + 4.0 * u * u / 45.0
    else:
        coef = np.arccos(cos_t) / np.sqrt(1.0 - cos_t * cos_t)
    return 0.5 * coef * vee


def matrix_exp3(rotvec):
    """utils/so3.py:199-237 — Rodrigues I + A K + B K^2, A = sin t / t, B = (1 - cos t) / t^2 with the
    t^2 < 1e-4 Taylor branch (1 - t^2/6 + t^4/120, 1/2 - t^2/24 + t^4/720)."""
    w = np.asarray(rotvec, dtype=np.float64)
    t2 = w @ w
    if t2 < 1e-4:
        A, B = 1.0 - t2 / 6.0 + t2 * t2 / 120.0, 0.5 - t2 / 24.0 + t2 * t2 / 720.0
    else:
        t = np.sqrt(t2)
        A, B = np.sin(t) / t, (1 - np.cos(t)) / t2
    K = skew(w)
    return np.eye(3) + A * K + B * (K @ K)


def cartesian_trajectory(Xstart, Xend, Tf, N, method):
    """planning/trajectory.py:504-594 + :676-737.  positions / orientations use cubic for method 3 and QUINTIC
    for anything else; velocities / accelerations use cubic (3), quintic (5), zeros otherwise.  float64 math,
    float32 rows.  N = 1 divides by zero (ZeroDivisionError), as in the reference."""
    Xs, Xe = np.asarray(Xstart, dtype=np.float64), np.asarray(Xend, dtype=np.float64)
    N = int(N)
    timegap = Tf / (N - 1.0)
    Rs, ps, Re, pe = Xs[:3, :3], Xs[:3, 3], Xe[:3, :3], Xe[:3, 3]
    w = matrix_log3(Rs.T @ Re)
    pos, ori, vel, acc = [], [], [], []
    for i in range(N):
        t = timegap * i
        s = 3 * (t / Tf) ** 2 - 2 * (t / Tf) ** 3 if method == 3 else 10 * (t / Tf) ** 3 - 15 * (t / Tf) ** 4 + 6 * (t / Tf) ** 5
        ori.append(Rs @ matrix_exp3(w * s))
        pos.append(s * pe + (1 - s) * ps)
        tau = (i * (Tf / (N - 1))) / Tf
        _, sd, sdd = time_scaling(tau, float(Tf), method)
        vel.append(sd * (pe - ps))
        acc.append(sdd * (pe - ps))
    f32 = lambda a, shape: np.asarray(a, dtype=np.float32).reshape(shape)  # noqa: E731
    return {"positions": f32(pos, (N, 3)), "velocities": f32(vel, (N, 3)), "accelerations": f32(acc, (N, 3)),
            "orientations": f32(ori, (N, 3, 3))}


# --------------------------------------------------------------------------- planner level
def joint_trajectory(joint_limits, thetastart, thetaend, Tf, N, method):
    """planning/trajectory.py:276-333 — generate, then clip POSITIONS to float32 joint limits."""
    lim = np.asarray(joint_limits, dtype=np.float32)  # planning/trajectory_planning.py:218
    pos, vel, acc = trajectory_points(thetastart, thetaend, Tf, N, method)
    pos = np.clip(pos, lim[:, 0], lim[:, 1])
    return {"positions": pos, "velocities": vel, "accelerations": acc}


def batch_joint_trajectory(joint_limits, start_batch, end_batch, Tf, N, method):
    """planning/trajectory.py:431-502 — per-trajectory loop, stack, clip; empty batch -> (0, N, n) zeros."""
    sb = np.asarray(start_batch)
    eb = np.asarray(end_batch)
    B, n = sb.shape
    if B == 0:
        z = np.zeros((0, N, n), dtype=np.float32)
        return {"positions": z, "velocities": z.copy(), "accelerations": z.copy()}
    rows = [trajectory_points(sb[i], eb[i], Tf, N, method) for i in range(B)]
    lim = np.asarray(joint_limits, dtype=np.float32)
    pos = np.clip(np.stack([r[0] for r in rows]), lim[:, 0], lim[:, 1])
    return {"positions": pos, "velocities": np.stack([r[1] for r in rows]),
            "accelerations": np.stack([r[2] for r in rows])}


def inverse_dynamics_trajectory(tab, q, qd, qdd, g=None, Ftip=None, torque_limits=None, dtype=np.float32):
    """planning/trajectory_dynamics.py:308-380 — per-row inverse_dynamics, rows cast to float32, clip.

    `dtype=np.float64` gives the per-point float64 oracle with the same clip.
    """
    g = G_DEFAULT if g is None else np.asarray(g, dtype=np.float64)
    Ftip = np.zeros(6) if Ftip is None else np.asarray(Ftip, dtype=np.float64)
    q = np.asarray(q)
    N, n = q.shape
    out = np.zeros((N, n), dtype=dtype)
    for i in range(N):
        out[i] = inverse_dynamics(tab, q[i], qd[i], qdd[i], g, Ftip).astype(dtype)
    if torque_limits is not None:
        tl = np.asarray(torque_limits, dtype=np.float32)  # planning/trajectory_planning.py:219-223
        out = np.clip(out, tl[:, 0].astype(dtype), tl[:, 1].astype(dtype))
    return out


def forward_dynamics_trajectory(tab, theta0, dtheta0, taumat, g, Ftipmat, dt, intRes, joint_limits=None):
    """planning/trajectory_dynamics.py:580-708 — semi-implicit Euler roll-out.

    Row 0 = initial state (acc 0).  For i >= 1, `intRes` sub-steps of dt/intRes:
    qdd = FD(q, qd, taumat[i], g, Ftipmat[i]); qd += qdd h; q += qd_new h; q = clip(q, float32 limits).
    Rows stored float32; the recorded acceleration is the LAST sub-step's.
    """
    lim = np.asarray(tab.joint_limits if joint_limits is None else joint_limits, dtype=np.float32)
    q = np.asarray(theta0)
    qd = np.asarray(dtheta0)
    taumat = np.asarray(taumat)
    N, n = taumat.shape[0], q.shape[0]
    if N == 0:
        raise IndexError("index 0 is out of bounds for axis 0 with size 0")
    P = [q.astype(np.float32)]
    V = [qd.astype(np.float32)]
    A = [np.zeros(n, dtype=np.float32)]
    h = dt / intRes
    for i in range(1, N):
        last = np.zeros(n, dtype=np.float32)
        for _ in range(intRes):
            qdd = forward_dynamics(tab, q, qd, taumat[i], g, Ftipmat[i])
            qd = (qd + qdd * h).astype(qd.dtype)
            q = (q + qd * h).astype(q.dtype)
            q = np.clip(q, lim[:, 0], lim[:, 1])
            last = qdd
        P.append(q.astype(np.float32))
        V.append(qd.astype(np.float32))
        A.append(np.asarray(last, dtype=np.float32))
    return {"positions": np.stack(P), "velocities": np.stack(V), "accelerations": np.stack(A)}

# --------------------------------------------------------------------------- inverse kinematics
def ik_geometric_error(T_curr, T_target):
    """kinematics/ik.py:88-140 — 6-vector [angular (space frame); linear], rotation angle, translation norm."""
    pos_err = T_target[:3, 3] - T_curr[:3, 3]
    trans_err = np.linalg.norm(pos_err)
    R_curr, R_target = T_curr[:3, :3], T_target[:3, :3]
    R_err = R_curr.T @ R_target
    angle = np.arccos(np.clip((np.trace(R_err) - 1) / 2, -1, 1))
    rot_err = abs(angle)
    vee = np.array([R_err[2, 1] - R_err[1, 2], R_err[0, 2] - R_err[2, 0], R_err[1, 0] - R_err[0, 1]])
    if angle < 1e-6:
        omega = vee / 2
    elif abs(angle - np.pi) < 1e-6:
        omega = angle * np.eye(3)[int(np.argmax(np.diag(R_err)))]
    else:
        omega = angle * vee / (2 * np.sin(angle) + 1e-10)
    return np.concatenate((R_curr @ omega, pos_err)), rot_err, trans_err


def iterative_inverse_kinematics(tab, T_desired, theta0, eomg=1e-6, ev=1e-6, max_iterations=10000, damping=2e-2, step_cap=0.3,
                                 weight_orientation=1.0, weight_position=1.0, joint_limits=None, rng=None, adaptive_tuning=False,
                                 backtracking=False):
    """kinematics/ik.py:39-311: damped least squares through the damped pseudo-inverse
    V diag(s / (s^2 + lambda^2 + 1e-12)) U^T  (:142-162), step cap (:243-246), joint-limit projection (:164-180),
    best-solution tracking (:196-203, :273-280), the stagnation restart (:205-213: after more than 20 iterations without
    a new best, restart from best + 0.1 randn, damping and its growth factor reset), the optional adaptive damping /
    step cap (:215-231) and the optional five-scale line search (:248-265).  The restart draws from
    `rng.standard_normal` here (NumPy's global stream in the reference; pass np.random.RandomState(seed) seeded like the
    caller seeded np.random to reproduce a reference run).  Returns (theta, success, iterations, restarts)."""
    lim = np.asarray(tab.joint_limits if joint_limits is None else joint_limits, dtype=np.float64)
    lower, upper = lim[:, 0], lim[:, 1]
    clip = lambda th: np.minimum(np.maximum(th, lower), upper)
    theta = np.array(theta0, dtype=np.float64)
    T_desired = np.asarray(T_desired, dtype=np.float64)
    best_theta, best_error, stall, restarts = theta.copy(), np.inf, 0, 0
    W = np.array([weight_orientation] * 3 + [weight_position] * 3)
    damping_local, step_cap_local, nu, prev_error = damping, step_cap, 2.0, np.inf
    success, k, current_error = False, -1, np.inf
    for k in range(max_iterations):
        V, rot_err, trans_err = ik_geometric_error(fk_space(tab, theta), T_desired)
        current_error = rot_err + trans_err
        if rot_err < eomg and trans_err < ev:
            success = True
            break
        if current_error < best_error:
            best_error, best_theta, stall = current_error, theta.copy(), 0
        else:
            stall += 1
        if stall > 20:
            noise = (rng.standard_normal(theta.shape[0]) if rng is not None else np.random.randn(theta.shape[0]))
            theta = clip(best_theta + 0.1 * noise)
            damping_local, stall, nu, restarts = damping, 0, 2.0, restarts + 1
            continue
        if adaptive_tuning and k > 0:
            if current_error < prev_error * 0.75:
                damping_local = max(1e-6, damping_local / 3)
                step_cap_local = min(step_cap * 1.5, step_cap_local * 1.2)
                nu = 2.0
            elif current_error < prev_error * 0.95:
                damping_local = max(1e-6, damping_local / 1.5)
            elif current_error > prev_error:
                damping_local = min(5e-1, damping_local * nu)
                nu = min(nu * 1.5, 8)
                step_cap_local = max(0.01, step_cap_local * 0.7)
        prev_error = current_error
        J = jacobian_space(tab, theta)
        U, s, Vt = np.linalg.svd(J, full_matrices=False)
        delta = Vt.T @ ((s / (s ** 2 + damping_local ** 2 + 1e-12)) * (U.T @ (V * W)))
        nd = np.linalg.norm(delta)
        if nd > step_cap_local:
            delta = delta * (step_cap_local / nd)
        if backtracking:
            keep, keep_err = theta, current_error
            for scale in (1.0, 0.5, 0.25, 0.125, 0.75):
                cand = clip(theta + scale * delta)
                _, r_try, t_try = ik_geometric_error(fk_space(tab, cand), T_desired)
                if r_try + t_try < keep_err:
                    keep, keep_err = cand, r_try + t_try
            theta = keep if keep_err < current_error * 1.1 else clip(theta + 0.1 * delta)
        else:
            theta = clip(theta + delta)
    else:
        k += 1
    if not success and best_error < current_error:
        theta = best_theta
        _, rot_err, trans_err = ik_geometric_error(fk_space(tab, theta), T_desired)
        success = bool(rot_err < eomg and trans_err < ev)
    return theta, bool(success), k + 1, restarts


# initial guesses (kinematics/ik_helpers.py:28-114, :179-212, :215-246); limits as (n,2) with +-inf for open ends
def ik_workspace_heuristic_guess(T, n, lim):
    a = np.zeros(n)
    p, R = T[:3, 3], T[:3, :3]
    if n >= 1:
        a[0] = np.arctan2(p[1], p[0])
    if n >= 2:
        r = np.sqrt(p[0] ** 2 + p[1] ** 2)
        a[1] = np.arctan2(p[2], r) if r > 1e-6 else 0.0
    if n >= 3:
        a[2] = np.pi / 4
    if n > 3:
        if abs(R[2, 2]) < 0.9999:
            if n >= 5:
                a[4] = np.arccos(np.clip(R[2, 2], -1, 1))
            a[3] = np.arctan2(R[1, 2], R[0, 2])
            if n >= 6:
                a[5] = np.arctan2(R[2, 1], -R[2, 0])
        else:
            a[3] = np.arctan2(R[1, 0], R[0, 0])
    return np.minimum(np.maximum(a, lim[:, 0]), lim[:, 1])


def ik_random_in_limits(lim):
    out = []
    for mn, mx in lim:
        if np.isfinite(mn) and np.isfinite(mx):
            out.append(np.random.uniform(mn, mx))
        elif np.isfinite(mn):
            out.append(mn + np.random.uniform(0, np.pi))
        elif np.isfinite(mx):
            out.append(mx - np.random.uniform(0, np.pi))
        else:
            out.append(np.random.uniform(-np.pi, np.pi))
    return np.array(out)


def ik_midpoint_of_limits(lim):
    return np.array([(mn + mx) / 2.0 if np.isfinite(mn) and np.isfinite(mx) else 0.0 for mn, mx in lim])


ROBUST_STRATEGIES = (("workspace_heuristic", 0.02, 0.3), ("midpoint", 0.02, 0.3), ("workspace_heuristic", 0.01, 0.4),
                     ("random", 0.02, 0.3), ("random", 0.03, 0.25), ("midpoint", 0.01, 0.4), ("random", 0.015, 0.35),
                     ("random", 0.025, 0.3), ("workspace_heuristic", 0.03, 0.25), ("random", 0.02, 0.35))


def robust_inverse_kinematics(tab, T_desired, joint_limits, max_attempts=10, eomg=2e-3, ev=2e-3, max_iterations=5000):
    """kinematics/ik.py:477-598 — sequential multi-start over ROBUST_STRATEGIES with adaptive tuning and backtracking, first
    success wins, otherwise the attempt with the smallest pose error.  Uses NumPy's GLOBAL random stream for the random
    guesses and the restarts, exactly like the reference (seed np.random before calling to reproduce one of its runs).
    Returns (theta, success, total_iterations, strategy, total_restarts)."""
    lim = np.asarray(joint_limits, dtype=np.float64)
    n = tab.n
    best_theta, best_error, total, winner, restarts = None, np.inf, 0, "none", 0
    for name, damping, cap in ROBUST_STRATEGIES[:min(max_attempts, len(ROBUST_STRATEGIES))]:
        theta0 = (ik_workspace_heuristic_guess(T_desired, n, lim) if name == "workspace_heuristic"
                  else ik_midpoint_of_limits(lim) if name == "midpoint" else ik_random_in_limits(lim))
        theta, ok, it, rs = iterative_inverse_kinematics(tab, T_desired, theta0, eomg, ev, max_iterations, damping, cap,
                                                         joint_limits=lim, adaptive_tuning=True, backtracking=True)
        total += it
        restarts += rs
        if ok:
            return theta, True, total, name, restarts
        Tc = fk_space(tab, theta)
        err = np.linalg.norm(Tc[:3, 3] - T_desired[:3, 3]) + np.arccos(np.clip((np.trace(Tc[:3, :3].T @ T_desired[:3, :3]) - 1) / 2, -1, 1))
        if err < best_error:
            best_error, best_theta, winner = err, theta.copy(), name
    if best_theta is None:
        best_theta = ik_midpoint_of_limits(lim)
    return best_theta, False, total, winner, restarts


# --------------------------------------------------------------------------- fused potential field
def potential_field(positions, goal, obstacles, influence_distance):
    """cuda_kernels/field_kernels.py:113-161 (`potential_field_cpu_fallback`) — attractive 0.5 |p - goal|^2 plus, for every
    obstacle with 0 < d^2 < influence^2, 0.5 (1/d - 1/influence)^2; gradient (p - goal) - (1/d - 1/influence) / d^3 (p - obstacle).
    float32 throughout, obstacles accumulated in order; a zero distance contributes nothing."""
    pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
    goal = np.ascontiguousarray(goal, dtype=np.float32).reshape(3)
    obs = np.ascontiguousarray(obstacles, dtype=np.float32).reshape(-1, 3)
    diff = pos - goal
    pot = np.float32(0.5) * np.sum(diff * diff, axis=1)
    grad = diff.copy()
    inv_infl = np.float32(1.0 / influence_distance) if influence_distance > 0.0 else np.float32(0.0)
    infl2 = np.float32(influence_distance * influence_distance)
    for o in obs:
        od = pos - o
        d2 = np.sum(od * od, axis=1)
        m = (d2 > 0.0) & (d2 < infl2)
        if not m.any():
            continue
        inv = np.float32(1.0) / np.sqrt(d2[m])
        term = inv - inv_infl
        pot[m] += np.float32(0.5) * term * term
        grad[m] += (-term * inv * inv * inv)[:, None] * od[m]
    return pot.astype(np.float32), grad.astype(np.float32)
