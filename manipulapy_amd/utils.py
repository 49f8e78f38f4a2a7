"""se(3) / so(3) helpers, screw-list builders and time scalings as plain NumPy functions.

The public names and semantics of the reference's `ManipulaPy.utils` (utils/so3.py, utils/se3.py, utils/screw.py,
utils/time_scaling.py).  The reference writes them against its array-backend protocol, with masked branches so that
autodiff backends can trace them; here they are host NumPy only (this package's compute path is the HIP library, which
carries its own device versions of the pieces the kernels need).  The numerically delicate choices are the reference's:
angle from atan2(|vee| / 2, cos), Taylor bands near the identity (cos > 1 - 5e-5 for the logarithm, theta^2 < 1e-4 for
the exponentials, theta^2 < 1e-2 for the se(3) logarithm's translation coefficient) and the symmetric-part axis within
1e-2 rad of a half turn.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

__all__ = ["NearZero", "skew_symmetric", "VecToso3", "skew_symmetric_to_vector", "rotation_logm", "MatrixLog3", "MatrixExp3",
           "rotation_matrix_to_euler_angles", "euler_to_rotation_matrix", "transform_from_twist", "adjoint_transform", "logm",
           "se3ToVec", "TransToRp", "TransInv", "MatrixLog6", "MatrixExp6", "VecTose3", "extract_r_list", "extract_omega_list",
           "extract_screw_list", "logm_to_twist", "CubicTimeScaling", "QuinticTimeScaling"]


# ------------------------------------------------------------------------------------------------ so(3)
def NearZero(z: float) -> bool:
    """|z| below the library's 1e-6 zero tolerance (utils/so3.py:15-17)."""
    return abs(z) < 1e-6


def skew_symmetric(v) -> np.ndarray:
    """[v]x (utils/so3.py:20-28)."""
    v = np.asarray(v, dtype=np.float64)
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


VecToso3 = skew_symmetric


def skew_symmetric_to_vector(matrix) -> np.ndarray:
    m = np.asarray(matrix)
    return np.array([m[2, 1], m[0, 2], m[1, 0]])


def _vee(R):
    return np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])


def _cos_and_angle(R) -> Tuple[float, float]:
    cs = float(np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0))
    vee = _vee(R)
    return cs, float(np.arctan2(np.sqrt(max(float(vee @ vee), 1e-300)) / 2.0, cs))


def _half_turn_axis(R, cs: float) -> np.ndarray:
    """Axis from the symmetric part (R + R^T) / 2 - cos I = (1 - cos) n n^T: the column with the largest usable diagonal
    entry, signed by the matching component of vee(R - R^T) (>= 0 counts as +) — utils/so3.py:31-67."""
    sym = 0.5 * (R + R.T) - cs * np.eye(3)
    vee = _vee(R)
    k = 2 if sym[2, 2] >= 1e-6 else (1 if sym[1, 1] >= 1e-6 else 0)
    col = sym[:, k]
    axis = col / np.sqrt(max(float(col @ col), 1e-24))
    return axis if vee[k] >= 0 else -axis


def rotation_logm(R) -> Tuple[np.ndarray, float]:
    """(unit axis, angle) of a rotation matrix; the axis is zero below 1e-6 rad (utils/so3.py:70-93)."""
    R = np.asarray(R, dtype=np.float64)
    if R.shape != (3, 3):
        raise ValueError(f"rotation_logm requires a 3x3 rotation matrix, got shape {R.shape}. Matrix:\\n{R}")
    cs, theta = _cos_and_angle(R)
    if theta < 1e-6:
        return np.zeros(3), theta
    if theta > np.pi - 1e-2:
        return _half_turn_axis(R, cs), theta
    return _vee(R) / max(2.0 * np.sin(theta), 1e-12), theta


def MatrixLog3(R) -> np.ndarray:
    """so(3) logarithm as a skew matrix (utils/so3.py:172-191)."""
    R = np.asarray(R, dtype=np.float64)
    cs, theta = _cos_and_angle(R)
    if theta > np.pi - 1e-2:
        return skew_symmetric(theta * _half_turn_axis(R, cs))
    u = 1.0 - cs
    if cs > 1.0 - 5e-5:
        coef = 1.0 + u / 3.0 + u * u * (4.0 / 45.0)
    else:
        c = float(np.clip(cs, -1.0 + 1e-7, 1.0 - 1e-7))
        coef = np.arccos(c) / np.sqrt(max(1.0 - c * c, 1e-30))
    return 0.5 * coef * (R - R.T)


def _exp3_coefficients(t2: float) -> Tuple[float, float]:
    if t2 < 1e-4:
        return 1.0 - t2 / 6.0 + t2 * t2 / 120.0, 0.5 - t2 / 24.0 + t2 * t2 / 720.0
    t = np.sqrt(max(t2, 1e-12))
    return np.sin(t) / t, (1.0 - np.cos(t)) / (t * t)


def MatrixExp3(so3mat) -> np.ndarray:
    """Rodrigues formula (utils/so3.py:199-237)."""
    K = np.asarray(so3mat, dtype=np.float64)
    w = skew_symmetric_to_vector(K)
    a, b = _exp3_coefficients(float(w @ w))
    return np.eye(3) + a * K + b * (K @ K)


def rotation_matrix_to_euler_angles(R) -> np.ndarray:
    """ZYX roll, pitch, yaw in radians (utils/so3.py:240-251)."""
    R = np.asarray(R, dtype=np.float64)
    assert R.shape == (3, 3), f"Expected 3x3 rotation matrix, got shape {R.shape}"
    sy = np.sqrt(R[0, 0] ** 2 + R[1, 0] ** 2)
    if sy < 1e-6:
        return np.array([np.arctan2(-R[1, 2], R[1, 1]), np.arctan2(-R[2, 0], sy), sy * 0])
    return np.array([np.arctan2(R[2, 1], R[2, 2]), np.arctan2(-R[2, 0], sy), np.arctan2(R[1, 0], R[0, 0])])


def euler_to_rotation_matrix(euler_deg) -> np.ndarray:
    """Rz(yaw) Ry(pitch) Rx(roll) from DEGREE-valued angles (utils/so3.py:254-270)."""
    roll, pitch, yaw = np.asarray(euler_deg, dtype=np.float64) * (np.pi / 180.0)
    cz, sz, cy, sy, cx, sx = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Rz = np.array([[cz, -sz, 0.0], [sz, cz, 0.0], [0.0, 0.0, 1.0]])
    Ry = np.array([[cy, 0.0, sy], [0.0, 1.0, 0.0], [-sy, 0.0, cy]])
    Rx = np.array([[1.0, 0.0, 0.0], [0.0, cx, -sx], [0.0, sx, cx]])
    return Rz @ Ry @ Rx


# ------------------------------------------------------------------------------------------------ se(3)
def _homogeneous(R, p, last: float = 1.0) -> np.ndarray:
    T = np.zeros((4, 4))
    T[:3, :3], T[:3, 3], T[3, 3] = R, p, last
    return T


def transform_from_twist(S, theta) -> np.ndarray:
    """exp([S] theta) for a screw with unit (or zero) angular part (utils/se3.py:33-42)."""
    S = np.asarray(S, dtype=np.float64)
    W = skew_symmetric(S[:3])
    W2 = W @ W
    R = np.eye(3) + np.sin(theta) * W + (1 - np.cos(theta)) * W2
    G = np.eye(3) * theta + (1 - np.cos(theta)) * W + (theta - np.sin(theta)) * W2
    return _homogeneous(R, G @ S[3:])


def adjoint_transform(T) -> np.ndarray:
    """[[R, 0], [[p]R, R]] for twists ordered [w; v] (utils/se3.py:45-52)."""
    T = np.asarray(T, dtype=np.float64)
    R, p = T[:3, :3], T[:3, 3]
    A = np.zeros((6, 6))
    A[:3, :3], A[3:, 3:], A[3:, :3] = R, R, skew_symmetric(p) @ R
    return A


def _theta_g_inverse(phi) -> np.ndarray:
    """theta G^-1 = I - Phi / 2 + a(theta^2) Phi^2, a = (1 - (theta / 2) cot(theta / 2)) / theta^2 (utils/se3.py:55-113)."""
    t2 = float(np.sum(phi * phi)) / 2.0
    if t2 < 1e-2:
        a = 1.0 / 12 + t2 / 720 + t2 ** 2 / 30240 + t2 ** 3 / 1209600
    else:
        t = np.sqrt(max(t2, 5e-3))
        a = (1.0 - t * np.sin(t) / max(2.0 * (1.0 - np.cos(t)), 1e-300)) / (t * t)
    return np.eye(3) - 0.5 * phi + a * (phi @ phi)


def logm(T) -> np.ndarray:
    """Six-vector logarithm [rotation vector; theta G^-1 p] (utils/se3.py:116-124)."""
    T = np.asarray(T, dtype=np.float64)
    phi = MatrixLog3(T[:3, :3])
    return np.concatenate((skew_symmetric_to_vector(phi), _theta_g_inverse(phi) @ T[:3, 3]))


def se3ToVec(se3_matrix) -> np.ndarray:
    m = np.asarray(se3_matrix)
    if m.shape != (4, 4):
        raise ValueError("Input matrix must be a 4x4 matrix.")
    return np.concatenate((np.array([m[2, 1], m[0, 2], m[1, 0]]), m[:3, 3]))


def TransToRp(T):
    return T[:3, :3], T[:3, 3]


def TransInv(T) -> np.ndarray:
    T = np.asarray(T, dtype=np.float64)
    Rt = T[:3, :3].T
    return _homogeneous(Rt, -(Rt @ T[:3, 3]))


def MatrixLog6(T) -> np.ndarray:
    """se(3) matrix logarithm (utils/se3.py:158-166)."""
    T = np.asarray(T, dtype=np.float64)
    phi = MatrixLog3(T[:3, :3])
    return _homogeneous(phi, _theta_g_inverse(phi) @ T[:3, 3], last=0.0)


def MatrixExp6(se3mat) -> np.ndarray:
    """SE(3) exponential of an se(3) matrix (utils/se3.py:169-214)."""
    m = np.asarray(se3mat, dtype=np.float64)
    if m.shape != (4, 4):
        raise ValueError("Input matrix must be of shape (4, 4)")
    K, v = m[:3, :3], m[:3, 3]
    w = skew_symmetric_to_vector(K)
    t2 = float(w @ w)
    if t2 < 1e-4:
        c1, c2 = 0.5 - t2 / 24.0 + t2 * t2 / 720.0, 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0
    else:
        t = np.sqrt(max(t2, 1e-12))
        c1, c2 = (1 - np.cos(t)) / (t * t), (t - np.sin(t)) / (t * t * t)
    kv = K @ v
    return _homogeneous(MatrixExp3(K), v + c1 * kv + c2 * (K @ kv))


def VecTose3(V) -> np.ndarray:
    V = np.asarray(V, dtype=np.float64)
    return _homogeneous(skew_symmetric(V[:3]), V[3:], last=0.0)


# ------------------------------------------------------------------------------------------------ screws
def extract_r_list(Slist) -> np.ndarray:
    """A point on each screw axis, -w x v / |w|^2 (zero for prismatic columns), as (n, 3) (utils/screw.py:18-28)."""
    if Slist is None:
        return np.array([])
    S = np.asarray(Slist, dtype=np.float64).T
    w, v = S[:, :3], S[:, 3:]
    n2 = np.sum(w * w, axis=1)
    r = -np.cross(w, v) / np.where(n2 != 0, n2, 1.0)[:, None]
    return np.where((n2 != 0)[:, None], r, 0.0)


def extract_omega_list(Slist) -> np.ndarray:
    """The first three entries of each ROW of the input, as the reference slices it (utils/screw.py:31-33)."""
    return np.asarray(Slist)[:, :3]


def extract_screw_list(omega_list, r_list) -> Optional[np.ndarray]:
    """(6, n) screws [w; -w x r] from axes and points, with the reference's reshaping / broadcasting rules
    (utils/screw.py:54-93)."""
    if omega_list is None or r_list is None:
        return None
    w, r = np.asarray(omega_list, dtype=np.float64), np.asarray(r_list, dtype=np.float64)
    if r.size == 0:
        r = np.zeros((3, w.shape[1] if w.ndim == 2 else w.shape[0] // 3))
    elif r.ndim == 1:
        if r.size % 3:
            raise ValueError(f"Cannot reshape r_list of size {r.size} into (3, n) format")
        r = r.reshape(3, r.size // 3)
    if w.ndim == 1:
        if w.size % 3:
            raise ValueError(f"Cannot reshape omega_list of size {w.size} into (3, n) format")
        w = w.reshape(3, w.size // 3)
    if w.shape[0] != 3 or r.shape[0] != 3:
        raise ValueError("omega_list and r_list must each have 3 rows.")
    if w.shape[1] != r.shape[1]:
        if r.shape[1] == 1 and w.shape[1] > 1:
            r = np.repeat(r, w.shape[1], axis=1)
        elif w.shape[1] == 1 and r.shape[1] > 1:
            w = np.repeat(w, r.shape[1], axis=1)
        else:
            raise ValueError(f"omega_list and r_list must have the same number of columns. Got {w.shape[1]} and {r.shape[1]}.")
    return np.concatenate((w, np.cross(-w.T, r.T).T), axis=0)


def logm_to_twist(logm_matrix) -> np.ndarray:
    m = np.asarray(logm_matrix)
    if m.shape != (4, 4):
        raise ValueError("logm must be a 4x4 matrix.")
    return np.concatenate((skew_symmetric_to_vector(m[:3, :3]), m[:3, 3]))


# ------------------------------------------------------------------------------------------------ time scalings
def CubicTimeScaling(Tf: float, t: float) -> float:
    return 3 * (t / Tf) ** 2 - 2 * (t / Tf) ** 3


def QuinticTimeScaling(Tf: float, t: float) -> float:
    return 10 * (t / Tf) ** 3 - 15 * (t / Tf) ** 4 + 6 * (t / Tf) ** 5
