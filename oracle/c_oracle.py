"""ctypes wrapper of oracle/_build/liboracle.so (C restatement of the reference algorithm).
TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")
_dp = ctypes.POINTER(ctypes.c_double)


def build() -> str:
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return _LIB


def _lib():
    if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "oracle.c")):
        build()
    return ctypes.CDLL(_LIB)


def _p(a):
    return a.ctypes.data_as(_dp)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def inverse_dynamics_rows(tab, q, qd, qdd, g=None, Ftip=None, nthreads=0):
    """(tau (rows, n) float64, threads used).  `tab` is an oracle.ref_numpy.RobotTables."""
    lib = _lib()
    q, qd, qdd = _c(q), _c(qd), _c(qdd)
    rows, n = q.shape
    g = _c([0.0, 0.0, -9.81] if g is None else g)
    F = _c(np.zeros(6) if Ftip is None else Ftip)
    S, Mc, G, Me = _c(tab.S), _c(tab.Mcom), _c(tab.G), _c(tab.M_ee)
    tau = np.zeros((rows, n))
    used = lib.oracle_inverse_dynamics_rows(n, _p(S), _p(Mc), _p(G), _p(Me), _p(q), _p(qd), _p(qdd), _p(g), _p(F),
                                            ctypes.c_long(rows), _p(tau), int(nthreads))
    if used < 0:
        raise ValueError("dof outside 1..8")
    return tau, int(used)


def mass_matrix_rows(tab, q):
    lib = _lib()
    q = _c(q)
    rows, n = q.shape
    S, Mc, G, Me = _c(tab.S), _c(tab.Mcom), _c(tab.G), _c(tab.M_ee)
    M = np.zeros((rows, n, n))
    if lib.oracle_mass_matrix_rows(n, _p(S), _p(Mc), _p(G), _p(Me), _p(q), ctypes.c_long(rows), _p(M)) < 0:
        raise ValueError("dof outside 1..8")
    return M


def forward_dynamics_rows(tab, q, qd, tau, g=None, Ftip=None):
    lib = _lib()
    q, qd, tau = _c(q), _c(qd), _c(tau)
    rows, n = q.shape
    g = _c([0.0, 0.0, -9.81] if g is None else g)
    F = _c(np.zeros(6) if Ftip is None else Ftip)
    S, Mc, G, Me = _c(tab.S), _c(tab.Mcom), _c(tab.G), _c(tab.M_ee)
    out = np.zeros((rows, n))
    rc = lib.oracle_forward_dynamics_rows(n, _p(S), _p(Mc), _p(G), _p(Me), _p(q), _p(qd), _p(tau), _p(g), _p(F),
                                          ctypes.c_long(rows), _p(out))
    if rc < 0:
        raise ValueError(f"oracle_forward_dynamics_rows failed ({rc})")
    return out


def fd_trajectory(tab, theta0, dtheta0, taumat, g=None, Ftipmat=None, dt=0.01, intRes=1, joint_limits=None, state_f32=False,
                  nthreads=0):
    """B roll-outs of planning/trajectory_dynamics.py:580-708: theta0 / dtheta0 (B, n), taumat (B, Nt, n), Ftipmat
    (B, Nt, 6) or None -> (pos, vel, acc) float32 (B, Nt, n) and the threads used.  See oracle.c for `state_f32`."""
    lib = _lib()
    theta0, dtheta0, taumat = _c(theta0), _c(dtheta0), _c(taumat)
    B, Nt, n = taumat.shape
    assert theta0.shape == (B, n) and dtheta0.shape == (B, n)
    g = _c([0.0, 0.0, -9.81] if g is None else g)
    lim = _c(np.asarray(tab.joint_limits if joint_limits is None else joint_limits, dtype=np.float32))
    S, Mc, G, Me = _c(tab.S), _c(tab.Mcom), _c(tab.G), _c(tab.M_ee)
    Fm = None if Ftipmat is None else _c(Ftipmat)
    if Fm is not None:
        assert Fm.shape == (B, Nt, 6)
    out = [np.zeros((B, Nt, n), dtype=np.float32) for _ in range(3)]
    fp = ctypes.POINTER(ctypes.c_float)
    used = lib.oracle_fd_trajectory(n, _p(S), _p(Mc), _p(G), _p(Me), _p(lim), _p(theta0), _p(dtheta0), _p(taumat), _p(g),
                                    None if Fm is None else _p(Fm), ctypes.c_long(B), ctypes.c_long(Nt), ctypes.c_double(dt),
                                    int(intRes), int(bool(state_f32)), out[0].ctypes.data_as(fp), out[1].ctypes.data_as(fp),
                                    out[2].ctypes.data_as(fp), int(nthreads))
    if used < 0:
        raise ValueError(f"oracle_fd_trajectory failed ({used})")
    return out[0], out[1], out[2], int(used)
