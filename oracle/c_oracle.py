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
