"""Randomised serial chains: the model compiler (modified-DH frame assignment) and the device math
(host instantiation) against the NumPy oracle on robots nobody hand-picked — skew, intersecting, parallel
and coincident consecutive axes, prismatic joints anywhere in the chain, arbitrary CoM frames, 1..8 DOF."""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT
from oracle import ref_numpy as ref
from test_host_logic import hostsim  # noqa: F401  (fixture: g++/clang++ build of the device templates)


def _rand_rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def random_robot(rng, n, flavour):
    """Space screws at the home pose + CoM frames + block-diagonal inertias, like URDF extraction produces."""
    S = np.zeros((6, n))
    Mcom = np.zeros((n, 4, 4))
    G = np.zeros((n, 6, 6))
    prev_w, prev_p = None, None
    for i in range(n):
        kind = flavour[i % len(flavour)]
        if kind in ("parallel", "coincident") and prev_w is not None:
            w = prev_w.copy()
            p = prev_p + (0 if kind == "coincident" else 1) * rng.uniform(-0.4, 0.4, 3)
            if kind == "coincident":
                p = prev_p + rng.uniform(-0.3, 0.3) * w
        elif kind == "intersect" and prev_w is not None:
            w = _rand_rot(rng)[:, 0]
            p = prev_p + rng.uniform(-0.3, 0.3) * prev_w  # a point of the previous axis
        elif kind == "axis":  # axis-aligned, like most industrial arms
            w = np.eye(3)[rng.integers(3)] * rng.choice([-1.0, 1.0])
            p = rng.uniform(-0.5, 0.5, 3).round(2)
        else:
            w = _rand_rot(rng)[:, 2]
            p = rng.uniform(-0.5, 0.5, 3)
        if kind == "prismatic":
            S[3:, i] = w
        else:
            S[:3, i] = w
            S[3:, i] = -np.cross(w, p)
            prev_w, prev_p = w, p
        Mcom[i] = np.eye(4)
        Mcom[i][:3, :3] = _rand_rot(rng)
        Mcom[i][:3, 3] = p + rng.uniform(-0.2, 0.2, 3)
        A = rng.normal(size=(3, 3))
        I = A @ A.T * 0.05 + np.eye(3) * 0.01
        m = rng.uniform(0.2, 5.0)
        G[i, :3, :3] = I
        G[i, 3:, 3:] = m * np.eye(3)
    M_ee = np.eye(4)
    M_ee[:3, :3] = _rand_rot(rng)
    M_ee[:3, 3] = rng.uniform(-1, 1, 3)
    lim = np.tile([-2.5, 2.5], (n, 1))
    return ref.RobotTables(S=S, M_ee=M_ee, G=G, Mcom=Mcom, joint_limits=lim)


FLAVOURS = [("general",), ("axis",), ("general", "parallel"), ("general", "intersect"), ("general", "coincident", "general"),
            ("general", "prismatic"), ("prismatic", "general", "parallel"), ("axis", "parallel", "parallel", "axis"),
            ("prismatic", "prismatic", "general")]


@pytest.mark.parametrize("seed", range(18))
def test_random_chain_matches_oracle(seed, hostsim):  # noqa: F811
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 9))
    tab = random_robot(rng, n, FLAVOURS[seed % len(FLAVOURS)])
    rows = 3
    q = rng.uniform(-2.5, 2.5, (rows, n))
    q[:, np.abs(tab.S[:3]).sum(axis=0) == 0] *= 0.1  # prismatic joints: decimetres, not radians
    qd = rng.uniform(-1, 1, (rows, n))
    qdd = rng.uniform(-2, 2, (rows, n))
    g = np.array([0.4, -0.3, -9.81])
    F = rng.uniform(-3, 3, 6)
    tau, T, J = hostsim(tab, q, qd, qdd, g, F, 0)
    for r in range(rows):
        np.testing.assert_allclose(T[r], ref.fk_space(tab, q[r]), atol=1e-10)
        np.testing.assert_allclose(J[r], ref.jacobian_space(tab, q[r]), atol=1e-10)
        want = ref.inverse_dynamics(tab, q[r], qd[r], qdd[r], g, F)
        np.testing.assert_allclose(tau[r], want, rtol=1e-6, atol=1e-6 * max(1.0, np.abs(want).max()))
    # mass matrix / forward dynamics through the same frames
    for mode in (0, 3):  # unit-acceleration recursions and the composite-rigid-body algorithm
        M = hostsim.fd(tab, mode, rows, q, None, None, g, np.zeros(6), outshape=(rows, n, n))
        for r in range(rows):
            np.testing.assert_allclose(M[r], ref.mass_matrix(tab, q[r]), rtol=1e-8, atol=1e-9)
    # float32, packed: loose tolerance (random robots are not well conditioned like real arms)
    t32, _, _ = hostsim(tab, q, qd, qdd, g, F, 2)
    for r in range(rows):
        want = ref.inverse_dynamics(tab, q[r], qd[r], qdd[r], g, F)
        assert np.abs(t32[r] - want).max() <= 2e-4 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("seed", range(12))
def test_random_chain_forward_dynamics_and_ik(seed, hostsim):  # noqa: F811
    """Forward dynamics (bias recursion + CRBA + Cholesky) and the IK iteration on the same randomised chains: FD against
    the oracle's solve(M, tau - c - g - J^T F); IK (plain, and with adaptive tuning + backtracking) against the oracle's
    restatement of the reference loop on reachable targets — same flag, same count (+-1), same solution."""
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.integers(2, 9))
    tab = random_robot(rng, n, FLAVOURS[seed % len(FLAVOURS)])
    prismatic = np.abs(tab.S[:3]).sum(axis=0) == 0
    rows = 2
    q = rng.uniform(-2.0, 2.0, (rows, n))
    q[:, prismatic] *= 0.1
    qd, tau = rng.uniform(-1, 1, (rows, n)), rng.uniform(-5, 5, (rows, n))
    g, F = np.array([0.4, -0.3, -9.81]), rng.uniform(-3, 3, 6)
    for r in range(rows):
        qdd = hostsim.fd(tab, 1, 1, q[r:r + 1], qd[r:r + 1], tau[r:r + 1], g, F, outshape=(1, n))
        want = ref.forward_dynamics(tab, q[r], qd[r], tau[r], g, F)
        np.testing.assert_allclose(qdd[0], want, rtol=2e-5, atol=2e-5 * max(1.0, np.abs(want).max()))  # the oracle's own FD noise x M^-1
    lim = np.tile([-2.5, 2.5], (n, 1)).astype(float)
    lim[prismatic] = [-0.4, 0.4]
    for case in range(3):
        q_true = rng.uniform(0.7 * lim[:, 0], 0.7 * lim[:, 1])
        T = ref.fk_space(tab, q_true)
        q0 = np.clip(q_true + rng.uniform(-0.2, 0.2, n) * np.where(prismatic, 0.2, 1.0), lim[:, 0], lim[:, 1])
        for adaptive, backtrack in ((False, False), (True, True)):
            params = [1e-6, 1e-6, 150, 2e-2, 0.3, 1.0, 1.0, float(adaptive), float(backtrack)]
            th, ok, it, rs = hostsim.ik(tab, lim, params, T[None], q0[None])
            o_th, o_ok, o_it, o_rs = ref.iterative_inverse_kinematics(tab, T, q0, 1e-6, 1e-6, 150, 2e-2, 0.3, joint_limits=lim,
                                                                      rng=np.random.RandomState(0), adaptive_tuning=adaptive,
                                                                      backtracking=backtrack)
            if o_rs or rs[0]:
                continue  # restart noise differs by design
            assert bool(ok[0]) == o_ok and abs(int(it[0]) - o_it) <= 1, (seed, case, adaptive, ok[0], it[0], o_ok, o_it)
            np.testing.assert_allclose(th[0], o_th, rtol=0, atol=1e-6 if o_ok else 1e-4)

