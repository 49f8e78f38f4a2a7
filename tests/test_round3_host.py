"""CPU-only tests of what round 3 added to the mirrored classes: the legacy (Mlist_per_link=None) dynamics object,
truncated-theta and body-frame kinematics of models whose B_list is not Ad(M^-1) S_list, and the inverse-kinematics CPU
launcher - each against outputs of the reference itself (tests/golden/make_golden.py legacy / ik)."""
import warnings
from math import pi

import numpy as np
import pytest

from conftest import ROBOTS, golden_path
from oracle import ref_numpy as ref

import manipulapy_amd as mp


def _legacy_objects():
    z = np.load(golden_path("legacy_dynamics.npz"))
    out = {}
    for tag in ("A", "B"):
        dyn = mp.ManipulatorDynamics(M_list=z[f"{tag}_M"], omega_list=None, r_list=None, b_list=None, S_list=z[f"{tag}_S"],
                                     B_list=z[f"{tag}_B"], Glist=z[f"{tag}_G"])
        out[tag] = dyn
    return z, out


def test_truncated_theta_and_body_frame_kinematics_match_the_reference():
    """forward_kinematics(theta[:k]) = prod_{j<k} exp([S_j] theta_j) . M and jacobian(theta[:k]) (6, k), space and body frame
    (reference kinematics/fk.py:59-80, jacobian.py:62-91) on a hand-built model whose B_list is NOT Ad(M^-1) S_list (A) and on
    the UR5 tables (B)."""
    z, objs = _legacy_objects()
    for tag, dyn in objs.items():
        n = dyn.S_list.shape[1]
        th = z[f"{tag}_theta"][0]
        for k in range(0, n + 1):
            for frame in ("space", "body"):
                np.testing.assert_allclose(dyn.forward_kinematics(th[:k], frame), z[f"{tag}_fk_{frame}_{k}"], rtol=1e-12, atol=1e-13)
                if frame == "space" or k >= 1:
                    J = dyn.jacobian(th[:k], frame)
                    assert J.shape == (6, k)
                    np.testing.assert_allclose(J, z[f"{tag}_jac_{frame}_{k}"], rtol=1e-12, atol=1e-13)
        with pytest.raises(IndexError):
            dyn.jacobian(th[:0], "body")
        with pytest.raises(IndexError):
            dyn.forward_kinematics(np.zeros(n + 1))
        # a (rows, k) batch of truncated vectors
        T = dyn.forward_kinematics(z[f"{tag}_theta"][:, :3])
        assert T.shape == (5, 4, 4)
        np.testing.assert_allclose(T[0], z[f"{tag}_fk_space_3"], rtol=1e-12, atol=1e-13)
    # the full-length space-frame call still goes through the registered operation and agrees with the product of exponentials
    sm, _, _ = mp.load_robot("ur5")
    q = np.linspace(-1, 1, 6)
    np.testing.assert_allclose(sm.forward_kinematics(q), sm._fk_poe(q, "space"), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(sm.jacobian(q, "body"), sm._jacobian_poe(q, "body"), rtol=1e-9, atol=1e-11)


def test_legacy_dynamics_object_reproduces_the_reference_with_its_warning():
    """ManipulatorDynamics(..., Mlist_per_link=None) - what URDF.to_manipulator_dynamics() returns (reference
    urdf/core.py:795-817) - evaluates the reference's legacy approximation (dynamics/mass_matrix.py:101-132,
    forces.py:136-154) and warns as the reference does; M, c, g, inverse and forward dynamics at 5 configurations."""
    z, objs = _legacy_objects()
    for tag, dyn in objs.items():
        th, dth, ddth, F = (z[f"{tag}_{k}"] for k in ("theta", "dtheta", "ddtheta", "ftip"))
        g = np.array([0.0, 0.0, -9.81])
        with pytest.warns(UserWarning, match="without Mlist_per_link"):
            dyn.mass_matrix(th[0])
        with pytest.warns(UserWarning, match="legacy approximation"):
            dyn.gravity_forces(th[0], g)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for k in range(len(th)):
                M = dyn.mass_matrix(th[k])
                np.testing.assert_allclose(M, z[f"{tag}_mass"][k], rtol=1e-10, atol=1e-12)
                assert np.array_equal(M, M.T)
                scale = np.abs(z[f"{tag}_c"][k]).max() + 1e-6
                np.testing.assert_allclose(dyn.velocity_quadratic_forces(th[k], dth[k]), z[f"{tag}_c"][k], rtol=0, atol=1e-6 * scale + 1e-8)
                np.testing.assert_allclose(dyn.gravity_forces(th[k], g), z[f"{tag}_g"][k], rtol=1e-10, atol=1e-12)
                tau = dyn.inverse_dynamics(th[k], dth[k], ddth[k], g, F[k])
                np.testing.assert_allclose(tau, z[f"{tag}_id"][k], rtol=1e-7, atol=1e-7)
                assert tau.dtype == np.float64 and tau.shape == (len(th[k]),)
                qdd = dyn.forward_dynamics(th[k], dth[k], z[f"{tag}_id"][k], g, F[k])
                np.testing.assert_allclose(qdd, z[f"{tag}_fd"][k], rtol=1e-5, atol=1e-6)
        with pytest.raises(NotImplementedError):
            dyn.hip_model()


def test_planner_on_a_legacy_dynamics_object_matches_the_reference_cpu_loops():
    """inverse_dynamics_trajectory / forward_dynamics_trajectory of a planner whose dynamics object is the legacy one: the
    reference's per-row host loops (planning/trajectory_dynamics.py:308-380, :580-708), float32 rows, under the NumPy
    backend."""
    z, objs = _legacy_objects()
    for tag, dyn in objs.items():
        n = dyn.S_list.shape[1]
        pl = mp.OptimizedTrajectoryPlanning(dyn, "nonexistent.urdf", dyn, [(-pi, pi)] * n, use_cuda=False)
        th = z[f"{tag}_theta"]
        traj = pl.joint_trajectory(th[0], th[1], 1.0, 8, 5)
        np.testing.assert_allclose(traj["positions"], z[f"{tag}_traj_pos"], rtol=3e-7, atol=1e-6)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tau = pl.inverse_dynamics_trajectory(z[f"{tag}_traj_pos"], z[f"{tag}_traj_vel"], z[f"{tag}_traj_acc"])
            assert tau.dtype == np.float32 and tau.shape == (8, n)
            np.testing.assert_allclose(tau, z[f"{tag}_traj_tau"], rtol=1e-5, atol=1e-5 * np.abs(z[f"{tag}_traj_tau"]).max())
            r = pl.forward_dynamics_trajectory(th[2], z[f"{tag}_dtheta"][2] * 0.1, z[f"{tag}_roll_tau"], np.array([0, 0, -9.81]),
                                               z[f"{tag}_roll_F"], 0.01, 2)
        for key, want in (("positions", "roll_pos"), ("velocities", "roll_vel"), ("accelerations", "roll_acc")):
            assert r[key].dtype == np.float32 and r[key].shape == (6, n)
            np.testing.assert_allclose(r[key], z[f"{tag}_{want}"], rtol=2e-4, atol=2e-5 * max(1.0, float(np.abs(z[f"{tag}_{want}"]).max())))
        with pytest.warns(UserWarning, match="legacy approximation"):
            pl.inverse_dynamics_trajectory(z[f"{tag}_traj_pos"][:2], z[f"{tag}_traj_vel"][:2], z[f"{tag}_traj_acc"][:2])


def _restarted(tab, z, robot, i):
    p = z[f"{robot}_params"][i]
    return ref.iterative_inverse_kinematics(tab, z[f"{robot}_T_desired"][i], z[f"{robot}_theta0"][i], p[0], p[1], int(p[2]), p[3], p[4],
                                            p[5], p[6], joint_limits=z[f"{robot}_joint_limits"], rng=np.random.RandomState(1234),
                                            adaptive_tuning=bool(p[7]), backtracking=bool(p[8]))[3] > 0


@pytest.mark.parametrize("robot", ROBOTS)
def test_inverse_kinematics_cpu_launcher_against_reference_runs(robot, tables):
    """The NumPy backend's launcher of "kinematics.inverse" (mp_inverse_kinematics_cpu_f64: the kernel's iteration on host
    threads) against the reference's own iterative_inverse_kinematics runs (tests/golden/ik.npz) - the test the HIP kernel
    passes, on the CPU - and batch == single."""
    z = np.load(golden_path("ik.npz"))
    tab = tables[robot]
    lim = z[f"{robot}_joint_limits"]
    sm, _, _ = mp.load_robot(robot)
    sm.joint_limits = [(None if not np.isfinite(lo) else float(lo), None if not np.isfinite(hi) else float(hi)) for lo, hi in lim]
    with mp.use_backend("numpy"):
        checked = 0
        for i in range(10):
            p = z[f"{robot}_params"][i]
            th, ok, it = sm.iterative_inverse_kinematics(z[f"{robot}_T_desired"][i], z[f"{robot}_theta0"][i], eomg=p[0], ev=p[1],
                                                         max_iterations=int(p[2]), damping=p[3], step_cap=p[4],
                                                         weight_orientation=p[5], weight_position=p[6],
                                                         adaptive_tuning=bool(p[7]), backtracking=bool(p[8]))
            want_ok, want_it = bool(z[f"{robot}_success"][i]), int(z[f"{robot}_iterations"][i])
            if _restarted(tab, z, robot, i):
                continue
            checked += 1
            assert ok == want_ok and abs(it - want_it) <= (1 if want_ok else 0), (robot, i, ok, it, want_it)
            np.testing.assert_allclose(th, z[f"{robot}_theta"][i], rtol=0, atol=1e-6 if want_ok else 1e-5)
        assert checked >= 5
        rng = np.random.default_rng(31)
        B = 64
        fin = np.where(np.isfinite(lim), lim, np.array([-np.pi, np.pi]))
        q_true = rng.uniform(0.6 * fin[:, 0], 0.6 * fin[:, 1], (B, tab.n))
        T = np.stack([ref.fk_space(tab, q) for q in q_true])
        q0 = np.clip(q_true + rng.uniform(-0.3, 0.3, (B, tab.n)), fin[:, 0], fin[:, 1])
        th, ok, it = sm.batch_inverse_kinematics(T, q0, max_iterations=300)
        assert ok.mean() > 0.5
        for b in np.flatnonzero(ok)[:10]:
            _, rot, tr = ref.ik_geometric_error(ref.fk_space(tab, th[b]), T[b])
            assert rot < 1e-6 and tr < 1e-6
        one = sm.iterative_inverse_kinematics(T[3], q0[3], max_iterations=300)
        np.testing.assert_array_equal(one[0], th[3])
        assert one[1] == ok[3] and one[2] == it[3]


# ----------------------------------------------------------------------------- more than 8 joints (csrc/mp_dyn.h)
def _jaco(name):
    import os

    z = np.load(golden_path("urdf_suite.npz"))
    proc = mp.URDFToSerialManipulator(golden_path(os.path.join("urdf_suite", f"{name}.urdf")), tip_link=str(z[f"{name}__ee"]))
    return z, proc


@pytest.mark.parametrize("name", ["jaco_6dof", "jaco_7dof"])
def test_robots_with_more_than_eight_joints_compute_on_the_cpu_launchers(name):
    """The reference's Jaco arms with their three-finger hands (9 / 10 actuated joints, ManipulaPy_data/__init__.py:174-189):
    forward kinematics, Jacobian, mass matrix, velocity / gravity forces, inverse and forward dynamics, and the planner's
    inverse-dynamics trajectory and roll-out through the run-time-n rows of csrc/mp_dyn.h (NumPy backend: *_cpu launchers),
    against the reference's own values for the same URDF (tests/golden/urdf_suite.npz)."""
    z, proc = _jaco(name)
    sm, dyn = proc.serial_manipulator, proc.dynamics
    n = proc.robot_data["actuated_joints_num"]
    assert n in (9, 10)
    th, dth, ddth = z[f"{name}__theta"], z[f"{name}__dtheta"], z[f"{name}__ddtheta"]
    g, F = np.array([0.0, 0.0, -9.81]), np.array([1.0, -2.0, 0.5, 3.0, -1.5, 0.75])
    np.testing.assert_allclose(sm.forward_kinematics(th), z[f"{name}__T"], atol=1e-10)
    np.testing.assert_allclose(sm.jacobian(th), z[f"{name}__J"], atol=1e-10)
    M = dyn.mass_matrix(th)
    np.testing.assert_allclose(M, z[f"{name}__mass"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(dyn.gravity_forces(th, g), z[f"{name}__g"], rtol=1e-9, atol=1e-10)
    c = dyn.velocity_quadratic_forces(th, dth)
    np.testing.assert_allclose(c, z[f"{name}__c"], rtol=0, atol=1e-7 * max(1.0, np.abs(z[f"{name}__c"]).max()))  # the reference's is a finite difference
    tau = dyn.inverse_dynamics(th, dth, ddth, g, F)
    np.testing.assert_allclose(tau, z[f"{name}__tau"], rtol=1e-6, atol=1e-6)
    qdd = dyn.forward_dynamics(th, dth, z[f"{name}__tau"], g, F)
    np.testing.assert_allclose(qdd, z[f"{name}__qdd"], rtol=1e-5, atol=1e-5 * max(1.0, np.abs(z[f"{name}__qdd"]).max()))
    np.testing.assert_allclose(qdd, ddth, rtol=1e-5, atol=1e-5)   # FD(ID(qdd)) == qdd
    lim = np.array([[-10.0, 10.0]] * n)
    pl = mp.OptimizedTrajectoryPlanning(sm, None, dyn, lim, use_cuda=False)
    rows = 70
    q = np.tile(th, (rows, 1)); qd = np.tile(dth, (rows, 1)); q2 = np.tile(ddth, (rows, 1))
    t64 = pl.inverse_dynamics_trajectory(q, qd, q2, g, F)
    assert t64.dtype == np.float32 and t64.shape == (rows, n)
    np.testing.assert_allclose(t64, np.tile(z[f"{name}__tau"], (rows, 1)), rtol=2e-6, atol=2e-6 * np.abs(z[f"{name}__tau"]).max())
    t32 = pl.inverse_dynamics_trajectory(q.astype(np.float32), qd.astype(np.float32), q2.astype(np.float32), g, F)
    np.testing.assert_allclose(t32, t64, rtol=1e-4, atol=5e-5 * np.abs(t64).max())
    # a short roll-out: row 1's acceleration is forward_dynamics of the initial state, as in the reference's loop
    taum = np.tile(z[f"{name}__tau"], (4, 1))
    r = pl.forward_dynamics_trajectory(th, dth, taum, g, np.tile(F, (4, 1)), 1e-4, 1)
    assert r["positions"].shape == (4, n) and r["accelerations"].dtype == np.float32
    np.testing.assert_allclose(r["accelerations"][1], z[f"{name}__qdd"], rtol=1e-4, atol=1e-4 * max(1.0, np.abs(z[f"{name}__qdd"]).max()))
    traj = pl.joint_trajectory(th, th + 0.1, 1.0, 16, 5)
    assert traj["positions"].shape == (16, n)
    # inverse kinematics on the run-time-n kinematics (MpIkLooped): from nearby guesses every target is met
    rng = np.random.default_rng(3)
    goal = mp.ik_helpers.clip_to_limits(th + rng.uniform(-0.2, 0.2, (24, n)), sm.joint_limits)
    Ts = np.stack([sm.forward_kinematics(x) for x in goal])
    sol, ok, it = sm.batch_inverse_kinematics(Ts, np.tile(th, (24, 1)), max_iterations=2000, adaptive_tuning=True, backtracking=True)
    assert ok.all() and (it < 2000).all(), (ok, it)
    for a, T in zip(sol, Ts):
        Tc = sm.forward_kinematics(a)
        assert np.linalg.norm(Tc[:3, 3] - T[:3, 3]) < 1e-6 and np.abs(Tc[:3, :3] - T[:3, :3]).max() < 1e-5
    one = sm.iterative_inverse_kinematics(Ts[3], th, max_iterations=2000, adaptive_tuning=True, backtracking=True)
    np.testing.assert_array_equal(one[0], sol[3])


@pytest.mark.parametrize("robot", ROBOTS)
def test_looped_rows_equal_the_unrolled_rows_on_the_benchmark_robots(robot, tables, dyn_golden, monkeypatch):
    """MANIPULAPY_HIP_LOOPED=1 builds a 6..8-joint model for the run-time-n path: its CPU launchers must reproduce the
    unrolled ones (same frames, same per-joint arithmetic) and the reference's goldens."""
    from manipulapy_amd import _hip

    tab, z = tables[robot], dyn_golden[robot]
    unrolled = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
    monkeypatch.setenv("MANIPULAPY_HIP_LOOPED", "1")
    looped = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
    monkeypatch.delenv("MANIPULAPY_HIP_LOOPED")
    assert looped.blob()["joints"].shape == (32, 18) and unrolled.blob()["joints"].shape == (8, 18)
    q, qd, qdd = z["thetas"], z["dthetas"], z["ddthetas"]
    g = z["g"]
    for dtype, tol in ((np.float64, 1e-12), (np.float32, 2e-5)):
        for F in (None, z["ftips"][0]):
            a = _hip.cpu_id_trajectory(unrolled, q, qd, qdd, g, F, dtype=dtype)
            b = _hip.cpu_id_trajectory(looped, q, qd, qdd, g, F, dtype=dtype)
            np.testing.assert_allclose(b, a, rtol=tol, atol=tol * np.abs(a).max())
    Ta, Ja, _ = _hip.cpu_fk_jac_id(unrolled, q)
    Tb, Jb, _ = _hip.cpu_fk_jac_id(looped, q)
    np.testing.assert_allclose(Tb, Ta, rtol=0, atol=1e-13); np.testing.assert_allclose(Jb, Ja, rtol=0, atol=1e-13)
    np.testing.assert_allclose(_hip.cpu_mass_matrix(looped, q), _hip.cpu_mass_matrix(unrolled, q), rtol=1e-12, atol=1e-13)
    tau = _hip.cpu_id_trajectory(unrolled, q, qd, qdd, g, z["ftips"][0], dtype=np.float64)
    fa = _hip.cpu_forward_dynamics(unrolled, q, qd, tau, g, z["ftips"][0])
    fb = _hip.cpu_forward_dynamics(looped, q, qd, tau, g, z["ftips"][0])
    np.testing.assert_allclose(fb, fa, rtol=1e-8, atol=1e-9 * max(1.0, np.abs(fa).max()))
    B, N = 5, 12
    rng = np.random.default_rng(5)
    tm = np.tile(tau[:B, None, :], (1, N, 1)) + rng.uniform(-1e-3, 1e-3, (B, N, tab.n))
    Fm = rng.uniform(-0.02, 0.02, (B, N, 6))
    for dtype in (np.float64, np.float32):
        ra = _hip.cpu_fd_trajectory(unrolled, q[:B], qd[:B] * 0.1, tm, g, Fm, 0.01, 2, dtype=dtype)
        rb = _hip.cpu_fd_trajectory(looped, q[:B], qd[:B] * 0.1, tm, g, Fm, 0.01, 2, dtype=dtype)
        for k in range(3):
            np.testing.assert_allclose(rb[k], ra[k], rtol=0, atol=(1e-6 if dtype == np.float64 else 2e-4) * max(1.0, float(np.abs(ra[k]).max())))
    # inverse kinematics: the same iteration template on both kinematics -> the same iterates (to rounding), the same counts
    q0 = q[4:10]   # (rows 0..3 are the zero / limit / near-zero configurations: singular starts)
    goal = np.clip(q0 + rng.uniform(-0.1, 0.1, (len(q0), tab.n)), tab.joint_limits[:, 0], tab.joint_limits[:, 1])
    Ts = _hip.cpu_fk_jac_id(unrolled, goal)[0]
    for opts in (dict(), dict(adaptive_tuning=True, backtracking=True)):
        ia = _hip.cpu_inverse_kinematics(unrolled, Ts, q0, tab.joint_limits, max_iterations=2000, **opts)
        ib = _hip.cpu_inverse_kinematics(looped, Ts, q0, tab.joint_limits, max_iterations=2000, **opts)
        assert ia[1].all() and ib[1].all()
        np.testing.assert_array_equal(ib[2], ia[2]); np.testing.assert_array_equal(ib[3], ia[3])
        np.testing.assert_allclose(ib[0], ia[0], rtol=0, atol=1e-9)
    with pytest.raises(_hip.HipError):
        looped.specialize_source()


def test_planner_rollout_layout_option_on_the_cpu_launcher(tables):
    """batch_forward_dynamics_trajectory(layout="time_major"): (N, B, *) host arrays in and out, the same numbers as the
    (B, N, *) call (NumPy backend: CPU launcher); a bad layout name raises."""
    sm, dyn, lim = mp.load_robot("xarm6")
    pl = mp.OptimizedTrajectoryPlanning(sm, None, dyn, lim, use_cuda=False)
    rng = np.random.default_rng(8)
    B, N, n = 5, 9, 6
    th0, dth0 = rng.uniform(-0.5, 0.5, (B, n)), rng.uniform(-0.2, 0.2, (B, n))
    tm, Fm = rng.uniform(-0.5, 0.5, (B, N, n)), rng.uniform(-0.02, 0.02, (B, N, 6))
    a = pl.batch_forward_dynamics_trajectory(th0, dth0, tm, None, Fm, 0.01, 2)
    sw = lambda x: np.ascontiguousarray(np.swapaxes(x, 0, 1))
    b = pl.batch_forward_dynamics_trajectory(th0, dth0, sw(tm), None, sw(Fm), 0.01, 2, layout="time_major")
    for k in ("positions", "velocities", "accelerations"):
        assert b[k].shape == (N, B, n)
        np.testing.assert_array_equal(sw(b[k]), a[k])
    with pytest.raises(ValueError):
        pl.batch_forward_dynamics_trajectory(th0, dth0, tm, None, Fm, 0.01, 2, layout="columns")


def test_bench_parity_rules_and_line_shape():
    """bench.py's asserted parity sample on synthetic data: inside the first bound -> ok; a row over the first bound but inside 4
    float32 input ulps -> ok after re-examination; over both -> fails; a NaN in the GPU result -> fails.  And the entry a
    configuration gets in the default line's "configs" object has the fields the driver-visible line promises."""
    import bench

    tab = ref.load_tables(golden_path("model_ur5.npz"))
    rng = np.random.default_rng(4)
    q, qd, qdd = (rng.uniform(-1, 1, (64, 6)).astype(np.float32) for _ in range(3))
    from oracle import c_oracle

    want = c_oracle.inverse_dynamics_rows(tab, q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64))[0]
    sens = bench.id_sensitivity(tab, q, qd, qdd)
    S = sens(np.arange(64))
    assert S.shape == (64, 6) and (S > 0).all() and S.max() < 1e-3          # one ulp of a |x| < 1 input moves tau by ~1e-6 N.m
    ok = bench.parity_rows(want + 1e-7, want, "f32", sens)
    assert ok["ok"] and ok["rows_over_first_bound"] == 0
    tol1 = 1e-4 * np.abs(want) + bench.F32_ROW * np.abs(want).max(axis=1, keepdims=True)
    bump = want.copy()
    bump[5, 2] += tol1[5, 2] + 2.0 * S[5, 2]                                 # over the bound by 2 input ulps: fails, the ulps are only reported
    mid = bench.parity_rows(bump, want, "f32", sens)
    assert not mid["ok"] and mid["rows_over_first_bound"] == 1 and mid["worst_over_first_bound"] > 1.0
    assert 1.5 < mid["worst_excess_over_input_ulps"] < 2.5
    assert not bench.parity_rows(bump, want, "f32")["ok"]
    bad = want.copy(); bad[0, 0] = np.nan
    assert not bench.parity_rows(bad, want, "f32", sens)["ok"]
    assert bench.parity_rows(want * (1 + 5e-7), want, "f64")["ok"] and not bench.parity_rows(want * (1 + 5e-6) + 1e-6, want, "f64")["ok"]
    # float64: the floor follows the oracle's finite-difference noise, 4e-9 per (rad/s)^2
    fast = np.full((64, 6), 5.0)                                              # |qd|^2 = 150 -> + 6e-7
    assert bench.parity_rows(want + 5e-7, want, "f64", qd=fast)["ok"] and not bench.parity_rows(want + 5e-7, want, "f64")["ok"]
    assert not bench.parity_rows(want + 5e-7, want, "f64", qd=fast * 0.1)["ok"]
    # round 6: how much of the float32 bound rides on the floor is IN the record - an element off by 3e-4 of a reference that is
    # 1e-3 of its row's largest torque fails 1e-4 |ref| alone and passes through 5e-6 max|row|
    small = want.copy()
    j, jm = int(np.argmin(np.abs(want[7]))), int(np.argmax(np.abs(want[7])))
    small[7, j] = want[7, jm] * 1e-3
    got = small.copy(); got[7, j] *= 1.0 + 3e-4
    fl = bench.parity_rows(got, small, "f32")
    assert fl["ok"] and fl["elements_over_pure_rel"]["count"] == 1 and fl["elements_over_pure_rel"]["rows"] == 1
    assert abs(fl["elements_over_pure_rel"]["max_ref_over_rowmax_of_those"] - 1e-3) < 1e-6
    assert abs(fl["elements_over_pure_rel"]["fraction"] - 1.0 / want.size) < 1e-12
    brief = bench.parity_brief(fl, rows_total=64, sets=1)
    assert brief["elements_over_pure_rel"]["count"] == 1 and brief["max_ref_over_rowmax_of_those"] > 0 and brief["rows_checked"] == 64
    assert bench.parity_rows(want + 1e-7, want, "f32")["elements_over_pure_rel"]["count"] >= 0
    # the gathered blocks' checksum depends on where the words sit (a plain sum passed a permuted block)
    words = rng.integers(0, 2**32, 5_000_003, dtype=np.uint64).astype(np.uint32)
    swapped = words.copy(); swapped[[11, 4_500_000]] = swapped[[4_500_000, 11]]
    assert int(words.sum(dtype=np.uint64)) == int(swapped.sum(dtype=np.uint64))
    assert bench.weighted_word_sum(words) != bench.weighted_word_sum(swapped) and bench.weighted_word_sum(words) == bench.weighted_word_sum(words.copy())
    assert bench.weighted_word_sum(np.zeros(0, np.uint32)) == 0
    # a single process has nothing to agree on; the flag passes through
    from manipulapy_amd import sharding
    one = sharding.HostGather(sharding.ShardInfo(0, 1, 0))
    assert bench.agree_hung(one, False, {}) is False and bench.agree_hung(one, True, {}) is True
    # the oracle's thread count is bench's own choice, not the launcher's: torch.distributed.run exports OMP_NUM_THREADS=1
    import os as _os
    saved = _os.environ.get("MANIPULAPY_BENCH_CPU_THREADS")
    _os.environ["MANIPULAPY_BENCH_CPU_THREADS"] = "3"
    try:
        assert bench.host_threads() == 3
        assert c_oracle.inverse_dynamics_rows(tab, q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64), nthreads=bench.host_threads())[1] == 3
    finally:
        if saved is None:
            _os.environ.pop("MANIPULAPY_BENCH_CPU_THREADS")
        else:
            _os.environ["MANIPULAPY_BENCH_CPU_THREADS"] = saved
    assert bench.host_threads() >= 1
    # a strong-scaled entry whose set-up fails on a rank is an entry with "error" (agreed over gloo, so that no rank waits in the
    # barriers that follow), not an exception
    import argparse

    class _NoDevice:
        def specialize(self, model):
            raise RuntimeError("no device in this test")

    entry, hung = bench.bench_strong("c4", argparse.Namespace(no_specialize=False), sharding.ShardInfo(0, 1, 0), one, _NoDevice(), {})
    assert hung is False and "no device in this test" in entry["error"] and entry["scaling"] == "strong"
    # the cold window's three figures and the clock pass on a stand-in context: event times are scripted, the arithmetic is bench's
    class _Ev:
        def __init__(self, ctx): self.ctx, self.t = ctx, None
        def record(self): self.t = self.ctx.now
        def elapsed_ms_since(self, other): return self.t - other.t
        def destroy(self): pass

    class _FakeCtx:
        """a launch costs 0.07 ms, the first one after a sleep 0.05 ms more (the wake-up)"""
        def __init__(self): self.now, self.cold, self.sampling = 0.0, True, False
        def event(self): return _Ev(self)
        def synchronize(self): pass
        def launch(self):
            self.now += 0.07 + (0.05 if self.cold else 0.0)
            self.cold = False
        def clock_sample_begin(self, ms): assert ms > 0; self.sampling = True
        def clock_sample_end(self): assert self.sampling; self.sampling = False; return 1.7e9, 0.3

    fake = _FakeCtx()
    real_sleep = bench.time.sleep
    bench.time.sleep = lambda s_: setattr(fake, "cold", True)     # (no half seconds of waiting in the suite)
    try:
        cold = bench.cold_figures(fake, fake.launch, 5)
    finally:
        bench.time.sleep = real_sleep
    assert abs(cold["first_launch_ms"] - 0.12) < 1e-12 and abs(cold["launch_period_ms_after_first"] - 0.07) < 1e-12
    assert abs(cold["kernel_ms_cold"] - (0.12 + 4 * 0.07) / 5) < 1e-12 and cold["clock_hz"] == 1.7e9 and cold["launches"] == 5
    cf = bench.cold_with_frac(cold, 393216000)
    assert abs(cf["frac_after_first"] - 393216000 / 0.07e-3 / 1e9 / 8000.0) < 1e-9 and "what" in cf
    clk = bench.sampled_clock(fake, fake.launch, 50, 0.07)
    assert clk["hz"] == 1.7e9 and abs(clk["kernel_ms_while_sampling"] - 0.07) < 1e-9
    assert "error" in bench.sampled_clock(object(), fake.launch, 5, 0.07)    # a diagnostic never raises
    # the "configs" entry
    full = {"metric": "m", "value": 1.0, "unit": "u", "ms_per_step": 0.1, "steps": 5, "dtype": "f32",
            "config": {"workload": "w", "kernel_variant": "generic"},
            "roofline": {"bound": "hbm", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1, "frac_cold": 0.1, "traffic": 3.0,
                         "algorithmic_bytes_per_launch": 3, "kernel": "k", "kernel_ms": 0.1, "kernel_ms_cold": 0.2, "device_copy": {},
                         "clock": {"hz": 2.1e9}},
            "roofline_valu": {"frac": 0.5, "valu_insts_per_launch": 1, "issue_cycles_per_inst": 2.0, "clock_hz": 1.0, "source": "s", "peak": 1},
            "parity_sample": {"ok": True}, "cpu_baseline": {"value": 1}}
    entry = bench.compact(full)
    for key in ("ms_per_step", "value", "kernel", "kernel_ms", "kernel_ms_cold", "roofline", "roofline_valu", "parity_sample", "workload"):
        assert key in entry
    assert entry["roofline"]["traffic"] == 3.0 and "device_copy" not in entry["roofline"] and "cpu_baseline" not in entry
    assert entry["roofline"]["clock"]["hz"] == 2.1e9
    assert set(bench.SECONDARY) == {"c2f", "c3", "c4", "c4s", "c5", "c5b"} and bench.CONFIGS["c5"]["layout"] == "time_major"
    for name, cfg in bench.CONFIGS.items():   # kernel names the traffic files must carry to be attached
        for spec in (True, False):
            assert isinstance(bench.kernel_name(dict(cfg, specialized=spec, dof=6)), str)


def test_trac_ik_multi_start():
    """manipulapy_amd/trac_ik.py: the batched form (all guesses in one kinematics.inverse launch), the host form on bare
    callables, an unreachable target (best configuration, success False, inside the limits), a 9-joint arm (host form)."""
    import manipulapy_amd as mp

    robot = mp.load_robot("ur5")[0]
    rng = np.random.default_rng(12)
    lo = np.array([l for l, _ in robot.joint_limits]); hi = np.array([h for _, h in robot.joint_limits])
    np.random.seed(5)
    solved = 0
    for _ in range(6):
        q = rng.uniform(np.maximum(lo, -2.5), np.minimum(hi, 2.5))
        T = robot.forward_kinematics(q)
        th, ok, t = robot.trac_ik(T, timeout=0.5, num_restarts=8)
        assert th.shape == (6,) and th.dtype == np.float64 and isinstance(ok, bool) and isinstance(t, float) and t >= 0
        assert (th >= lo - 1e-12).all() and (th <= hi + 1e-12).all()
        if ok:
            solved += 1
            Tc = robot.forward_kinematics(th)
            assert np.linalg.norm(Tc[:3, 3] - T[:3, 3]) < 1e-4
            assert np.arccos(np.clip(0.5 * (np.trace(Tc[:3, :3].T @ T[:3, :3]) - 1), -1, 1)) < 1e-4
    assert solved >= 5
    # bare callables: the sequential host loop; a warm start converges at once
    q = rng.uniform(-1.5, 1.5, 6)
    T = robot.forward_kinematics(q)
    solver = mp.TracIKSolver(lambda th: robot.forward_kinematics(th), lambda th: robot.jacobian(th), robot.joint_limits, 6)
    th, ok, t = solver.solve(T, theta0=q + 0.05, timeout=2.0)
    assert ok and np.abs(robot.forward_kinematics(th) - T).max() < 2e-4
    # unreachable: 5 m away
    far = T.copy(); far[:3, 3] = [5.0, 0.0, 0.0]
    for s in (lambda: robot.trac_ik(far, timeout=0.05), lambda: solver.solve(far, timeout=0.05)):
        th, ok, t = s()
        assert ok is False and np.isfinite(th).all() and (th >= lo - 1e-12).all() and (th <= hi + 1e-12).all()
    with pytest.raises(ValueError):
        robot.trac_ik(np.eye(3))
    # 9 joints: the batched solver stops at 8, the host loop serves
    z, proc = _jaco("jaco_6dof")
    jaco = proc.serial_manipulator
    n = len(jaco.joint_limits)
    assert n == 9
    q = z["jaco_6dof__theta"]
    T = jaco.forward_kinematics(q)
    th, ok, t = jaco.trac_ik(T, theta0=q + 0.02, timeout=3.0)
    assert th.shape == (n,) and ok and np.abs(jaco.forward_kinematics(th) - T).max() < 2e-4


def _gain_sweep_cases():
    z = np.load(golden_path("gain_sweep_ur5.npz"))
    return z, [(t, z[f"{t}_theta"], z[f"{t}_des"], float(z[f"{t}_dt"]), int(z[f"{t}_steps"])) for t in "abc"]


def check_gain_sweep(controller):
    """find_ultimate_gain_and_period against the reference's own runs (tests/golden/gain_sweep_ur5.npz: 6, 24 and 1 gains
    visited): the same ultimate gain / period, the same gains, every run's error history (the reference's Coriolis term is a
    finite difference of mass matrices: 1e-7 of the history's scale is its noise over <= 40 steps)."""
    z, cases = _gain_sweep_cases()
    for tag, th, des, dt, steps in cases:
        th_in = th.copy()
        Ku, Tu, gains, errs = controller.find_ultimate_gain_and_period(th_in, des, dt, steps)
        assert isinstance(Ku, float) and isinstance(Tu, float) and isinstance(gains, list) and isinstance(errs, list)
        np.testing.assert_array_equal(th_in, th)                                   # the caller's state is not advanced
        assert Ku == pytest.approx(float(z[f"{tag}_Ku"]), rel=1e-12) and Tu == pytest.approx(float(z[f"{tag}_Tu"]), rel=1e-12)
        np.testing.assert_allclose(gains, z[f"{tag}_gains"], rtol=1e-12)
        want = z[f"{tag}_errors"]
        assert len(errs) == len(want)
        np.testing.assert_allclose(np.stack(errs), want, rtol=0, atol=1e-7 * np.abs(want).max())


def test_gain_sweep_on_the_cpu_launcher():
    sm, dyn, lim = mp.load_robot("ur5")
    check_gain_sweep(mp.ManipulatorController(dyn))
    from manipulapy_amd import _hip

    # the launcher itself: per-run gains / targets, a damped run, the blow-up stop, zero steps, bad shapes
    model = dyn.hip_model()
    th0 = np.tile(np.full(6, 0.1), (4, 1)); des = np.tile(np.full(6, 0.5), (4, 1))
    err, cnt = _hip.cpu_pd_regulation(model, th0, des, [5.0, 5.0, 7.0, 0.0], [0.0, 1e-3, 0.0, 0.0], None, 0.01, 60)
    assert err.shape == (4, 60) and cnt.tolist()[:2] == [60, 60] and cnt[3] == 60
    assert np.isfinite(err[:2, :10]).all() and not np.array_equal(err[0, :10], err[1, :10])   # the derivative gain acts
    # a gain that diverges slowly enough to pass 1e10 after step 10: the run stops there, the rest keeps the NaN fill
    assert 11 < cnt[2] < 60 and np.isnan(err[2, cnt[2]:]).all() and err[2, cnt[2] - 1] > 1e10 and np.isfinite(err[2, :cnt[2]]).all()
    e0, c0 = _hip.cpu_pd_regulation(model, th0, des, np.ones(4), np.zeros(4), [0, 0, -9.81], 0.01, 0)
    assert e0.shape == (4, 0) and (c0 == 0).all()
    with pytest.raises(ValueError):
        _hip.cpu_pd_regulation(model, th0[:, :5], des, np.ones(4), np.zeros(4), None, 0.01, 5)
    # one thread == all threads, and a 9-joint arm runs (run-time-n body)
    a = _hip.cpu_pd_regulation(model, th0, des, [1, 2, 3, 4], np.zeros(4), None, 0.01, 30, nthreads=1)
    b = _hip.cpu_pd_regulation(model, th0, des, [1, 2, 3, 4], np.zeros(4), None, 0.01, 30, nthreads=4)
    np.testing.assert_array_equal(a[0], b[0])
    zj, proc = _jaco("jaco_6dof")
    th = zj["jaco_6dof__theta"]
    Ku, Tu, gains, errs = mp.ManipulatorController(proc.dynamics).find_ultimate_gain_and_period(th, th + 0.05, 0.002, 20)
    assert Ku > 0 and Tu > 0 and len(gains) == len(errs) >= 1 and all(np.isfinite(e).all() for e in errs)
    # its first run = stepping forward_dynamics by hand
    q, w = th.copy(), np.zeros_like(th)
    for step in range(3):
        al = proc.dynamics.forward_dynamics(q, w, gains[0] * (th + 0.05 - q), [0, 0, -9.81], np.zeros(6))
        w = w + al * 0.002; q = q + w * 0.002
        assert errs[0][step] == pytest.approx(np.linalg.norm(q - th - 0.05), rel=1e-9)
    legacy = _legacy_objects()[1]["A"]
    with pytest.raises(NotImplementedError):
        mp.ManipulatorController(legacy).find_ultimate_gain_and_period(np.zeros(6), np.ones(6) * 0.1, 0.01, 3)


def check_urdf_processor_surface():
    """URDFToSerialManipulator's convenience surface against the reference on ten of the urdf_suite files
    (tests/golden/urdf_api.npz): names, all-link forward kinematics (one configuration, a batch, the persisting current
    configuration), transforms between links, end-effector batches, the validation report."""
    import os

    z = np.load(golden_path("urdf_api.npz"))
    names = sorted({k.split("__")[0] for k in z.files if "__" in k})
    assert len(names) == 10
    kernel_routed = 0
    for name in names:
        proc = mp.URDFToSerialManipulator(golden_path(os.path.join("urdf_suite", f"{name}.urdf")), tip_link=str(z[f"{name}__ee"]))
        ref_joints = [str(x) for x in z[f"{name}__joint_names"]]
        assert proc.num_dofs == int(z[f"{name}__num_dofs"]) and proc.end_effector_name == str(z[f"{name}__ee"])
        # (several roots: the reference orders them by a hash-seeded set, this reader by the file - same joints, other order)
        assert sorted(proc.joint_names) == sorted(ref_joints) and (proc.joint_names == ref_joints or name == "fixture_multi_root")
        assert proc.link_names == [str(x) for x in z[f"{name}__link_names"]]
        assert proc.print_joint_info() == {"num_joints": len(z[f"{name}__all_joint_names"]), "joint_names": [str(x) for x in z[f"{name}__all_joint_names"]]}
        perm = [ref_joints.index(j) for j in proc.joint_names]          # reference column of each of this reader's joints
        np.testing.assert_allclose(proc.joint_limits_array, z[f"{name}__limits"][perm], atol=0)
        cfg, cfgs = z[f"{name}__cfg"][perm], z[f"{name}__cfgs"][:, perm]
        links = [str(x) for x in z[f"{name}__fk_links"]]
        fk = proc.link_fk(cfg)
        assert sorted(fk) == sorted(links)
        for k, T in zip(links, z[f"{name}__fk"]):
            np.testing.assert_allclose(fk[k], T, atol=1e-12, err_msg=f"{name} {k}")
        again = proc.link_fk(None)                                         # the configuration persists
        for k, T in zip(links, z[f"{name}__fk_current"]):
            np.testing.assert_allclose(again[k], T, atol=1e-12)
        by_name = proc.link_fk({j: float(v) for j, v in zip(proc.joint_names, cfg)})
        np.testing.assert_array_equal(by_name[links[-1]], fk[links[-1]])
        fkb = proc.batch_forward_kinematics(cfgs)
        for k, T in zip(links, z[f"{name}__fkb"]):
            np.testing.assert_allclose(fkb[k], T, atol=1e-12)
        np.testing.assert_allclose(proc.get_end_effector_transforms(cfgs), z[f"{name}__ee_batch"], atol=1e-12)
        np.testing.assert_allclose(proc.batch_forward_kinematics(cfgs, links[2 % len(links)]), z[f"{name}__fkb"][2 % len(links)], atol=1e-12)
        kernel_routed += proc._serial_to_tip()
        a, b = (str(x) for x in z[f"{name}__tf_pair"])
        np.testing.assert_allclose(proc.get_transform(a, b, cfg), z[f"{name}__tf"], atol=1e-12)
        np.testing.assert_allclose(proc.get_transform(a, "world"), z[f"{name}__tf_world"], atol=1e-12)
        if name != "fixture_multi_root":                                   # (the screw model's column order follows the joint order)
            np.testing.assert_allclose(proc.forward_kinematics(cfg), z[f"{name}__fk_ee"], atol=1e-12)
            np.testing.assert_allclose(proc.jacobian(cfg), z[f"{name}__jac"], atol=1e-12)
        v = proc.validate()
        assert v["valid"] == bool(z[f"{name}__valid"])
        assert ([f"{i['severity']}|{i['message']}" for i in v["issues"]] or ["<none>"]) == [str(x) for x in z[f"{name}__issues"]]
        with pytest.raises(ValueError):
            proc.get_transform("no_such_link")
        with pytest.raises(ValueError):
            proc.batch_forward_kinematics(cfgs[:, :-1] if proc.num_dofs > 1 else np.zeros((2, 5)))
        with pytest.raises(ValueError):
            proc.batch_forward_kinematics(cfgs, "no_such_link")
        assert repr(proc).startswith("URDFToSerialManipulator(urdf=") and proc.get_link("no_such_link") is None
        assert proc.get_link(links[0]).name == links[0] and proc.get_link(proc, links[0]).name == links[0]
        np.testing.assert_array_equal(proc.transform_to_xyz(fk[links[-1]]), fk[links[-1]][:3, 3])
    assert kernel_routed >= 2          # ur5 and the prismatic chain are plain serial chains: their tip batches are kernel launches
    np.testing.assert_allclose(mp.URDFToSerialManipulator.w_p_to_slist(z["w_p_in_w"], z["w_p_in_p"], 4), z["w_p_out"], atol=1e-15)


def test_urdf_processor_surface_on_the_cpu_launchers():
    check_urdf_processor_surface()
    import os

    proc = mp.URDFToSerialManipulator(golden_path(os.path.join("urdf_suite", "ur5.urdf")), tip_link="tool0")
    with pytest.warns(DeprecationWarning):
        assert proc.load_urdf("ignored") is proc.robot_data
    assert proc.initialize_serial_manipulator() is proc.serial_manipulator and proc.initialize_manipulator_dynamics() is proc.dynamics
    sm2, dyn2 = proc.get_serial_manipulator(), proc.get_manipulator_dynamics()
    q = np.array([0.1, -0.7, 0.5, 0.2, -0.3, 0.4])
    np.testing.assert_allclose(sm2.forward_kinematics(q), proc.forward_kinematics(q), atol=1e-14)
    assert dyn2.Mlist_per_link is None                                      # the reference's to_manipulator_dynamics(): the legacy object
    with pytest.warns(UserWarning):
        dyn2.mass_matrix(q)
    T = proc.forward_kinematics(q)
    for method, kw in (("iterative", dict(max_iterations=300)), ("smart", dict(max_iterations=300)), ("robust", dict(max_iterations=300, max_attempts=3))):
        th, ok, it = proc.inverse_kinematics(T, initial_guess=q + 0.05, method=method, eomg=1e-5, ev=1e-5, **kw)
        assert th.shape == (6,) and isinstance(it, (int, np.integer))
        if ok:
            assert np.abs(proc.forward_kinematics(th)[:3, 3] - T[:3, 3]).max() < 1e-4
    broken = mp.URDFToSerialManipulator(golden_path(os.path.join("urdf_suite", "fixture_multi_root.urdf")))
    assert broken.validate()["valid"] and len(broken.validate()["issues"]) == 2


def test_adaptive_multi_start_ik_ladder():
    """ik_helpers.adaptive_multi_start_ik: the ladder is walked in order with the reference's parameters, stops at the first
    success, sums the iterations, skips raising attempts, reports "none (failed)" with the last attempt's configuration."""
    calls = []

    def solver(T, strategy, eomg, ev, max_iterations, damping, step_cap):
        calls.append((strategy, damping, step_cap, eomg, ev, max_iterations))
        if len(calls) == 2:
            raise RuntimeError("skipped")
        return np.full(3, float(len(calls))), len(calls) == 4, 10 * len(calls)

    th, ok, total, name = mp.ik_helpers.adaptive_multi_start_ik(solver, np.eye(4))
    assert ok and name == "random" and total == 10 + 30 + 40 and th[0] == 4.0
    assert [c[:3] for c in calls] == [("workspace_heuristic", 0.02, 0.3), ("midpoint", 0.03, 0.3), ("random", 0.02, 0.3), ("random", 0.03, 0.25)]
    assert calls[0][3:] == (2e-3, 2e-3, 1500)
    calls.clear()
    th, ok, total, name = mp.ik_helpers.adaptive_multi_start_ik(lambda *a, **k: (np.zeros(2), False, 7), np.eye(4), max_attempts=3)
    assert not ok and name == "none (failed)" and total == 21
    # on a real robot
    sm = mp.load_robot("ur5")[0]
    q = np.array([0.3, -1.0, 0.8, -0.4, 0.5, 0.2])
    T = sm.forward_kinematics(q)
    np.random.seed(3)
    th, ok, total, name = mp.ik_helpers.adaptive_multi_start_ik(sm.smart_inverse_kinematics, T, max_attempts=6)
    assert ok and np.abs(sm.forward_kinematics(th)[:3, 3] - T[:3, 3]).max() < 5e-3 and total > 0


def test_plan_trajectory_and_the_mesh_less_collision_checker():
    """OptimizedTrajectoryPlanning.plan_trajectory against the reference's waypoints on the mesh-less UR5 (tests/golden/plan_ur5.npz):
    with obstacles (one potential-field step per waypoint, the checker reports no collision), without, and with neither checker
    nor field (an unreadable URDF: both None, as in the reference)."""
    z = np.load(golden_path("plan_ur5.npz"))
    sm, dyn, lim = mp.load_robot("ur5")
    pl = mp.OptimizedTrajectoryPlanning(sm, mp.robot_urdf("ur5"), dyn, lim, use_cuda=False)
    assert pl.collision_checker is not None and pl.potential_field is not None and pl.collision_checker.convex_hulls == {}
    obstacles = [o for o in z["obstacles"]]
    got = pl.plan_trajectory(z["start"].tolist(), z["target"].tolist(), obstacles)
    assert isinstance(got, list) and len(got) == 6 and isinstance(got[0], list)
    np.testing.assert_allclose(got, z["with_obstacles"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(pl.plan_trajectory(z["start"], z["target"], []), z["without_obstacles"], rtol=0, atol=1e-15)
    assert [pl.collision_checker.check_collision(q) for q in (z["start"], z["target"], np.zeros(6))] == z["collision_free"].tolist()
    pl2 = mp.OptimizedTrajectoryPlanning(sm, "nonexistent.urdf", dyn, lim, use_cuda=False)
    assert (pl2.collision_checker is None, pl2.potential_field is None) == tuple(z["no_checker"].tolist())
    np.testing.assert_allclose(pl2.plan_trajectory(z["start"], z["target"], obstacles), z["no_checker_plan"], rtol=0, atol=1e-15)
    from manipulapy_amd.potential_field import CollisionChecker

    with pytest.raises(FileNotFoundError):
        CollisionChecker("nonexistent.urdf")
    with pytest.raises(NotImplementedError):
        CollisionChecker(mp.robot_urdf("ur5"), backend="pybullet")


def test_planner_benchmark_helpers_without_a_gpu():
    """benchmark_performance / benchmark_all_kernels (reference planning/trajectory_planning.py:526-830): result keys and shapes;
    without GPU routing the kernel sweep returns {} like the reference without CUDA, and no CPU-vs-CPU speed-up is invented."""
    sm, dyn, lim = mp.load_robot("ur5")
    pl = mp.OptimizedTrajectoryPlanning(sm, mp.robot_urdf("ur5"), dyn, lim, use_cuda=False)
    r = pl.benchmark_performance([{"N": 40, "joints": 6, "name": "tiny"}, {"N": 80, "joints": 6, "name": "small"}])
    assert sorted(r) == ["small", "tiny"]
    for name, N in (("tiny", 40), ("small", 80)):
        e = r[name]
        assert e["trajectory_shape"] == (N, 6) and e["used_gpu"] is False and "actual_speedup" not in e
        assert e["min_time"] <= e["mean_time"] <= e["max_time"] and e["elements_per_second"] > 0 and e["stats"]["cpu_calls"] == 3
    assert set(pl.benchmark_performance()) == {"Small", "Medium", "Large", "Very Large"}
    assert pl.benchmark_all_kernels(N=20, num_runs=1) == {}


def test_f32_rows_adaptive_precision_on_the_cpu_launcher(tables):
    """Round 4: the float32 inverse dynamics (kernels and CPU launcher share mp_rnea_row, csrc/mp_core.h) takes joint offsets as
    exact rotations of (sin q, cos q) and evaluates ill-conditioned rows - intermediate wrenches above 8 x the row's largest torque (MpRowScale) - in
    float64.  On 100 000 c2-distributed UR5 rows against the pinned C oracle: every row inside 1e-4 |ref| + 5e-6 max|row| with
    room to spare, a small share of the rows in float64 (in runs of consecutive timesteps), those at <= 0.2 x the bound, and the
    verdict a function of the row alone (the same row gives the same bits in any batch)."""
    import bench
    from manipulapy_amd import _hip
    from oracle import c_oracle

    tab = tables["ur5"]
    lim = tab.joint_limits
    rng = np.random.default_rng(20260705 + 2)
    s_ = rng.uniform(lim[:, 0], lim[:, 1], (100, 6)).astype(np.float32)
    e_ = rng.uniform(lim[:, 0], lim[:, 1], (100, 6)).astype(np.float32)
    o = ref.batch_joint_trajectory(lim, s_, e_, 2.0, 1000, 5)
    q, qd, qdd = (np.ascontiguousarray(o[k].reshape(-1, 6), dtype=np.float32) for k in ("positions", "velocities", "accelerations"))
    want = c_oracle.inverse_dynamics_rows(tab, q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64))[0]
    m = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, lim)
    tau = _hip.cpu_id_trajectory(m, q, qd, qdd, dtype=np.float32)
    par = bench.parity_rows(tau, want, "f32")
    assert par["ok"] and par["rows_over_first_bound"] == 0 and par["worst_over_tol"] <= 0.6, par
    in_f64 = _hip.cpu_id_row_precision(m, q, qd, qdd)
    assert 0.002 <= in_f64.mean() <= 0.05, in_f64.mean()
    tol = 1e-4 * np.abs(want) + 5e-6 * np.abs(want).max(axis=1, keepdims=True)
    assert (np.abs(tau[in_f64].astype(np.float64) - want[in_f64]) / tol[in_f64]).max() <= 0.2
    # float64 rows equal the float64 launcher's result rounded to float32 up to the float32 MODEL constants (1e-6 relative to the row)
    t64 = _hip.cpu_id_trajectory(m, q[in_f64].astype(np.float64), qd[in_f64].astype(np.float64), qdd[in_f64].astype(np.float64), dtype=np.float64)
    np.testing.assert_allclose(tau[in_f64], t64, rtol=0, atol=2e-6 * np.abs(t64).max(axis=1, keepdims=True).max())
    # per-row determinism: a shuffled batch gives the same bits row by row
    perm = rng.permutation(len(q))[:5000]
    np.testing.assert_array_equal(_hip.cpu_id_trajectory(m, q[perm], qd[perm], qdd[perm], dtype=np.float32), tau[perm])
    # the float64 floor follows the oracle's finite-difference noise at these speeds
    t64_all = _hip.cpu_id_trajectory(m, q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64), dtype=np.float64)
    assert bench.parity_rows(t64_all, want, "f64", qd=qd)["ok"]
