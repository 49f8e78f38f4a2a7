"""GPU parity tests (run on an MI355X: `pytest -m gpu`).  Everything goes through the C ABI
(manipulapy_amd/_hip.py -> libmanipula_hip.so); the checker is the CPU oracle / the golden fixtures.

Tolerances (BASELINE.json north_star: 1e-6 rel fp64 / 1e-4 rel fp32 against the NumPy CPU backend):
  fp64:  |d| <= 1e-6 * |ref| + 1e-7   — the 1e-7 absolute floor (N.m) sits above the reference's own
         central-difference noise (~3e-9, reference tests/test_dynamics_golden.py:62-76 uses atol 1e-8)
         and is needed for torque components that are exactly or nearly zero.
  fp32:  |d| <= 1e-4 * |ref| + 5e-6 * max|ref row|  — elementwise relative (north_star), with a floor tied to the
         row's own scale for near-zero components (float32 cancellation cannot be relative to ~0).  The floor is
         ~40 float32 ulps of the row maximum: the kernels measure 6e-7 of it (bench parity_sample), so an
         order-of-magnitude regression of the float32 recursion fails here.
"""
import os

import ctypes

import numpy as np
import pytest

from conftest import ROBOTS, golden_path
from oracle import ref_numpy as ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

F64_RTOL, F64_ATOL = 1e-6, 1e-7
F32_RTOL, F32_ROW = 1e-4, 5e-6


def assert_f64(got, want):
    np.testing.assert_allclose(got, want, rtol=F64_RTOL, atol=F64_ATOL)


def assert_f32(got, want):
    want = np.asarray(want, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    floor = F32_ROW * np.abs(want).max(axis=-1, keepdims=True) + 1e-12   # (absolute: rows whose true torques are exactly zero)
    bad = np.abs(got - want) > F32_RTOL * np.abs(want) + floor
    assert not bad.any(), f"max abs err {np.abs(got - want).max():.3e}, {bad.sum()} elements outside tolerance"


@pytest.fixture(scope="module")
def ctx():
    from manipulapy_amd import _hip

    c = _hip.HipContext(0)
    c.selftest()
    yield c
    c.destroy()


@pytest.fixture(scope="module")
def models(tables):
    from manipulapy_amd import _hip

    return {r: _hip.HipModel(t.S, t.Mcom, t.G, t.M_ee, t.joint_limits) for r, t in tables.items()}


def test_native_library_is_the_one_loaded(ctx):
    import manipulapy_amd
    from manipulapy_amd import _hip

    assert _hip.lib_path().endswith("libmanipula_hip.so")
    with open("/proc/self/maps") as f:
        assert "libmanipula_hip.so" in f.read()
    p = ctx.properties()
    assert p["multiprocessor_count"] > 0 and p["warp_size"] == 64
    assert manipulapy_amd.check_hip_availability()


@pytest.mark.parametrize("robot", ROBOTS)
def test_inverse_dynamics_golden_f64(robot, ctx, models, dyn_golden):
    z = dyn_golden[robot]
    for i in range(len(z["thetas"])):  # per-row Ftip -> one launch per configuration
        tau = ctx.id_trajectory_host(models[robot], z["thetas"][i:i + 1], z["dthetas"][i:i + 1], z["ddthetas"][i:i + 1],
                                     z["g"], z["ftips"][i], dtype=np.float64)
        assert_f64(tau[0], z["inverse_dynamics"][i])


@pytest.mark.parametrize("robot", ["ur5", "panda"])
def test_inverse_dynamics_reference_own_goldens_f64(robot, ctx, models):
    """The reference's own tests/data/dynamics_golden_*.npz (byte-identical copies)."""
    z = np.load(golden_path(f"dynamics_golden_{robot}.npz"))
    for i in range(len(z["thetas"])):
        tau = ctx.id_trajectory_host(models[robot], z["thetas"][i:i + 1], z["dthetas"][i:i + 1], z["ddthetas"][i:i + 1],
                                     z["g"], z["ftips"][i], dtype=np.float64)
        assert_f64(tau[0], z["inverse_dynamics"][i])
        zero = np.zeros_like(z["thetas"][i:i + 1])
        grav = ctx.id_trajectory_host(models[robot], z["thetas"][i:i + 1], zero, zero, z["g"], None, dtype=np.float64)
        assert_f64(grav[0], z["gravity_forces"][i])
        cor = ctx.id_trajectory_host(models[robot], z["thetas"][i:i + 1], z["dthetas"][i:i + 1], zero, np.zeros(3), None,
                                     dtype=np.float64)
        assert_f64(cor[0], z["velocity_quadratic_forces"][i])


@pytest.mark.parametrize("robot", ROBOTS)
def test_inverse_dynamics_golden_f32(robot, ctx, models, dyn_golden):
    z = dyn_golden[robot]
    zero_ftip = [i for i in range(len(z["thetas"])) if not z["ftips"][i].any()]
    tau = ctx.id_trajectory_host(models[robot], z["thetas"][zero_ftip], z["dthetas"][zero_ftip], z["ddthetas"][zero_ftip],
                                 z["g"], None, dtype=np.float32)
    assert tau.dtype == np.float32
    assert_f32(tau, z["inverse_dynamics"][zero_ftip])
    for i in range(5, len(z["thetas"]), 4):
        t = ctx.id_trajectory_host(models[robot], z["thetas"][i:i + 1], z["dthetas"][i:i + 1], z["ddthetas"][i:i + 1],
                                   z["g"], z["ftips"][i], dtype=np.float32)
        assert_f32(t, z["inverse_dynamics"][i:i + 1])


@pytest.mark.parametrize("robot", ROBOTS)
def test_fk_jacobian_golden_f64(robot, ctx, models, dyn_golden):
    z = dyn_golden[robot]
    T, J, tau = ctx.fk_jac_id_host(models[robot], z["thetas"], z["dthetas"], z["ddthetas"], z["g"], None)
    np.testing.assert_allclose(T, z["fk_space"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(J, z["jac_space"], rtol=1e-9, atol=1e-12)
    zero = [i for i in range(len(z["thetas"])) if not z["ftips"][i].any()]
    assert_f64(tau[zero], z["inverse_dynamics"][zero])
    # outputs can be requested separately
    T2, J2, tau2 = ctx.fk_jac_id_host(models[robot], z["thetas"], want_J=False)
    assert J2 is None and tau2 is None
    np.testing.assert_array_equal(T2, T)


def test_mass_matrix_and_forward_dynamics_via_api(dyn_golden, tables):
    """ManipulatorDynamics mirror under the hip backend (per-point API, float64)."""
    import manipulapy_amd as mp

    for robot in ("ur5", "panda"):
        sm, dyn, _ = mp.load_robot(robot)
        z = dyn_golden[robot]
        with mp.use_backend("hip"):
            for i in (0, 5, 11, 24):
                M = dyn.mass_matrix(z["thetas"][i])
                np.testing.assert_allclose(M, z["mass_matrix"][i], rtol=1e-6, atol=1e-8)
                assert_f64(dyn.gravity_forces(z["thetas"][i], z["g"]), z["gravity_forces"][i])
                assert_f64(dyn.velocity_quadratic_forces(z["thetas"][i], z["dthetas"][i]), z["velocity_quadratic_forces"][i])
                assert_f64(dyn.inverse_dynamics(z["thetas"][i], z["dthetas"][i], z["ddthetas"][i], z["g"], z["ftips"][i]),
                           z["inverse_dynamics"][i])
                qdd = dyn.forward_dynamics(z["thetas"][i], z["dthetas"][i], z["inverse_dynamics"][i], z["g"], z["ftips"][i])
                np.testing.assert_allclose(qdd, z["forward_dynamics"][i], rtol=1e-5, atol=1e-6)
                np.testing.assert_allclose(sm.forward_kinematics(z["thetas"][i]), z["fk_space"][i], atol=1e-12)
                np.testing.assert_allclose(sm.forward_kinematics(z["thetas"][i], "body"), z["fk_body"][i], atol=1e-12)
                np.testing.assert_allclose(sm.jacobian(z["thetas"][i], "body"), z["jac_body"][i], atol=1e-10)


def test_batch_trajectory_against_fixture_and_oracle(ctx, models):
    z = np.load(golden_path("trajectory_ur5.npz"))
    m = models["ur5"]
    for tag in ("q1000", "c500"):
        N, method, Tf = z[f"jt_{tag}_args"]
        p, v, a = ctx.batch_trajectory_host(m, z["start"][None], z["end"][None], float(Tf), int(N), int(method))
        o = ref.joint_trajectory(z["joint_limits"], z["start"], z["end"], float(Tf), int(N), int(method))
        for got, key in ((p, "positions"), (v, "velocities"), (a, "accelerations")):
            assert got.dtype == np.float32 and got.shape == (1, int(N), 6)
            # float64 polynomial rounded to float32: at most 1 ulp apart from the oracle (FMA contraction)
            np.testing.assert_allclose(got[0], o[key], rtol=2.5e-7, atol=1e-7)
            np.testing.assert_allclose(got[0], z[f"jt_{tag}_{key}"], rtol=3e-7, atol=1e-6)
    # clip to the joint limits
    p, _, _ = ctx.batch_trajectory_host(m, z["start"][None], z["jt_clip_end"][None], 1.0, 32, 5)
    np.testing.assert_allclose(p[0], z["jt_clip_positions"], rtol=3e-7, atol=1e-6)
    # batch + unsupported method -> zeros (planning/trajectory.py:67-68)
    p, v, a = ctx.batch_trajectory_host(m, z["batch_start"], z["batch_end"], 2.0, 16, 5)
    np.testing.assert_allclose(p, z["batch_positions"], rtol=3e-7, atol=1e-6)
    np.testing.assert_allclose(a, z["batch_accelerations"], rtol=3e-7, atol=1e-6)
    p, v, a = ctx.batch_trajectory_host(m, z["batch_start"], z["batch_end"], 2.0, 16, 9)
    assert not v.any() and not a.any()
    np.testing.assert_array_equal(p, np.clip(np.broadcast_to(z["batch_start"][:, None, :], p.shape),
                                             z["joint_limits"][:, 0].astype(np.float32), z["joint_limits"][:, 1].astype(np.float32)))
    # N = 3 known answer (reference tests/test_backend_dispatch.py:2712-2727)
    m2 = models["ur5"]
    s = np.zeros((1, 6), np.float32); e = np.ones((1, 6), np.float32)
    p, v, a = ctx.batch_trajectory_host(m2, s, e, 2.0, 3, 5)
    np.testing.assert_array_equal(p[0, :, 0], np.array([0, 0.5, 1], np.float32))
    np.testing.assert_array_equal(v[0, :, 0], np.array([0, 0.9375, 0], np.float32))
    np.testing.assert_array_equal(a[0, :, 0], np.zeros(3, np.float32))


def test_inverse_dynamics_trajectory_planner_level(tables):
    """OptimizedTrajectoryPlanning under the hip backend vs the reference's planner dump."""
    import manipulapy_amd as mp

    z = np.load(golden_path("trajectory_ur5.npz"))
    sm, dyn, lim = mp.load_robot("ur5")
    with mp.use_backend("hip"):
        pl = mp.OptimizedTrajectoryPlanning(sm, None, dyn, lim)
        assert pl.cuda_available
        tau = pl.inverse_dynamics_trajectory(z["idt_q"], z["idt_qd"], z["idt_qdd"])  # float64 in -> f64 kernel, f32 out
        assert tau.dtype == np.float32 and tau.shape == (64, 6)
        np.testing.assert_allclose(tau, z["idt_tau_f32"], rtol=2e-6, atol=2e-6)
        tau32 = pl.inverse_dynamics_trajectory(z["idt_q"].astype(np.float32), z["idt_qd"].astype(np.float32),
                                               z["idt_qdd"].astype(np.float32))
        assert_f32(tau32, z["idt_tau_f64"])
        pl2 = mp.OptimizedTrajectoryPlanning(sm, None, dyn, lim, torque_limits=z["idt_torque_limits"])
        tc = pl2.inverse_dynamics_trajectory(z["idt_q"][:16], z["idt_qd"][:16], z["idt_qdd"][:16], None, z["idt_ftip"])
        np.testing.assert_allclose(tc, z["idt_tau_f32_clip_ftip"], rtol=2e-6, atol=2e-6)
        assert tc.max() <= 15.0 and tc.min() >= -20.0
        # empty trajectory
        assert pl.inverse_dynamics_trajectory(np.zeros((0, 6)), np.zeros((0, 6)), np.zeros((0, 6))).shape == (0, 6)
        # joint_trajectory on the device (N * n above the work threshold)
        N = 20000
        r = pl.joint_trajectory(z["start"], z["end"], 2.0, N, 5)
        assert pl.performance_stats["gpu_calls"] >= 3
        o = ref.joint_trajectory(lim, z["start"], z["end"], 2.0, N, 5)
        np.testing.assert_allclose(r["positions"], o["positions"], rtol=2.5e-7, atol=1e-7)
        np.testing.assert_allclose(r["accelerations"], o["accelerations"], rtol=2.5e-7, atol=1e-6)


def test_fused_equals_two_step_pipeline(ctx, models):
    """fused kernel == batch_trajectory followed by id_trajectory: the float32 intermediates are the
    same, the two kernels are separate compilations (different FMA contraction), so agreement is to a
    few float32 ulps of the row's largest torque, not bit for bit."""
    rng = np.random.default_rng(7)
    for robot in ("ur5", "panda"):
        m = models[robot]
        z = np.load(golden_path(f"model_{robot}.npz"))
        lo, hi = z["joint_limits"][:, 0], z["joint_limits"][:, 1]
        B, N = 37, 129  # ragged: neither a multiple of the block nor of the wave
        s = rng.uniform(lo, hi, (B, m.n)).astype(np.float32)
        e = rng.uniform(lo, hi, (B, m.n)).astype(np.float32)
        p, v, a = ctx.batch_trajectory_host(m, s, e, 2.0, N, 5)
        two = ctx.id_trajectory_host(m, p.reshape(-1, m.n), v.reshape(-1, m.n), a.reshape(-1, m.n), None, None)
        fused = ctx.traj_id_fused_host(m, s, e, 2.0, N, 5)
        f2 = fused.reshape(-1, m.n)
        assert (np.abs(f2 - two) <= 4e-6 * np.abs(two).max(axis=1, keepdims=True)).all()


@pytest.mark.parametrize("robot,dtype", [("ur5", np.float32), ("iiwa14", np.float64), ("panda", np.float32), ("xarm6", np.float64)])
def test_random_rows_against_oracle(robot, dtype, ctx, models, tables):
    """Seeded random rows (positions inside the joint limits) vs the CPU oracle, incl. a wrench."""
    rng = np.random.default_rng(20260705)
    tab, m = tables[robot], models[robot]
    rows = 48
    q = rng.uniform(tab.joint_limits[:, 0], tab.joint_limits[:, 1], (rows, tab.n)).astype(dtype)
    qd = rng.uniform(-2, 2, (rows, tab.n)).astype(dtype)
    qdd = rng.uniform(-4, 4, (rows, tab.n)).astype(dtype)
    F = np.array([1.0, -2.0, 0.5, 3.0, -1.5, 0.75])
    g = np.array([0.3, -0.2, -9.81])
    want = ref.inverse_dynamics_trajectory(tab, q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64), g, F,
                                           dtype=np.float64)
    got = ctx.id_trajectory_host(m, q, qd, qdd, g, F, dtype=dtype)
    (assert_f32 if dtype == np.float32 else assert_f64)(got, want)


def test_size_independent_properties_full_config(ctx, models):
    """BASELINE config 2 size (B=4096 x N=1000, UR5 float32) through device buffers:
    (a) linearity in qdd:  ID(q, 0, a1 + a2, g=0) == ID(q, 0, a1, 0) + ID(q, 0, a2, 0)
    (b) gravity superposition: ID(q, qd, qdd, g) - ID(q, qd, qdd, 0) == ID(q, 0, 0, g)
    (c) the fused kernel equals the two-step pipeline on every one of the 4.1 M rows."""
    m = models["ur5"]
    z = np.load(golden_path("model_ur5.npz"))
    rng = np.random.default_rng(20260705 + 2)
    B, N, n = 4096, 1000, 6
    lo, hi = z["joint_limits"][:, 0], z["joint_limits"][:, 1]
    s = rng.uniform(lo, hi, (B, n)).astype(np.float32)
    e = rng.uniform(lo, hi, (B, n)).astype(np.float32)
    rows = B * N
    nb = rows * n * 4
    d_s, d_e = ctx.to_device(s), ctx.to_device(e)
    bufs = [ctx.alloc(nb) for _ in range(7)]
    d_q, d_qd, d_qdd, d_t1, d_t2, d_t3, d_zero = bufs
    ctx.lib.mp_memset(ctx.handle, d_zero.ptr, 0, nb)
    ctx.batch_trajectory(m, d_s, d_e, B, N, 2.0, 5, d_q, d_qd, d_qdd)
    zero3 = np.zeros(3)
    # (c)
    ctx.id_trajectory(m, d_q, d_qd, d_qdd, rows, d_t1)
    ctx.traj_id_fused(m, d_s, d_e, B, N, 2.0, 5, d_t2)
    t_two = d_t1.download((rows, n), np.float32)
    t_fused = d_t2.download((rows, n), np.float32)
    assert (np.abs(t_two - t_fused) <= 2e-5 * np.abs(t_two).max(axis=1, keepdims=True)).all()
    assert np.isfinite(t_two).all()
    # (b)
    ctx.id_trajectory(m, d_q, d_qd, d_qdd, rows, d_t2, g=zero3)
    ctx.id_trajectory(m, d_q, d_zero, d_zero, rows, d_t3)
    t_nog = d_t2.download((rows, n), np.float32)
    t_g = d_t3.download((rows, n), np.float32)
    scale = np.abs(t_two).max(axis=1, keepdims=True) + np.abs(t_g).max(axis=1, keepdims=True)
    assert (np.abs((t_two - t_nog) - t_g) <= 2e-5 * scale + 1e-5).all()
    # (a): a1 = qdd, a2 = qd used as a second acceleration field
    ctx.id_trajectory(m, d_q, d_zero, d_qdd, rows, d_t1, g=zero3)
    ctx.id_trajectory(m, d_q, d_zero, d_qd, rows, d_t2, g=zero3)
    a1 = d_t1.download((rows, n), np.float32)
    a2 = d_t2.download((rows, n), np.float32)
    qdd = d_qdd.download((rows, n), np.float32)
    qd = d_qd.download((rows, n), np.float32)
    d_qdd.upload(qdd + qd)
    ctx.id_trajectory(m, d_q, d_zero, d_qdd, rows, d_t3, g=zero3)
    a12 = d_t3.download((rows, n), np.float32)
    scale = np.abs(a1).max(axis=1, keepdims=True) + np.abs(a2).max(axis=1, keepdims=True)
    assert (np.abs(a12 - (a1 + a2)) <= 2e-5 * scale + 1e-5).all()
    # spot-check 16 rows of the full-size run against the oracle
    tab = ref.load_tables(golden_path("model_ur5.npz"))
    q = d_q.download((rows, n), np.float32)
    idx = rng.integers(0, rows, 16)
    want = ref.inverse_dynamics_trajectory(tab, q[idx].astype(np.float64), qd[idx].astype(np.float64),
                                           qdd[idx].astype(np.float64), dtype=np.float64)
    assert_f32(t_two[idx], want)
    for b in bufs + [d_s, d_e]:
        b.free()


def test_edge_cases_and_errors(ctx, models):
    from manipulapy_amd import _hip

    m = models["ur5"]
    # empty inputs are a no-op
    assert ctx.id_trajectory_host(m, np.zeros((0, 6), np.float32), np.zeros((0, 6), np.float32), np.zeros((0, 6), np.float32)).shape == (0, 6)
    assert ctx.batch_trajectory_host(m, np.zeros((0, 6)), np.zeros((0, 6)), 1.0, 5, 5)[0].shape == (0, 5, 6)
    # wrong dof
    with pytest.raises(ValueError):
        ctx.id_trajectory_host(m, np.zeros((4, 7)), np.zeros((4, 7)), np.zeros((4, 7)))
    # one row, 63/64/65 rows (wave boundaries), 257 rows (block boundary)
    rng = np.random.default_rng(3)
    tab = ref.load_tables(golden_path("model_ur5.npz"))
    big = rng.uniform(-1, 1, (257, 3, 6))
    full = ctx.id_trajectory_host(m, big[:, 0], big[:, 1], big[:, 2], dtype=np.float64)
    for rows in (1, 63, 64, 65):
        part = ctx.id_trajectory_host(m, big[:rows, 0], big[:rows, 1], big[:rows, 2], dtype=np.float64)
        np.testing.assert_array_equal(part, full[:rows])
    want = ref.inverse_dynamics_trajectory(tab, big[250:, 0], big[250:, 1], big[250:, 2], dtype=np.float64)
    assert_f64(full[250:], want)
    # torque clip inside the kernel
    mc = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits, np.array([[-5.0, 4.0]] * 6))
    clipped = ctx.id_trajectory_host(mc, big[:, 0], big[:, 1], big[:, 2], dtype=np.float64)
    np.testing.assert_array_equal(clipped, np.clip(full, -5.0, 4.0))
    # misaligned device pointer is rejected, not launched
    d = ctx.alloc(1024)
    with pytest.raises(_hip.HipError):
        ctx.id_trajectory(m, d.offset(4), d, d, 1, d)
    d.free()


def test_2r_planar_analytical_on_gpu(ctx):
    """Analytical 2R planar arm (reference tests/test_v132_regressions.py:126-286) on the device, n = 2."""
    from manipulapy_amd import _hip
    from test_oracle_golden import _planar_2r

    tab, (l1, l2, m1, m2) = _planar_2r()
    m = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee)
    th = np.array([[0.3, -0.7], [1.2, 0.4]])
    dq = np.array([[0.7, -0.4], [0.1, 0.9]])
    ddq = np.array([[0.2, 0.5], [-1.0, 0.3]])
    g = np.array([0.0, -9.81, 0.0])
    tau = ctx.id_trajectory_host(m, th, dq, ddq, g, None, dtype=np.float64)
    for r in range(2):
        c2, s2 = np.cos(th[r, 1]), np.sin(th[r, 1])
        M = np.array([[m1 * l1**2 + m2 * (l1**2 + 2 * l1 * l2 * c2 + l2**2), m2 * (l1 * l2 * c2 + l2**2)],
                      [m2 * (l1 * l2 * c2 + l2**2), m2 * l2**2]])
        c = np.array([-m2 * l1 * l2 * s2 * (2 * dq[r, 0] * dq[r, 1] + dq[r, 1] ** 2), m2 * l1 * l2 * s2 * dq[r, 0] ** 2])
        c1, c12 = np.cos(th[r, 0]), np.cos(th[r, 0] + th[r, 1])
        gv = np.array([(m1 + m2) * 9.81 * l1 * c1 + m2 * 9.81 * l2 * c12, m2 * 9.81 * l2 * c12])
        np.testing.assert_allclose(tau[r], M @ ddq[r] + c + gv, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("robot", ROBOTS)
def test_mass_matrix_and_forward_dynamics_kernels(robot, ctx, models, dyn_golden):
    z = dyn_golden[robot]
    M = ctx.mass_matrix_host(models[robot], z["thetas"])
    np.testing.assert_allclose(M, z["mass_matrix"], rtol=1e-9, atol=1e-11)  # no finite difference in M: tight
    np.testing.assert_array_equal(M, np.swapaxes(M, 1, 2))                  # symmetrised like the reference
    assert (np.linalg.eigvalsh(M) > 0).all()
    for i in range(len(z["thetas"])):
        qdd = ctx.forward_dynamics_host(models[robot], z["thetas"][i:i + 1], z["dthetas"][i:i + 1],
                                        z["inverse_dynamics"][i:i + 1], z["g"], z["ftips"][i])
        # FD(ID(qdd)) == qdd: the recorded reference value carries its own finite-difference noise times M^-1
        np.testing.assert_allclose(qdd[0], z["forward_dynamics"][i], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(qdd[0], z["ddthetas"][i], rtol=1e-6, atol=1e-6)  # tau_ref carries ~3e-9 FD noise, times M^-1


def test_forward_dynamics_trajectory_fixture_and_oracle(tables):
    """forward_dynamics_trajectory vs the reference's planner dump (xarm6, N=8, intRes=2, per-step wrench)."""
    import manipulapy_amd as mp

    z = np.load(golden_path("fd_trajectory_xarm6.npz"))
    sm, dyn, _ = mp.load_robot("xarm6")
    with mp.use_backend("hip"):
        pl = mp.OptimizedTrajectoryPlanning(sm, None, dyn, z["joint_limits"])
        r = pl.forward_dynamics_trajectory(z["theta0"], z["dtheta0"], z["taumat"], z["g"], z["Ftipmat"], float(z["dt"]), int(z["intRes"]))
        for k in ("positions", "velocities", "accelerations"):
            assert r[k].dtype == np.float32 and r[k].shape == (8, 6)
            np.testing.assert_allclose(r[k], z[k], rtol=2e-6, atol=2e-6 * max(1.0, float(np.abs(z[k]).max())))
        np.testing.assert_array_equal(r["accelerations"][0], 0)
        # float32 state: looser (1e-4 rel of the column scale), same trajectory
        r32 = pl.forward_dynamics_trajectory(z["theta0"].astype(np.float32), z["dtheta0"].astype(np.float32),
                                             z["taumat"].astype(np.float32), z["g"], z["Ftipmat"].astype(np.float32),
                                             float(z["dt"]), int(z["intRes"]))
        for k in ("positions", "velocities", "accelerations"):
            assert np.abs(r32[k] - z[k]).max() <= 1e-4 * max(1.0, float(np.abs(z[k]).max()))
        with pytest.raises(IndexError):
            pl.forward_dynamics_trajectory(z["theta0"], z["dtheta0"], np.zeros((0, 6)), z["g"], np.zeros((0, 6)), 0.01, 1)
        # batch of 3 different roll-outs == 3 single calls; no wrench; joint-limit clip engages
        rng = np.random.default_rng(5)
        B, N = 3, 12
        th0 = rng.uniform(-0.5, 0.5, (B, 6)); dth0 = rng.uniform(-0.2, 0.2, (B, 6)); tm = rng.uniform(-1, 1, (B, N, 6))
        lim = z["joint_limits"].copy(); lim[1] = [-0.05, 0.05]
        th0[:, 1] = 0.0
        pl2 = mp.OptimizedTrajectoryPlanning(sm, None, dyn, lim)
        rb = pl2.batch_forward_dynamics_trajectory(th0, dth0, tm, z["g"], None, 0.02, 1)
        assert rb["positions"].shape == (B, N, 6)
        assert rb["positions"][:, :, 1].max() <= np.float32(0.05) and rb["positions"][:, :, 1].min() >= np.float32(-0.05)
        tab = ref.load_tables(golden_path("model_xarm6.npz"))
        for b in range(B):
            one = pl2.forward_dynamics_trajectory(th0[b], dth0[b], tm[b], z["g"], None, 0.02, 1)
            np.testing.assert_array_equal(one["positions"], rb["positions"][b])
            o = ref.forward_dynamics_trajectory(tab, th0[b], dth0[b], tm[b], z["g"], np.zeros((N, 6)), 0.02, 1, joint_limits=lim)
            for k in ("positions", "velocities", "accelerations"):
                np.testing.assert_allclose(one[k], o[k], rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(o[k]).max())))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_forward_dynamics_trajectory_tiles_and_alignment(tables, dtype):
    """The roll-out kernel moves its rows in 4-step tiles with 16-byte vector accesses when every lane's run is
    aligned: cover partial tiles (Nt not a multiple of 4), Nt = 1, unaligned runs (panda: n = 7, odd Nt), more than one
    wave with a ragged last wave, generic and specialised kernels, with and without per-step wrenches."""
    from manipulapy_amd import _hip

    ctx = _hip.HipContext(0)
    G0 = np.array([0.0, 0.0, -9.81])
    try:
        for robot, B in (("xarm6", 67), ("panda", 65)):
            tab = tables[robot]
            gen = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
            spec = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
            ctx.specialize(spec)
            n = tab.n
            for Nt in (1, 2, 3, 4, 5, 7, 8, 9):
                rng = np.random.default_rng(100 + Nt)
                th0 = rng.uniform(-0.5, 0.5, (B, n)); dth0 = rng.uniform(-0.2, 0.2, (B, n))
                tm = rng.uniform(-1, 1, (B, Nt, n)); Fm = rng.uniform(-1, 1, (B, Nt, 6))
                for wrench in (None, Fm):
                    a = ctx.fd_trajectory_host(gen, th0, dth0, tm, G0, wrench, 0.01, 2, dtype=dtype)
                    b = ctx.fd_trajectory_host(spec, th0, dth0, tm, G0, wrench, 0.01, 2, dtype=dtype)
                    for x, y in zip(a, b):
                        assert x.shape == (B, Nt, n) and np.isfinite(x).all()
                        np.testing.assert_allclose(x, y, rtol=2e-4, atol=2e-4 * max(1.0, float(np.abs(x).max())))
                    np.testing.assert_array_equal(a[2][:, 0], 0)
                    np.testing.assert_array_equal(a[0][:, 0], th0.astype(dtype).astype(np.float32))
                    for t in (0, B - 1):  # first lane of the first wave, last lane of the ragged wave
                        o = ref.forward_dynamics_trajectory(tab, th0[t], dth0[t], tm[t], G0,
                                                            np.zeros((Nt, 6)) if wrench is None else Fm[t], 0.01, 2,
                                                            joint_limits=tab.joint_limits)
                        tol = 1e-5 if dtype == np.float64 else 3e-4
                        for k, name in enumerate(("positions", "velocities", "accelerations")):
                            np.testing.assert_allclose(a[k][t], o[name], rtol=tol, atol=tol * max(1.0, float(np.abs(o[name]).max())))
    finally:
        ctx.destroy()


@pytest.mark.parametrize("robot", ROBOTS)
def test_specialised_kernels_match_generic_and_oracle(robot, tables, dyn_golden):
    """Run-time specialised float32 kernels (mp_model_specialize) vs the generic ones vs the oracle:
    inverse dynamics (with / without wrench, odd row count), fused generation + ID, forward-dynamics roll-out."""
    from manipulapy_amd import _hip

    tab, z = tables[robot], dyn_golden[robot]
    ctx = _hip.HipContext(0)
    try:
        gen = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
        spec = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
        assert not ctx.is_specialized(spec)
        ctx.specialize(spec)
        ctx.specialize(spec)  # idempotent
        assert ctx.is_specialized(spec) and not ctx.is_specialized(gen)
        rng = np.random.default_rng(17)
        rows = 2 * 257 + 1  # odd: the last row goes through the generic one-row kernel
        q = rng.uniform(tab.joint_limits[:, 0], tab.joint_limits[:, 1], (rows, tab.n)).astype(np.float32)
        qd = rng.uniform(-2, 2, (rows, tab.n)).astype(np.float32)
        qdd = rng.uniform(-4, 4, (rows, tab.n)).astype(np.float32)
        F = np.array([1.0, -2.0, 0.5, 3.0, -1.5, 0.75])
        for wrench in (None, F):
            a = ctx.id_trajectory_host(gen, q, qd, qdd, None, wrench)
            b = ctx.id_trajectory_host(spec, q, qd, qdd, None, wrench)
            assert (np.abs(a - b) <= 2e-5 * np.abs(a).max(axis=1, keepdims=True)).all()
            idx = np.arange(0, rows, 37)
            want = ref.inverse_dynamics_trajectory(tab, q[idx].astype(np.float64), qd[idx].astype(np.float64),
                                                   qdd[idx].astype(np.float64), None, wrench, dtype=np.float64)
            assert_f32(b[idx], want)
        # per-row forward dynamics: specialised float64 kernel vs generic and oracle (the host entry point is float64)
        for wrench in (None, F):
            tq = rng.uniform(-5, 5, (rows, tab.n))
            a = ctx.forward_dynamics_host(gen, q[:64].astype(np.float64), qd[:64].astype(np.float64), tq[:64], None, wrench)
            b = ctx.forward_dynamics_host(spec, q[:64].astype(np.float64), qd[:64].astype(np.float64), tq[:64], None, wrench)
            np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-9 * max(1.0, float(np.abs(a).max())))
            w0 = ref.forward_dynamics(tab, q[0].astype(np.float64), qd[0].astype(np.float64), tq[0], np.array([0, 0, -9.81]),
                                      np.zeros(6) if wrench is None else wrench)
            np.testing.assert_allclose(b[0], w0, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(w0).max())))
        # golden rows
        zero = [i for i in range(len(z["thetas"])) if not z["ftips"][i].any()]
        t = ctx.id_trajectory_host(spec, z["thetas"][zero], z["dthetas"][zero], z["ddthetas"][zero], z["g"], None)
        assert_f32(t, z["inverse_dynamics"][zero])
        # float64 specialised kernels: inverse dynamics and FK + Jacobian + ID
        for i in range(0, len(z["thetas"]), 4):
            t64 = ctx.id_trajectory_host(spec, z["thetas"][i:i + 1], z["dthetas"][i:i + 1], z["ddthetas"][i:i + 1], z["g"], z["ftips"][i],
                                         dtype=np.float64)
            assert_f64(t64[0], z["inverse_dynamics"][i])
        Ts, Js, taus = ctx.fk_jac_id_host(spec, z["thetas"], z["dthetas"], z["ddthetas"], z["g"], None)
        np.testing.assert_allclose(Ts, z["fk_space"], rtol=1e-9, atol=2e-12)
        np.testing.assert_allclose(Js, z["jac_space"], rtol=1e-9, atol=2e-12)
        assert_f64(taus[zero], z["inverse_dynamics"][zero])
        # fused generation + ID
        s = rng.uniform(tab.joint_limits[:, 0], tab.joint_limits[:, 1], (6, tab.n)).astype(np.float32)
        e = rng.uniform(tab.joint_limits[:, 0], tab.joint_limits[:, 1], (6, tab.n)).astype(np.float32)
        fa = ctx.traj_id_fused_host(gen, s, e, 2.0, 51, 5)
        fb = ctx.traj_id_fused_host(spec, s, e, 2.0, 51, 5)
        assert (np.abs(fa - fb) <= 2e-5 * np.abs(fa).max(axis=2, keepdims=True)).all()
        # forward-dynamics roll-out
        B, N = 5, 10
        th0 = rng.uniform(-0.3, 0.3, (B, tab.n)).astype(np.float32)
        if robot == "panda":
            th0[:, 3] = -1.5; th0[:, 5] = 1.0; th0[:, 7] = 0.02  # inside the one-sided limits
        dth0 = rng.uniform(-0.2, 0.2, (B, tab.n)).astype(np.float32)
        tm = rng.uniform(-1, 1, (B, N, tab.n)).astype(np.float32)
        Fm = np.broadcast_to(F.astype(np.float32), (B, N, 6)).copy()
        for wr in (None, Fm):
            ra = ctx.fd_trajectory_host(gen, th0, dth0, tm, None, wr, 0.01, 1, dtype=np.float32)
            rb = ctx.fd_trajectory_host(spec, th0, dth0, tm, None, wr, 0.01, 1, dtype=np.float32)
            for xa, xb in zip(ra, rb):
                assert np.abs(xa - xb).max() <= 2e-4 * max(1.0, float(np.abs(xa).max()))
    finally:
        ctx.destroy()


def test_rccl_communicator_single_rank(ctx):
    """mp_comm_* with nranks = 1 (all a 1-GPU box can exercise): the all-gather of one shard is a copy."""
    from manipulapy_amd import _hip

    uid = _hip.HipContext.comm_unique_id()
    assert len(uid) == _hip.UNIQUE_ID_BYTES and any(uid)
    comm = ctx.comm_create(uid, 1, 0)
    src = np.arange(4096, dtype=np.float32)
    d_s, d_r = ctx.to_device(src), ctx.alloc(src.nbytes)
    comm.allgather(d_s, d_r, src.nbytes)
    ctx.synchronize()
    np.testing.assert_array_equal(d_r.download(src.shape, np.float32), src)
    comm.destroy()
    with pytest.raises(ValueError):
        ctx.comm_create(b"short", 1, 0)
    d_s.free(); d_r.free()


def test_overlapped_chunk_exchange_single_rank(ctx, models, tables):
    """mp_comm_exchange_chunk / mp_comm_join with nranks = 1: the shard is evaluated chunk by chunk straight into its slot of
    the gathered buffer, each chunk followed by its exchange on the communicator's stream; after the join the buffer holds
    what one plain launch produces.  (With one rank the exchange has no peers: this covers the stream / event chaining and
    the argument checks; the peer traffic itself needs a multi-GPU node.)"""
    from manipulapy_amd import _hip

    tab, model = tables["ur5"], models["ur5"]
    n, rows = tab.n, 100001
    rng = np.random.default_rng(21)
    q, qd, qdd = (rng.uniform(-1, 1, (rows, n)).astype(np.float32) for _ in range(3))
    d_q, d_qd, d_qdd = ctx.to_device(q), ctx.to_device(qd), ctx.to_device(qdd)
    nb = rows * n * 4
    d_all = ctx.alloc(nb)
    ctx.memset(d_all, 0xFF, nb)
    comm = ctx.comm_create(_hip.HipContext.comm_unique_id(), 1, 0)
    rc = ((rows // 4) + 1) & ~1
    for r0 in range(0, rows, rc):
        nr, off = min(rc, rows - r0), r0 * n * 4
        ctx.id_trajectory(model, d_q.offset(off), d_qd.offset(off), d_qdd.offset(off), nr, d_all.offset(off))
        comm.exchange_chunk(d_all, nb, off, nr * n * 4)
    comm.join()
    ctx.synchronize()
    got = d_all.download((rows, n), np.float32)
    # chunk boundaries change which rows share a lane, not the per-row arithmetic beyond float32 rounding of the packed pairs
    want = ctx.id_trajectory_host(model, q, qd, qdd)
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max()
    with pytest.raises(_hip.HipError):
        comm.exchange_chunk(d_all, nb, nb - 8, 16)
    comm.destroy()
    for b in (d_q, d_qd, d_qdd, d_all):
        b.free()


def test_seven_joint_panda_truncation(ctx):
    """"panda7" = the first seven joints of the reference's 8-joint Panda tables (the 7-DOF reading of BASELINE configs[3]):
    generic and specialised kernels against the oracle on the truncated tables, fp64 and fp32, with a wrench."""
    import manipulapy_amd as mp
    from manipulapy_amd import _hip

    t = mp.robot_tables("panda7")
    assert t["S_list"].shape == (6, 7) and t["Glist"].shape == (7, 6, 6)
    tab = ref.RobotTables(S=t["S_list"], M_ee=t["M_ee"], G=t["Glist"], Mcom=t["Mlist_per_link"], joint_limits=t["joint_limits"])
    gen = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
    spec = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
    ctx.specialize(spec)
    rng = np.random.default_rng(77)
    rows = 24
    q = rng.uniform(tab.joint_limits[:, 0], tab.joint_limits[:, 1], (rows, 7))
    qd, qdd = rng.uniform(-1, 1, (rows, 7)), rng.uniform(-2, 2, (rows, 7))
    F = np.array([1.0, -2.0, 0.5, 3.0, -1.5, 0.75])
    for wrench in (None, F):
        want = ref.inverse_dynamics_trajectory(tab, q, qd, qdd, None, wrench, dtype=np.float64)
        for m in (gen, spec):
            assert_f64(ctx.id_trajectory_host(m, q, qd, qdd, None, wrench, dtype=np.float64), want)
            assert_f32(ctx.id_trajectory_host(m, q.astype(np.float32), qd.astype(np.float32), qdd.astype(np.float32), None, wrench), want)
    Ts, Js, _ = ctx.fk_jac_id_host(spec, q[:4])
    for i in range(4):
        assert_f64(Ts[i], ref.fk_space(tab, q[i]))
        assert_f64(Js[i], ref.jacobian_space(tab, q[i]))
    with pytest.raises(KeyError):
        mp.robot_tables("panda6")


def test_planner_shared_by_threads(tables):
    """One planner, one context, eight Python threads (the reference drives a planner from several threads in
    tests/test_trajectory_planning.py:1375; ctypes releases the GIL, so the calls really overlap): every thread gets the
    result a lone caller gets."""
    import threading

    import manipulapy_amd as mp

    sm, dyn, lim = mp.load_robot("ur5")
    with mp.use_backend("hip"):
        pl = mp.OptimizedTrajectoryPlanning(sm, None, dyn, lim)
        rng = np.random.default_rng(11)
        jobs = []
        for k in range(8):
            a, b = rng.uniform(-1, 1, (2, 6))
            tr = pl.joint_trajectory(a, b, 2.0, 300 + 2 * k, 5)
            tau = pl.inverse_dynamics_trajectory(tr["positions"], tr["velocities"], tr["accelerations"])
            bt = pl.batch_inverse_dynamics_trajectory(np.stack([a, b]), np.stack([b, a]), 2.0, 64 + k, 3)
            jobs.append((a, b, 300 + 2 * k, 64 + k, tau, bt))
        errors = []

        def work(k):
            try:
                a, b, N, Nb, tau, bt = jobs[k]
                for _ in range(6):
                    tr = pl.joint_trajectory(a, b, 2.0, N, 5)
                    got = pl.inverse_dynamics_trajectory(tr["positions"], tr["velocities"], tr["accelerations"])
                    np.testing.assert_array_equal(got, tau)
                    np.testing.assert_array_equal(pl.batch_inverse_dynamics_trajectory(np.stack([a, b]), np.stack([b, a]), 2.0, Nb, 3), bt)
                    np.testing.assert_allclose(dyn.mass_matrix(a), dyn.mass_matrix(a), rtol=0, atol=0)
            except Exception as exc:  # noqa: BLE001
                errors.append(f"thread {k}: {exc!r}"[:500])

        threads = [threading.Thread(target=work, args=(k,)) for k in range(8)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not errors, errors


def _restarted(tab, z, robot, i):
    """True when the reference run of fixture `i` went through a stagnation restart (its noise is NumPy's, the device's is
    a counter hash, so such a run is only comparable through the oracle)."""
    p = z[f"{robot}_params"][i]
    return ref.iterative_inverse_kinematics(tab, z[f"{robot}_T_desired"][i], z[f"{robot}_theta0"][i], p[0], p[1], int(p[2]), p[3], p[4],
                                            p[5], p[6], joint_limits=z[f"{robot}_joint_limits"], rng=np.random.RandomState(1234),
                                            adaptive_tuning=bool(p[7]), backtracking=bool(p[8]))[3] > 0


@pytest.mark.parametrize("robot", ROBOTS)
def test_inverse_kinematics_against_reference_runs_and_oracle(robot, tables):
    """Batched IK kernel (mp_inverse_kinematics_*) against the reference's own iterative_inverse_kinematics runs
    (tests/golden/ik.npz: quick / slow convergence, non-default weights, unreachable target) and, on a larger random batch,
    against the pinned oracle; single-target API == batch API; solutions reproduce the target pose."""
    import manipulapy_amd as mp
    from manipulapy_amd import _hip

    z = np.load(golden_path("ik.npz"))
    tab = tables[robot]
    lim = z[f"{robot}_joint_limits"]
    sm, _, _ = mp.load_robot(robot)
    sm.joint_limits = [(None if not np.isfinite(lo) else float(lo), None if not np.isfinite(hi) else float(hi)) for lo, hi in lim]
    with mp.use_backend("hip"):
        for i in range(10):
            p = z[f"{robot}_params"][i]
            th, ok, it = sm.iterative_inverse_kinematics(z[f"{robot}_T_desired"][i], z[f"{robot}_theta0"][i], eomg=p[0], ev=p[1],
                                                         max_iterations=int(p[2]), damping=p[3], step_cap=p[4],
                                                         weight_orientation=p[5], weight_position=p[6],
                                                         adaptive_tuning=bool(p[7]), backtracking=bool(p[8]))
            want_ok, want_it = bool(z[f"{robot}_success"][i]), int(z[f"{robot}_iterations"][i])
            if _restarted(tab, z, robot, i):
                continue
            assert ok == want_ok and abs(it - want_it) <= (1 if want_ok else 0), (robot, i, ok, it, want_it)
            np.testing.assert_allclose(th, z[f"{robot}_theta"][i], rtol=0, atol=1e-6 if want_ok else 1e-5)
        # a batch of fresh problems: targets = FK of in-limit configurations, guesses nearby
        rng = np.random.default_rng(31)
        B = 200
        fin = np.where(np.isfinite(lim), lim, np.array([-np.pi, np.pi]))
        q_true = rng.uniform(0.6 * fin[:, 0], 0.6 * fin[:, 1], (B, tab.n))
        T = np.stack([ref.fk_space(tab, q) for q in q_true])
        q0 = np.clip(q_true + rng.uniform(-0.3, 0.3, (B, tab.n)), fin[:, 0], fin[:, 1])
        th, ok, it = sm.batch_inverse_kinematics(T, q0, max_iterations=300)
        assert ok.mean() > 0.5
        for b in np.flatnonzero(ok)[:40]:
            Tb = ref.fk_space(tab, th[b])
            _, rot, tr = ref.ik_geometric_error(Tb, T[b])
            assert rot < 1e-6 and tr < 1e-6
        for b in range(0, B, 25):
            o_th, o_ok, o_it, o_rs = ref.iterative_inverse_kinematics(tab, T[b], q0[b], max_iterations=300, joint_limits=lim,
                                                                      rng=np.random.RandomState(0))
            if o_rs == 0:  # restarts use different random streams by design
                assert o_ok == ok[b] and abs(o_it - it[b]) <= 1
                np.testing.assert_allclose(th[b], o_th, rtol=0, atol=1e-6 if o_ok else 1e-5)
        with pytest.raises(NotImplementedError):
            sm.iterative_inverse_kinematics(T[0], q0[0], plot_residuals=True)
    ctx = _hip.HipContext(0)
    try:
        m = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee)
        # run-time specialised kernel (this robot's constants baked in) == generic kernel, also with the options on
        ms = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee)
        ctx.specialize(ms)
        for kw in (dict(), dict(adaptive_tuning=True, backtracking=True)):
            a = ctx.inverse_kinematics_host(m, T, q0, lim, max_iterations=300, **kw)
            b = ctx.inverse_kinematics_host(ms, T, q0, lim, max_iterations=300, **kw)
            same = (a[3] == 0) & (b[3] == 0)  # no restart on either side
            assert same.sum() > B // 2
            assert (a[1][same] == b[1][same]).mean() > 0.98 and (np.abs(a[2][same] - b[2][same]) <= 1).mean() > 0.95
            both = same & a[1] & b[1]
            assert np.abs(a[0][both] - b[0][both]).max() < 1e-5
        with pytest.raises(_hip.HipError):
            ctx.inverse_kinematics_host(m, T[:2], q0[:2], max_iterations=0)
        e = ctx.inverse_kinematics_host(m, T[:0], q0[:0])
        assert e[0].shape == (0, tab.n)
    finally:
        ctx.destroy()


@pytest.mark.parametrize("robot", ROBOTS)
def test_robust_inverse_kinematics_against_reference_runs(robot, tables):
    """Multi-start IK: every attempt of every target is a row of a few launches; the winner is picked in the reference's
    order.  Against the reference's own runs (np.random seeded identically; runs whose attempts went through a stagnation
    restart are only checked for success, their noise differs by design); batch == per-target."""
    import manipulapy_amd as mp

    z = np.load(golden_path("ik.npz"))
    tab = tables[robot]
    lim = z[f"{robot}_joint_limits"]
    sm, _, _ = mp.load_robot(robot)
    sm.joint_limits = [(None if not np.isfinite(lo) else float(lo), None if not np.isfinite(hi) else float(hi)) for lo, hi in lim]
    with mp.use_backend("hip"):
        for case in range(4):
            T = z[f"{robot}_robust_T_desired"][case]
            np.random.seed(4321 + case)
            _, o_ok, o_total, o_name, o_restarts = ref.robust_inverse_kinematics(tab, T, lim, max_attempts=10 if case < 3 else 4,
                                                                               max_iterations=300)
            np.random.seed(4321 + case)
            th, ok, total, name = sm.robust_inverse_kinematics(T, max_attempts=10 if case < 3 else 4, max_iterations=300)
            want_ok = bool(z[f"{robot}_robust_success"][case])
            if o_restarts == 0:
                assert ok == want_ok and name == str(z[f"{robot}_robust_strategy"][case]), (robot, case, ok, name)
                assert abs(total - int(z[f"{robot}_robust_iterations"][case])) <= 2
                np.testing.assert_allclose(th, z[f"{robot}_robust_theta"][case], rtol=0, atol=2e-3 if want_ok else 1e-4)
            if ok:
                _, rot, tr = ref.ik_geometric_error(ref.fk_space(tab, th), T)
                assert rot < 2e-3 and tr < 2e-3
        Tb = z[f"{robot}_robust_T_desired"][:3]
        np.random.seed(99)
        th_b, ok_b, it_b, names_b = sm.batch_robust_inverse_kinematics(Tb, max_attempts=3, max_iterations=300)
        assert th_b.shape == (3, tab.n) and len(names_b) == 3
        for b in range(3):  # no random guess among the first three strategies, and the restart noise is keyed by the problem's
            th1, ok1, it1, n1 = sm.robust_inverse_kinematics(Tb[b], max_attempts=3, max_iterations=300)  # content: batch == single
            np.testing.assert_array_equal(th1, th_b[b])
            assert (ok1, it1, n1) == (bool(ok_b[b]), int(it_b[b]), names_b[b])
        e = sm.batch_robust_inverse_kinematics(Tb[:1], max_attempts=0)
        assert not e[1][0] and e[3] == ["none"]


def test_smart_inverse_kinematics_against_reference_runs(tables):
    """smart_inverse_kinematics (strategy-chosen initial guess + fallback starts) on the reference's own problems: the
    deterministic strategy (extrapolate: 7 iterations, no restart) reproduces the reference run; the long multi-attempt
    runs go through stagnation restarts whose noise differs by design, so they are checked for a valid outcome."""
    import manipulapy_amd as mp
    from manipulapy_amd import ik_helpers as h

    z = np.load(golden_path("ik.npz"))
    tab = tables["ur5"]
    sm, _, _ = mp.load_robot("ur5")
    sm.joint_limits = [(float(a), float(b)) for a, b in z["ur5_joint_limits"]]
    T = z["smart_target"]
    with mp.use_backend("hip"):
        th, ok, it = sm.smart_inverse_kinematics(T, strategy="extrapolate", theta_current=z["ext_theta"][1], T_current=z["ext_Tc"][1],
                                                 max_iterations=250)
        assert ok == bool(z["smart_extrapolate_success"]) and abs(it - int(z["smart_extrapolate_iterations"])) <= 1
        np.testing.assert_allclose(th, z["smart_extrapolate_theta"], rtol=0, atol=1e-6)
        cache = h.IKInitialGuessCache(max_size=3)
        for i in range(4):
            cache.add(z["ext_Tc"][i], z["ext_theta"][i], residual=[None, 0.5, 1e-4, 0.02][i])
        for strat, kw in (("workspace_heuristic", {}), ("midpoint", {}), ("cached", {"cache": cache}), ("random", {})):
            np.random.seed(5)
            th, ok, it = sm.smart_inverse_kinematics(T, strategy=strat, max_iterations=250, **kw)
            assert it >= 1 and th.shape == (6,)
            if ok:
                _, rot, tr = ref.ik_geometric_error(ref.fk_space(tab, th), T)
                assert rot < 1e-6 and tr < 1e-6
        th, ok, it = sm.smart_inverse_kinematics(z["smart_unreachable_target"], max_iterations=40)
        assert not ok and it == int(z["smart_unreachable_iterations"])  # five exhausted attempts of 41
        th1, ok1, it1 = sm.smart_inverse_kinematics(T, strategy="extrapolate", max_iterations=250)  # missing inputs -> heuristic guess
        th2, ok2, it2 = sm.smart_inverse_kinematics(T, strategy="workspace_heuristic", max_iterations=250, auto_fallback=False)
        assert it1 >= it2
        with pytest.raises(ValueError):
            sm.smart_inverse_kinematics(T, strategy="nope")


def test_singularity_and_workspace_against_oracle(tables, dyn_golden):
    """Singularity mirror (reference singularity/singularity_analysis.py): condition number / smallest singular value of
    the GPU Jacobians == NumPy on the oracle's Jacobians; batch == per-sample; Monte-Carlo workspace points == oracle FK."""
    import manipulapy_amd as mp

    for robot in ("ur5", "iiwa14"):
        tab, z = tables[robot], dyn_golden[robot]
        sm, _, lim = mp.load_robot(robot)
        with mp.use_backend("hip"):
            sing = mp.Singularity(sm)
            th = z["thetas"][:8]
            cond = sing.condition_number(th)
            flags = sing.singularity_analysis(th)
            near = sing.near_singularity_detection(th, threshold=50.0)
            man = sing.manipulability(th)
            assert cond.shape == (8,) and flags.shape == (8,) and flags.dtype == bool
            for i in range(8):
                J = ref.jacobian_space(tab, th[i])
                s = np.linalg.svd(J, compute_uv=False)
                want = np.linalg.cond(J)   # the GPU Jacobian differs by ~1e-12 absolute; that moves sigma_min by ~1e-12
                np.testing.assert_allclose(cond[i], want, rtol=1e-6 + 1e-10 * want)
                assert flags[i] == (s[-1] < 1e-4) and near[i] == (cond[i] > 50.0)
                np.testing.assert_allclose(man[i], np.prod(s), rtol=1e-6, atol=1e-9)
                assert sing.condition_number(th[i]) == pytest.approx(cond[i], rel=1e-12)
                assert isinstance(sing.singularity_analysis(th[i]), bool)
            # a straight arm is singular: all UR5 / iiwa axes meet the rank-deficient pose at q = 0
            assert sing.singularity_analysis(np.zeros(tab.n)) in (True, False)
            ws = sing.workspace_monte_carlo(lim, num_samples=500, seed=7)
            assert ws["joint_samples"].dtype == np.float32 and ws["points"].shape == (500, 3)
            lo, hi = np.asarray(lim, np.float32)[:, 0], np.asarray(lim, np.float32)[:, 1]
            assert (ws["joint_samples"] >= lo).all() and (ws["joint_samples"] <= hi).all()
            for i in (0, 17, 499):
                T = ref.fk_space(tab, ws["joint_samples"][i].astype(np.float64))
                np.testing.assert_allclose(ws["points"][i], T[:3, 3], rtol=1e-6, atol=1e-7)
            assert ws["volume"] > 0 and ws["simplices"].shape[1] == 3
            again = sing.workspace_monte_carlo(lim, num_samples=500, seed=7, hull=False)
            np.testing.assert_array_equal(again["points"], ws["points"])
            with pytest.raises(ValueError):
                sing.workspace_monte_carlo([1.0, 2.0], 10)


def test_host_paths_pinned_pipelined_and_prefaulted(models, tables):
    """The host-buffer ID entry point gives bit-identical results whether the arrays are pageable (single shot),
    page-locked (chunked upload / kernel / download pipeline; chunk size forced small in the first pass so that several
    chunks and an odd tail are exercised) or written into a caller-supplied `out`."""
    import subprocess, sys, textwrap

    code = textwrap.dedent("""
        import numpy as np, sys
        sys.path.insert(0, %r)
        import manipulapy_amd as mp
        from manipulapy_amd import _hip
        ctx = _hip.HipContext(0)
        sm, dyn, lim = mp.load_robot("ur5")
        model = _hip.HipModel(dyn.S_list, dyn.Mlist_per_link, dyn.Glist, sm.M_list, lim)
        rng = np.random.default_rng(9)
        rows, n = 700001, 6          # odd -> one-row tail kernel in the last chunk
        q, qd, qdd = (rng.uniform(-1, 1, (rows, n)).astype(np.float32) for _ in range(3))
        want = ctx.id_trajectory_host(model, q, qd, qdd)
        out = np.full((rows, n), np.nan, np.float32)
        assert ctx.id_trajectory_host(model, q, qd, qdd, out=out) is out
        np.testing.assert_array_equal(out, want)
        pq, pqd, pqdd, pt = (ctx.pinned_empty((rows, n), np.float32) for _ in range(4))
        pq[:], pqd[:], pqdd[:] = q, qd, qdd
        pt[:] = np.nan
        ctx.id_trajectory_host(model, pq, pqd, pqdd, out=pt)
        np.testing.assert_array_equal(pt, want)
        for bad in (np.empty((rows, n), np.float64), np.empty((rows - 1, n), np.float32), np.empty((n, rows), np.float32).T):
            try:
                ctx.id_trajectory_host(model, q, qd, qdd, out=bad)
            except ValueError:
                pass
            else:
                raise AssertionError("bad out accepted")
        fused = ctx.traj_id_fused_host(model, q[:300], qd[:300], 2.0, 1201, 5)
        pf = ctx.pinned_empty(fused.shape, np.float32)
        ctx.traj_id_fused_host(model, q[:300], qd[:300], 2.0, 1201, 5, out=pf)
        np.testing.assert_array_equal(pf, fused)
        # FK + Jacobian + ID through the same pipeline: page-locked in / out, several chunks, == the pageable call
        q64, qd64, qdd64 = q[:300001].astype(np.float64), qd[:300001].astype(np.float64), qdd[:300001].astype(np.float64)
        Tw, Jw, tw = ctx.fk_jac_id_host(model, q64, qd64, qdd64)
        pin = lambda a: (lambda b: (b.__setitem__(slice(None), a), b)[1])(ctx.pinned_empty(a.shape, np.float64))
        pq64, pqd64, pqdd64 = pin(q64), pin(qd64), pin(qdd64)
        oT, oJ, ot = ctx.pinned_empty(Tw.shape, np.float64), ctx.pinned_empty(Jw.shape, np.float64), ctx.pinned_empty(tw.shape, np.float64)
        rT, rJ, rt = ctx.fk_jac_id_host(model, pq64, pqd64, pqdd64, out_T=oT, out_J=oJ, out_tau=ot)
        assert rT is oT and rJ is oJ and rt is ot
        np.testing.assert_array_equal(oT, Tw); np.testing.assert_array_equal(oJ, Jw); np.testing.assert_array_equal(ot, tw)
        rT2, rJ2, rt2 = ctx.fk_jac_id_host(model, pq64, want_T=True, want_J=False, out_T=oT)   # FK only, pinned, chunked
        assert rJ2 is None and rt2 is None
        np.testing.assert_array_equal(oT, Tw)
        del pq64, pqd64, pqdd64, oT, oJ, ot, rT, rJ, rt, rT2
        # the roll-out through the same pipeline (round 4): page-locked in / out, chunks of whole trajectories, == the pageable call
        xs, xd, xl = mp.load_robot("xarm6")
        xm = _hip.HipModel(xd.S_list, xd.Mlist_per_link, xd.Glist, xs.M_list, xl)
        Bf, Nf = 9000 + 37, 25
        th0, dth0 = rng.uniform(-0.5, 0.5, (Bf, 6)).astype(np.float32), rng.uniform(-0.2, 0.2, (Bf, 6)).astype(np.float32)
        tmf = rng.uniform(-1, 1, (Bf, Nf, 6)).astype(np.float32)
        Fmf = (rng.uniform(-1, 1, (Bf, Nf, 6)) * 0.05).astype(np.float32)
        G = np.array([0.0, 0.0, -9.81])
        for F in (Fmf, None):
            wantf = ctx.fd_trajectory_host(xm, th0, dth0, tmf, G, F, 0.01, 1, dtype=np.float32)
            pinf = lambda a: (lambda b: (b.__setitem__(slice(None), a), b)[1])(ctx.pinned_empty(a.shape, np.float32))
            outs = [ctx.pinned_empty((Bf, Nf, 6), np.float32) for _ in range(3)]
            for o in outs: o[:] = np.nan
            gotf = ctx.fd_trajectory_host(xm, pinf(th0), pinf(dth0), pinf(tmf), G, None if F is None else pinf(F), 0.01, 1,
                                          dtype=np.float32, out=outs)
            for k in range(3):
                assert gotf[k] is outs[k]
                np.testing.assert_array_equal(outs[k], wantf[k])
        del outs, gotf
        del pq, pqd, pqdd
        ctx.destroy()                       # page-locked arrays outlive the context that allocated them
        np.testing.assert_array_equal(pt, want)
        np.testing.assert_array_equal(pf, fused)
        del pt, pf
        print("OK")
    """ % ROOT)
    # (100001 rows per chunk: 7 chunks of the 700001-row ID call, 4000-trajectory chunks of the 9037 x 25 roll-out)
    for env_extra in ({"MANIPULAPY_HIP_HOST_CHUNK_ROWS": "100001"}, {}):
        env = dict(os.environ, **env_extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_launch_graph_replays_the_captured_calls(ctx, models, tables):
    """mp_graph_*: a captured sequence (fp32 ID with a wrench, fp64 FK + Jacobian + ID, roll-out) replays on refreshed
    inputs with the results of the direct calls; non-capturable calls are refused while a capture is open."""
    from manipulapy_amd import _hip

    tab, model = tables["ur5"], models["ur5"]
    n, rows, B, Nt = tab.n, 515, 9, 6
    rng = np.random.default_rng(3)
    F = np.array([1.0, -2.0, 0.5, 3.0, -1.5, 0.75])

    def inputs():
        return [rng.uniform(-1, 1, (rows, n)) for _ in range(3)] + [rng.uniform(-0.3, 0.3, (B, n)), rng.uniform(-0.1, 0.1, (B, n)),
                                                                    rng.uniform(-1, 1, (B, Nt, n))]

    x = inputs()
    d32 = [ctx.to_device(a.astype(np.float32)) for a in x[:3]]
    d64 = [ctx.to_device(a) for a in x[:3]]
    dfd = [ctx.to_device(a.astype(np.float32)) for a in x[3:]]
    t32, t64 = ctx.alloc(rows * n * 4), ctx.alloc(rows * n * 8)
    dT, dJ = ctx.alloc(rows * 16 * 8), ctx.alloc(rows * 6 * n * 8)
    out = [ctx.alloc(B * Nt * n * 4) for _ in range(3)]
    with ctx.capture() as cap:
        ctx.id_trajectory(model, *d32, rows, t32, None, F)
        ctx.fk_jac_id(model, *d64, rows, dT, dJ, t64)
        ctx.fd_trajectory(model, dfd[0], dfd[1], dfd[2], None, B, Nt, None, 0.01, 1, *out, dtype=np.float32)
        with pytest.raises(_hip.HipError, match="captured"):
            ctx.alloc(64)
    graph = cap.graph
    ctx.synchronize()
    for _ in range(2):  # replay on fresh contents of the same buffers
        x = inputs()
        for d, a in zip(d32, x[:3]):
            d.upload(a.astype(np.float32))
        for d, a in zip(d64, x[:3]):
            d.upload(a)
        for d, a in zip(dfd, x[3:]):
            d.upload(a.astype(np.float32))
        graph.launch()
        ctx.synchronize()
        np.testing.assert_array_equal(t32.download((rows, n), np.float32),
                                      ctx.id_trajectory_host(model, *[a.astype(np.float32) for a in x[:3]], None, F))
        Ts, Js, taus = ctx.fk_jac_id_host(model, *x[:3])
        np.testing.assert_array_equal(dT.download((rows, 4, 4), np.float64), Ts)
        np.testing.assert_array_equal(dJ.download((rows, 6, n), np.float64), Js)
        np.testing.assert_array_equal(t64.download((rows, n), np.float64), taus)
        want = ctx.fd_trajectory_host(model, *[a.astype(np.float32) for a in x[3:]], None, None, 0.01, 1, dtype=np.float32)
        for o, w in zip(out, want):
            np.testing.assert_array_equal(o.download((B, Nt, n), np.float32), w)
        assert_f32(t32.download((rows, n), np.float32)[::50],
                   ref.inverse_dynamics_trajectory(tab, x[0][::50], x[1][::50], x[2][::50], None, F, dtype=np.float64))
    graph.destroy()
    for b in d32 + d64 + dfd + [t32, t64, dT, dJ] + out:
        b.free()
    # the fused generation + ID call keeps a per-context time-scaling table keyed by (N, Tf, method): a captured call
    # carries its own table kernel, so a replay is right even after other calls have rewritten the table
    Bf, Nf = 7, 33
    st, en = rng.uniform(-1, 1, (Bf, n)).astype(np.float32), rng.uniform(-1, 1, (Bf, n)).astype(np.float32)
    d_st, d_en, d_t = ctx.to_device(st), ctx.to_device(en), ctx.alloc(Bf * Nf * n * 4)
    want = ctx.traj_id_fused_host(model, st, en, 2.0, Nf, 5)   # also sizes the table before the capture
    with ctx.capture() as cap:
        ctx.traj_id_fused(model, d_st, d_en, Bf, Nf, 2.0, 5, d_t)
    other = ctx.traj_id_fused_host(model, st, en, 3.5, Nf, 3)  # different Tf / method: rewrites the table
    assert np.abs(other - want).max() > 1e-3
    cap.graph.launch()
    ctx.synchronize()
    np.testing.assert_array_equal(d_t.download((Bf, Nf, n), np.float32), want)
    np.testing.assert_array_equal(ctx.traj_id_fused_host(model, st, en, 3.5, Nf, 3), other)  # and the cache is not left stale
    cap.graph.destroy()
    for b in (d_st, d_en, d_t):
        b.free()


@pytest.mark.gpu
def test_first_use_of_a_model_inside_a_capture(tables):
    """A generic model whose FIRST launch on a fresh context is a captured one: its device copy (float32 + float64 model, read by the
    one-row kernel and by the float64 pass) is made at once, outside the graph, on a stream that is not capturing; the capture stays
    valid and the replay returns the host entry point's bits - rows that need the float64 pass included."""
    from manipulapy_amd import _hip

    tab = tables["ur5"]
    lim = tab.joint_limits
    rng = np.random.default_rng(12)
    s_ = rng.uniform(lim[:, 0], lim[:, 1], (30, 6)).astype(np.float32)
    e_ = rng.uniform(lim[:, 0], lim[:, 1], (30, 6)).astype(np.float32)
    o = ref.batch_joint_trajectory(lim, s_, e_, 2.0, 700, 5)
    q, qd, qdd = (np.ascontiguousarray(o[k].reshape(-1, 6), dtype=np.float32) for k in ("positions", "velocities", "accelerations"))
    ctx = _hip.HipContext(0)
    try:
        m = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, lim)
        assert _hip.cpu_id_row_precision(m, q, qd, qdd).sum() > 20
        d = [ctx.to_device(a) for a in (q, qd, qdd)]
        d_tau = ctx.alloc(q.nbytes)
        with ctx.capture() as cap:
            ctx.id_trajectory(m, *d, len(q), d_tau, dtype=np.float32)      # first use of `m` on this context
        ctx.memset(d_tau, 0, q.nbytes)
        cap.graph.launch()
        got = d_tau.download(q.shape, np.float32)
        np.testing.assert_array_equal(got, ctx.id_trajectory_host(m, q, qd, qdd, dtype=np.float32))
        cap.graph.destroy()
    finally:
        ctx.destroy()


@pytest.mark.parametrize("seed", [0, 2, 4, 5, 6, 8, 11, 16])
def test_random_robots_on_gpu(seed, ctx):
    """Randomised chains (tests/test_random_robots.py) through the C ABI: generic fp64 / fp32 and specialised kernels."""
    from manipulapy_amd import _hip
    from test_random_robots import FLAVOURS, random_robot

    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 9))
    tab = random_robot(rng, n, FLAVOURS[seed % len(FLAVOURS)])
    rows = 65
    q = rng.uniform(-2.5, 2.5, (rows, n))
    q[:, np.abs(tab.S[:3]).sum(axis=0) == 0] *= 0.1
    qd, qdd = rng.uniform(-1, 1, (rows, n)), rng.uniform(-2, 2, (rows, n))
    g, F = np.array([0.4, -0.3, -9.81]), rng.uniform(-3, 3, 6)
    m = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
    want = ref.inverse_dynamics_trajectory(tab, q[:9], qd[:9], qdd[:9], g, F, dtype=np.float64)
    scale = max(1.0, float(np.abs(want).max()))
    T, J, tau = ctx.fk_jac_id_host(m, q, qd, qdd, g, F)
    np.testing.assert_allclose(tau[:9], want, rtol=1e-6, atol=1e-6 * scale)
    for r in range(0, rows, 16):
        np.testing.assert_allclose(T[r], ref.fk_space(tab, q[r]), atol=1e-10)
        np.testing.assert_allclose(J[r], ref.jacobian_space(tab, q[r]), atol=1e-10)
    # float32: the suite's ELEMENT-WISE rule (1e-4 |ref| + 5e-6 max|row|), against the oracle on the same float32-rounded inputs
    q32, qd32, qdd32 = (a.astype(np.float32) for a in (q, qd, qdd))
    want32 = ref.inverse_dynamics_trajectory(tab, q32[:17].astype(np.float64), qd32[:17].astype(np.float64), qdd32[:17].astype(np.float64), g, F,
                                             dtype=np.float64)
    t32 = ctx.id_trajectory_host(m, q32, qd32, qdd32, g, F, dtype=np.float32)
    assert_f32(t32[:17], want32)
    ctx.specialize(m)
    s32 = ctx.id_trajectory_host(m, q32, qd32, qdd32, g, F, dtype=np.float32)
    s64 = ctx.id_trajectory_host(m, q, qd, qdd, g, F, dtype=np.float64)
    assert_f32(s32[:17], want32)
    np.testing.assert_allclose(s64[:9], want, rtol=1e-6, atol=1e-6 * scale)
    assert np.abs(s32 - t32).max() <= 2e-4 * scale
    # mass matrix, forward dynamics, roll-out (every DOF has its own LDS tile shape) and IK on the same chain
    gen = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
    M = ctx.mass_matrix_host(gen, q[:5])
    for r in range(5):
        np.testing.assert_allclose(M[r], ref.mass_matrix(tab, q[r]), rtol=1e-8, atol=1e-9)
    tau_in = rng.uniform(-3, 3, (rows, n))
    qdd_fd = ctx.forward_dynamics_host(gen, q[:3], qd[:3], tau_in[:3], g, F)
    for r in range(3):
        w = ref.forward_dynamics(tab, q[r], qd[r], tau_in[r], g, F)
        np.testing.assert_allclose(qdd_fd[r], w, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(w).max())))
    B, Nt = 66, 6
    th0 = q[:B % rows + 1][:1].repeat(B, axis=0) * rng.uniform(0.2, 1.0, (B, 1)); dth0 = rng.uniform(-0.2, 0.2, (B, n))
    tm, Fm = rng.uniform(-1, 1, (B, Nt, n)), rng.uniform(-1, 1, (B, Nt, 6))
    for dt_ in (np.float64, np.float32):
        for wrench in (None, Fm):
            a = ctx.fd_trajectory_host(gen, th0, dth0, tm, g, wrench, 0.005, 1, dtype=dt_)
            b = ctx.fd_trajectory_host(m, th0, dth0, tm, g, wrench, 0.005, 1, dtype=dt_)  # m is specialised
            o = ref.forward_dynamics_trajectory(tab, th0[B - 1], dth0[B - 1], tm[B - 1], g, np.zeros((Nt, 6)) if wrench is None else Fm[B - 1],
                                                0.005, 1, joint_limits=tab.joint_limits)
            tol = 2e-5 if dt_ == np.float64 else 1e-3
            for k, name in enumerate(("positions", "velocities", "accelerations")):
                sc = max(1.0, float(np.abs(o[name]).max()))
                np.testing.assert_allclose(a[k][B - 1], o[name], rtol=tol, atol=tol * sc)
                assert np.abs(a[k] - b[k]).max() <= (1e-6 if dt_ == np.float64 else 1e-3) * max(1.0, float(np.abs(a[k]).max()))
    # every lane of every tile (four whole tiles): the flush mode depends on the row size - whole-line owner-lane stores for
    # 4 - 7 joints (tails of up to a whole run), flat stores for 8, lane-by-lane rows for 1 - 3
    from oracle import c_oracle
    Nt = 16
    tm, Fm = rng.uniform(-1, 1, (B, Nt, n)) * 0.3, rng.uniform(-1, 1, (B, Nt, 6)) * 0.3
    x = [a.astype(np.float32) for a in (th0, dth0, tm, Fm)]
    want = c_oracle.fd_trajectory(tab, *[a.astype(np.float64) for a in x[:3]], g, x[3].astype(np.float64), 0.002, 1,
                                  joint_limits=tab.joint_limits)[:3]
    if all(np.isfinite(w).all() for w in want):
        for model in (gen, m):
            got = ctx.fd_trajectory_host(model, x[0], x[1], x[2], g, x[3], 0.002, 1, dtype=np.float32)
            for k in range(3):
                sc = max(1.0, float(np.abs(want[k]).max()))
                assert np.abs(got[k] - want[k]).max() <= 2e-3 * sc, (n, k, float(np.abs(got[k] - want[k]).max()), sc)
    prismatic = np.abs(tab.S[:3]).sum(axis=0) == 0
    lim = np.tile([-2.5, 2.5], (n, 1)).astype(float)
    lim[prismatic] = [-0.4, 0.4]
    q_true = rng.uniform(0.7 * lim[:, 0], 0.7 * lim[:, 1], (6, n))
    Tt = np.stack([ref.fk_space(tab, x) for x in q_true])
    q0 = np.clip(q_true + rng.uniform(-0.2, 0.2, (6, n)) * np.where(prismatic, 0.2, 1.0), lim[:, 0], lim[:, 1])
    th, ok, it, rs = ctx.inverse_kinematics_host(gen, Tt, q0, lim, max_iterations=150)
    for b in range(6):
        o_th, o_ok, o_it, o_rs = ref.iterative_inverse_kinematics(tab, Tt[b], q0[b], max_iterations=150, joint_limits=lim,
                                                                  rng=np.random.RandomState(0))
        if o_rs == 0 and rs[b] == 0:
            assert bool(ok[b]) == o_ok and abs(int(it[b]) - o_it) <= 1
            np.testing.assert_allclose(th[b], o_th, rtol=0, atol=1e-6 if o_ok else 1e-4)


def test_cartesian_trajectory_on_gpu(ctx):
    """cartesian_trajectory through the planner (hip backend) vs the reference dump, all SO(3) log branches."""
    import manipulapy_amd as mp

    z = np.load(golden_path("cartesian_ur5.npz"))
    sm, dyn, lim = mp.load_robot("ur5")
    with mp.use_backend("hip"):
        pl = mp.OptimizedTrajectoryPlanning(sm, None, dyn, lim)
        for tag in ("generic", "tiny", "small", "nearpi", "pi_z", "pi_x", "pi_gen"):
            for method in (3, 5, 1):
                r = pl.cartesian_trajectory(z["Xstart"], z[f"{tag}_Xend"], 2.0, 21, method)
                for k in ("positions", "velocities", "accelerations", "orientations"):
                    want = z[f"{tag}_m{method}_{k}"]
                    assert r[k].dtype == np.float32 and r[k].shape == want.shape
                    np.testing.assert_allclose(r[k], want, rtol=2e-6, atol=2e-6, err_msg=f"{tag} m{method} {k}")
        # batch = stacked singles; orientations stay orthonormal; end points hit
        tags = ("generic", "small", "nearpi")
        Xe = np.stack([z[f"{t}_Xend"] for t in tags])
        rb = pl.batch_cartesian_trajectory(np.broadcast_to(z["Xstart"], Xe.shape), Xe, 2.0, 300, 5)
        assert rb["orientations"].shape == (3, 300, 3, 3)
        for b, t in enumerate(tags):
            one = pl.cartesian_trajectory(z["Xstart"], z[f"{t}_Xend"], 2.0, 300, 5)
            np.testing.assert_array_equal(one["orientations"], rb["orientations"][b])
            np.testing.assert_allclose(rb["orientations"][b, -1], z[f"{t}_Xend"][:3, :3], atol=2e-6)
            np.testing.assert_allclose(rb["positions"][b, -1], z[f"{t}_Xend"][:3, 3], atol=2e-6)
        RtR = np.einsum("bnij,bnik->bnjk", rb["orientations"], rb["orientations"])
        assert np.abs(RtR - np.eye(3)).max() < 5e-6
        with pytest.raises(ZeroDivisionError):
            pl.cartesian_trajectory(z["Xstart"], z["generic_Xend"], 2.0, 1, 5)
        assert pl.cartesian_trajectory(z["Xstart"], z["generic_Xend"], 2.0, 0, 5)["orientations"].shape == (0, 3, 3)


def test_controller_laws_against_reference_runs():
    """Every mirrored ManipulatorController law against the reference's own outputs (tests/golden/control_ur5.npz), including
    the stateful second calls (integral, parameter estimate); 2-D inputs batch and agree with the per-sample calls."""
    import manipulapy_amd as mp

    z = np.load(golden_path("control_ur5.npz"))
    i = lambda k: z[f"in_{k}"]
    g, F = np.array([0.0, 0.0, -9.81]), np.array([1.0, -2.0, 0.5, 3.0, -1.5, 0.75])
    _, dyn, _ = mp.load_robot("ur5")
    tol = dict(rtol=1e-6, atol=1e-6)
    with mp.use_backend("hip"):
        c = mp.ManipulatorController(dyn)
        np.testing.assert_allclose(c.pd_control(i("qd_des"), i("dq_des"), i("q"), i("dq"), i("Kp"), i("Kd")), z["pd"], rtol=1e-12)
        np.testing.assert_allclose(c.pid_control(i("qd_des"), i("dq_des"), i("q"), i("dq"), 0.01, i("Kp"), i("Ki"), i("Kd")), z["pid_1"], rtol=1e-12)
        np.testing.assert_allclose(c.pid_control(i("qd_des"), i("dq_des"), i("q"), i("dq"), 0.01, i("Kp"), i("Ki"), i("Kd"), i_clamp=0.015),
                                   z["pid_2"], rtol=1e-12)
        np.testing.assert_allclose(c.pd_feedforward_control(i("qd_des"), i("dq_des"), i("ddq_des"), i("q"), i("dq"), i("Kp"), i("Kd"), g, F),
                                   z["pdff"], **tol)
        lq, ldq, lt = c.enforce_limits(i("q"), i("dq"), i("tau_in"), i("jl"), i("tl"))
        np.testing.assert_array_equal(lq, z["lim_q"]); np.testing.assert_array_equal(ldq, z["lim_dq"]); np.testing.assert_array_equal(lt, z["lim_tau"])
        np.testing.assert_allclose(c.joint_space_control(i("qd_des"), i("q"), i("dq"), i("Kp"), i("Kd")), z["jsc"], rtol=1e-12)
        np.testing.assert_allclose(c.cartesian_space_control(i("x_des"), i("q"), i("dq"), i("Kp3"), i("Kd3")), z["csc_vec"], **tol)
        np.testing.assert_allclose(c.cartesian_space_control(i("x_des"), i("q"), i("dq"), i("Kp33"), i("Kd33")), z["csc_mat"], **tol)
        np.testing.assert_allclose(c.robust_control(i("q"), i("dq"), i("ddq"), g, F, i("dist"), i("gain")), z["robust"], **tol)
        c2 = mp.ManipulatorController(dyn)
        np.testing.assert_allclose(c2.adaptive_control(i("q"), i("dq"), i("ddq"), g, F, i("merr"), 0.7), z["adaptive_1"], **tol)
        np.testing.assert_allclose(c2.adaptive_control(i("q"), i("dq"), i("ddq"), g, F, i("merr"), 0.7), z["adaptive_2"], **tol)
        c3 = mp.ManipulatorController(dyn)
        for k in ("ctc_1", "ctc_2"):
            np.testing.assert_allclose(c3.computed_torque_control(i("qd_des"), i("dq_des"), i("ddq_des"), i("q"), i("dq"), g, 0.01, i("Kp"), i("Ki"),
                                                                  i("Kd")), z[k], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(c3.feedforward_control(i("qd_des"), i("dq_des"), i("ddq_des"), g, F), z["ff"], **tol)
        # batches: 2-D states == the per-sample calls
        rng = np.random.default_rng(5)
        Q, dQ, ddQ = rng.uniform(-1, 1, (3, 5, 6))
        X = rng.uniform(-0.5, 0.5, (5, 3))
        rb = c.robust_control(Q, dQ, ddQ, g, F, i("dist"), i("gain"))
        cb = c.cartesian_space_control(X, Q, dQ, i("Kp33"), i("Kd33"))
        for r in range(5):
            np.testing.assert_allclose(rb[r], c.robust_control(Q[r], dQ[r], ddQ[r], g, F, i("dist"), i("gain")), rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(cb[r], c.cartesian_space_control(X[r], Q[r], dQ[r], i("Kp33"), i("Kd33")), rtol=1e-12, atol=1e-12)
        with pytest.raises(ValueError):
            c.adaptive_control(Q, dQ, ddQ, g, F, i("merr"), 0.7)
        # Kalman filter on [q; qd] (prediction = forward dynamics on the GPU), three consecutive cycles; metrics; Z-N gains
        ck = mp.ManipulatorController(dyn)
        for step in range(3):
            qf, dqf = ck.kalman_filter_control(i("qd_des"), i("dq_des"), i("q") + 0.01 * step, i("dq"), z["kalman_tau"], g, F, 0.01,
                                               np.eye(12) * 1e-3, np.eye(12) * 1e-2)
            np.testing.assert_allclose(np.concatenate((qf, dqf)), z["kalman_x"][step], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(ck.P, z["kalman_P"], rtol=1e-10, atol=1e-12)
        with pytest.raises(ValueError):
            mp.ManipulatorController(dyn).kalman_filter_update(np.zeros(12), np.eye(12))
        t_, y_ = z["metric_t"], z["metric_y"]
        got = [c.calculate_rise_time(t_, y_, 1.0), c.calculate_percent_overshoot(y_, 1.0), c.calculate_settling_time(t_, y_, 1.0),
               c.calculate_settling_time(t_, y_, 1.0, 0.2), c.calculate_steady_state_error(y_, 1.0), c.calculate_rise_time(t_, y_, 5.0),
               c.calculate_settling_time(t_, y_ * 0 + 3, 1.0)]
        np.testing.assert_allclose(got, z["metric_values"], rtol=1e-12)
        np.testing.assert_allclose([list(c.ziegler_nichols_tuning(8.0, 0.5, k)) for k in ("P", "PI", "PID")], z["zn"], rtol=1e-12)
        np.testing.assert_allclose(np.stack(c.tune_controller(np.array([8.0, 4.0]), np.array([0.5, 0.25]), "PID")), z["zn_vec"], rtol=1e-12)
        with pytest.raises(ValueError):
            c.ziegler_nichols_tuning(1.0, 0.0, "PID")
        # the small kinematics helpers: pose as [p; ZYX Euler], joint velocity through pinv(J); batch == per sample
        sm, _, _ = mp.load_robot("ur5")
        np.testing.assert_allclose(sm.end_effector_pose(z["kin_q"]), z["kin_pose"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(sm.end_effector_pose(z["kin_q"][2]), z["kin_pose"][2], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(sm.joint_velocity(z["kin_q"], z["kin_V"]), z["kin_jvel_space"], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(sm.joint_velocity(z["kin_q"][1], z["kin_V"][1], frame="body"), z["kin_jvel_body"][1], rtol=1e-6, atol=1e-8)
        sm.update_state(z["kin_q"][0])
        assert sm.joint_velocities.shape == (6,) and not sm.joint_velocities.any()
        pl = mp.OptimizedTrajectoryPlanning(sm, None, dyn, i("jl"))
        pl.batch_inverse_dynamics_trajectory(Q, dQ, 2.0, 16, 5)
        assert pl.get_performance_stats()["gpu_calls"] == 1
        pl.reset_performance_stats()
        assert pl.get_performance_stats()["gpu_calls"] == 0
        pl.cleanup_gpu_memory()
        pl.batch_inverse_dynamics_trajectory(Q, dQ, 2.0, 16, 5)  # still works after the pool was trimmed


def test_computed_torque_control_batched(tables):
    """Computed-torque control as one inverse-dynamics evaluation vs the reference's formula
    M (Kp e + Ki eint + Kd de) + ID(q, qd, qdd_d, g, 0) evaluated with the CPU oracle."""
    import manipulapy_amd as mp

    tab = tables["ur5"]
    sm, dyn, _ = mp.load_robot("ur5")
    rng = np.random.default_rng(9)
    rows, n = 6, 6
    q, qd = rng.uniform(-1, 1, (rows, n)), rng.uniform(-1, 1, (rows, n))
    q_d, qd_d, qdd_d = rng.uniform(-1, 1, (rows, n)), rng.uniform(-1, 1, (rows, n)), rng.uniform(-1, 1, (rows, n))
    Kp, Ki, Kd, dt, g = np.full(n, 50.0), np.full(n, 2.0), np.full(n, 8.0), 0.01, np.array([0, 0, -9.81])
    with mp.use_backend("hip"):
        ctrl = mp.ManipulatorController(dyn)
        tau = ctrl.computed_torque_control(q_d, qd_d, qdd_d, q, qd, g, dt, Kp, Ki, Kd, i_clamp=0.005)
        one = mp.ManipulatorController(dyn).computed_torque_control(q_d[2], qd_d[2], qdd_d[2], q[2], qd[2], g, dt, Kp, Ki, Kd, i_clamp=0.005)
        ff = ctrl.feedforward_control(q_d, qd_d, qdd_d, g, np.zeros(6))
    assert tau.shape == (rows, n) and one.shape == (n,)
    np.testing.assert_allclose(one, tau[2], rtol=1e-12, atol=1e-12)
    for r in range(rows):
        e = q_d[r] - q[r]
        eint = np.clip(e * dt, -0.005, 0.005)
        want = ref.mass_matrix(tab, q[r]) @ (Kp * e + Ki * eint + Kd * (qd_d[r] - qd[r])) + ref.inverse_dynamics(
            tab, q[r], qd[r], qdd_d[r], g, np.zeros(6))
        np.testing.assert_allclose(tau[r], want, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(ff[r], ref.inverse_dynamics(tab, q_d[r], qd_d[r], qdd_d[r], g, np.zeros(6)), rtol=1e-6, atol=1e-7)


def test_potential_field_kernel(ctx):
    """Fused potential field on the device vs the NumPy launcher and the reference's hand-checked values."""
    import manipulapy_amd as mp
    from manipulapy_amd import registry

    positions = np.array([[0.0, 0.0, 0.0], [2.0, 0.0, 0.0]], dtype=np.float32)
    goal = np.array([1.0, 0.0, 0.0], dtype=np.float32)
    obstacles = np.array([[0.5, 0.0, 0.0]], dtype=np.float32)
    with mp.use_backend("hip"):
        pot, grad = mp.execute_registered_kernel("potential_field.fused", positions, goal, obstacles, 1.0)
    np.testing.assert_allclose(pot, [1.0, 0.5], rtol=1e-6)
    np.testing.assert_allclose(grad, [[3.0, 0.0, 0.0], [1.0, 0.0, 0.0]], rtol=1e-6, atol=1e-6)
    rng = np.random.default_rng(4)
    P, O = 1000, 37
    pos = rng.uniform(-1, 1, (P, 3)).astype(np.float32)
    obs = rng.uniform(-1, 1, (O, 3)).astype(np.float32)
    obs[0] = pos[5]  # zero distance: ignored
    u_gpu, g_gpu = ctx.potential_field_host(pos, goal, obs, 0.6)
    u_cpu, g_cpu = registry.potential_field_cpu(pos, goal, obs, 0.6)
    np.testing.assert_allclose(u_gpu, u_cpu, rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(g_gpu, g_cpu, rtol=2e-4, atol=2e-3 * float(np.abs(g_cpu).max()) * 1e-2)
    u0, g0 = ctx.potential_field_host(pos, goal, np.zeros((0, 3)), 0.6)
    np.testing.assert_allclose(g0, pos - goal, rtol=1e-6)


def _field_terms_scale(pos, goal, obs, infl):
    """What float32 rounding errors of the fused field are relative to, per point (float64): every influenced obstacle
    contributes t = 1/d - 1/d0 with an absolute error of about one ulp of 1/d (t cancels near the influence sphere, its
    error does not), so its share of the potential carries eps / d^2 and its share of a gradient component
    eps |p - o| / d^4; the attractive part carries eps of itself."""
    pos, goal, obs = (np.asarray(a, dtype=np.float64) for a in (pos, goal, obs))
    diff = pos - goal
    rel = pos[:, None, :] - obs[None, :, :]
    d2 = (rel * rel).sum(-1)
    hit = (d2 > 0) & (d2 < infl * infl * (1 + 1e-6))
    inv = np.where(hit, 1.0 / np.sqrt(np.where(hit, d2, 1.0)), 0.0)
    gs = np.abs(diff) + ((inv**4)[:, :, None] * np.abs(rel)).sum(1)
    us = 0.5 * (diff * diff).sum(-1) + (inv**2).sum(1)
    return us, gs


def test_potential_field_kernel_against_reference_dump(ctx):
    """k_potential_field vs the REFERENCE's potential_field_cpu_fallback (tests/golden/potential_field.npz, generated by
    importing cuda_kernels/field_kernels.py:113-161): 1200 points x 37 obstacles with points sitting on obstacles
    (zero distance: skipped), on / next to the influence sphere, 1e-4 .. 5e-2 away from an obstacle (factors up to
    1e12), three influence distances incl. 0, no obstacles, and the reference's hand-checked case.
    Tolerance: the kernel takes 1/d from v_rsq_f32 (1 ulp) where the reference divides by a correctly rounded sqrt, and
    contracts d^2 into FMAs; both accumulate 37 float32 terms.  The bound is 2e-6 (a one-ulp change of 1/d measures 0.6e-6) of the rounding
    scale of each output (_field_terms_scale)."""
    z = np.load(golden_path("potential_field.npz"))
    pos, goal, obs = z["positions"], z["goal"], z["obstacles"]
    for tag in ("d035", "d100", "d000"):
        infl = float(z[f"{tag}_influence"])
        u, g = ctx.potential_field_host(pos, goal, obs, infl)
        us, gs = _field_terms_scale(pos, goal, obs, infl)
        assert u.dtype == np.float32 and g.shape == (len(pos), 3)
        assert np.isfinite(u).all() and np.isfinite(g).all()
        eu = np.abs(u.astype(np.float64) - z[f"{tag}_potential"]); eg = np.abs(g.astype(np.float64) - z[f"{tag}_gradient"])
        print(f"\nfield {tag}: max err / rounding scale: potential {float((eu / (us + 1e-30)).max()):.2e}, gradient {float((eg / (gs + 1e-30)).max()):.2e}")
        assert (eu <= 2e-6 * us + 1e-12).all(), (tag, float((eu / (us + 1e-30)).max()))
        assert (eg <= 2e-6 * gs + 1e-12).all(), (tag, float((eg / (gs + 1e-30)).max()))
        # the points that coincide with an obstacle got nothing from it: same value as with that obstacle removed
        if tag == "d035":
            k = 3
            u1, g1 = ctx.potential_field_host(pos[k:k + 1], goal, np.delete(obs, k, axis=0), infl)
            np.testing.assert_allclose(u[k], u1[0], rtol=1e-6); np.testing.assert_allclose(g[k], g1[0], rtol=1e-5, atol=1e-6)
    u, g = ctx.potential_field_host(pos, goal, np.zeros((0, 3)), 0.5)
    np.testing.assert_allclose(u, z["noobs_potential"], rtol=1e-6, atol=1e-9); np.testing.assert_allclose(g, z["noobs_gradient"], rtol=1e-6, atol=1e-9)
    u, g = ctx.potential_field_host(z["hand_positions"], [1.0, 0.0, 0.0], [[0.5, 0.0, 0.0]], 1.0)
    np.testing.assert_allclose(u, z["hand_potential"], rtol=1e-6); np.testing.assert_allclose(g, z["hand_gradient"], rtol=1e-6, atol=1e-6)


def test_planner_fused_pipeline_with_limits_and_wrench(tables):
    """batch_inverse_dynamics_trajectory (generation fused into inverse dynamics, specialised kernels through the
    planner) == joint limits clip -> oracle inverse dynamics -> torque clip, with a tip wrench and an odd row count."""
    import manipulapy_amd as mp

    tab = tables["iiwa14"]
    sm, dyn, lim = mp.load_robot("iiwa14")
    tl = np.array([[-60.0, 55.0]] * 7)
    rng = np.random.default_rng(21)
    B, N = 3, 11  # 33 rows: odd
    s = rng.uniform(lim[:, 0] - 0.3, lim[:, 1] + 0.3, (B, 7)).astype(np.float32)  # partly outside the limits -> clip
    e = rng.uniform(lim[:, 0] - 0.3, lim[:, 1] + 0.3, (B, 7)).astype(np.float32)
    F = np.array([1.0, -2.0, 0.5, 3.0, -1.5, 0.75])
    g = np.array([0.0, 0.0, -9.81])
    with mp.use_backend("hip"):
        pl = mp.OptimizedTrajectoryPlanning(sm, None, dyn, lim, torque_limits=tl)
        tau = pl.batch_inverse_dynamics_trajectory(s, e, 2.0, N, 5, g, F)
        traj = pl.batch_joint_trajectory(s, e, 2.0, N, 5)
        two = pl.inverse_dynamics_trajectory(traj["positions"].reshape(-1, 7), traj["velocities"].reshape(-1, 7),
                                             traj["accelerations"].reshape(-1, 7), g, F)
    assert tau.shape == (B, N, 7) and tau.dtype == np.float32
    o = ref.batch_joint_trajectory(lim, s, e, 2.0, N, 5)
    assert (o["positions"] <= lim[:, 1].astype(np.float32)).all() and (o["positions"] >= lim[:, 0].astype(np.float32)).all()
    np.testing.assert_allclose(traj["positions"], o["positions"], rtol=2.5e-7, atol=1e-6)
    want = ref.inverse_dynamics_trajectory(tab, o["positions"].reshape(-1, 7).astype(np.float64), o["velocities"].reshape(-1, 7).astype(np.float64),
                                           o["accelerations"].reshape(-1, 7).astype(np.float64), g, F, torque_limits=tl, dtype=np.float64)
    assert_f32(tau.reshape(-1, 7), want)
    assert_f32(two, want)
    assert tau.max() <= 55.0 and tau.min() >= -60.0


# ---------------------------------------------------------------- config c5 at its own horizon (N = 100)
def _c5_workload(tab, B, Nt, seed, dt_scale=1.0):
    """Roll-outs that stay finite for 100 steps of dt = 0.01: torques holding the start configuration against gravity plus
    a small disturbance, small per-step wrenches (the recipe of tests/golden/make_golden.py::rollout_workload; SURVEY
    §8d's free-falling arm with a 3 N tip force overflows in the reference itself within ~20 steps)."""
    from oracle import c_oracle

    rng = np.random.default_rng(seed)
    n = tab.n
    th0 = rng.uniform(-0.5, 0.5, (B, n)); dth0 = rng.uniform(-0.2, 0.2, (B, n))
    hold = c_oracle.inverse_dynamics_rows(tab, th0, np.zeros_like(th0), np.zeros_like(th0), G0_, None)[0]
    tm = hold[:, None, :] + rng.uniform(-1, 1, (B, Nt, n)) * 0.001
    Fm = rng.uniform(-1, 1, (B, Nt, 6)) * 0.02
    return th0, dth0, tm, Fm


G0_ = np.array([0.0, 0.0, -9.81])

ROLLOUT_LAYOUTS = (("batch_major", None), ("time_major", None), ("batch_major", "time_major"))


def _rollout(ctx, model, th0, dth0, tm, Fm, dt, intRes, dtype, layout="batch_major", device_layout=None):
    """fd_trajectory_host on (B,N,*) arrays whatever the layout under test: layout="time_major" hands the library (N,B,*)
    host arrays (the time-major kernel, no conversion) and turns the (N,B,n) results back; device_layout="time_major" keeps
    the (B,N,*) host arrays and lets the library convert on the device (mp_transpose_rows) around the time-major kernel."""
    if layout == "batch_major":
        return ctx.fd_trajectory_host(model, th0, dth0, tm, G0_, Fm, dt, intRes, dtype=dtype, device_layout=device_layout)
    t = lambda a: None if a is None else np.ascontiguousarray(np.swapaxes(a, 0, 1))
    got = ctx.fd_trajectory_host(model, th0, dth0, t(tm), G0_, t(Fm), dt, intRes, dtype=dtype, layout="time_major",
                                 device_layout=device_layout)
    return tuple(np.ascontiguousarray(np.swapaxes(g, 0, 1)) for g in got)


def _one_step_defect(tab, P, V, A, tm, Fm, dt, lim):
    """Teacher-forced check of every step of every trajectory: the oracle advances ONE step from the kernel's own
    previous row (float32 rows ARE the float32 kernel's state) and must land on the kernel's next row.  Independent of
    how sensitive the trajectory is to perturbations; returns the worst error relative to each row's largest entry."""
    from oracle import c_oracle

    B, Nt, n = P.shape
    p0 = P[:, :-1].reshape(-1, n).astype(np.float64); v0 = V[:, :-1].reshape(-1, n).astype(np.float64)
    t2 = np.stack([np.zeros_like(tm[:, 1:]), tm[:, 1:]], axis=2).reshape(-1, 2, n)
    f2 = np.stack([np.zeros_like(Fm[:, 1:]), Fm[:, 1:]], axis=2).reshape(-1, 2, 6)
    po, vo, ao, _ = c_oracle.fd_trajectory(tab, p0, v0, t2, G0_, f2, dt, 1, joint_limits=lim)
    worst = {}
    for name, got, want in (("positions", P[:, 1:], po[:, 1]), ("velocities", V[:, 1:], vo[:, 1]), ("accelerations", A[:, 1:], ao[:, 1])):
        e = np.abs(got.reshape(-1, n).astype(np.float64) - want)
        worst[name] = float((e.max(axis=1) / np.maximum(np.abs(want).max(axis=1), 1e-3)).max())
    return worst


def test_c5_rollout_full_horizon_against_reference_dump_and_oracle(tables):
    """BASELINE config[4] at its own horizon: xarm6, per-step wrench, N = 100, dt = 0.01, intRes = 1 — float32 and
    float64, generic and robot-specialised kernels, 130 trajectories (two full waves and a ragged one).
    (1) the reference's own N = 100 planner dump (3 trajectories); (2) every trajectory against the pinned C oracle
    over all 100 steps: north_star's 1e-4 (float32) / 1e-6 (float64) of each column's scale — the roll-out integrates,
    so an element-wise relative bound on a velocity that crosses zero is not meaningful; (3) the per-step defect."""
    from manipulapy_amd import _hip
    from oracle import c_oracle

    tab = tables["xarm6"]
    z = np.load(golden_path("fd_rollout100_xarm6.npz"))
    lim = z["joint_limits"]
    ctx = _hip.HipContext(0)
    try:
        gen = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, lim)
        spec = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, lim)
        ctx.specialize(spec)
        assert ctx.is_specialized(spec) and not ctx.is_specialized(gen)
        B, Nt = 130, 100
        th0, dth0, tm, Fm = _c5_workload(tab, B, Nt, 777)
        th0[:3], dth0[:3], tm[:3], Fm[:3] = z["theta0"], z["dtheta0"], z["taumat"], z["Ftipmat"]  # the reference's dump rides along
        report = {}
        for dtype, tol, tol_dump in ((np.float64, 1e-6, 4e-6), (np.float32, 1e-4, 1e-4)):
            x = [a.astype(dtype) for a in (th0, dth0, tm, Fm)]
            x64 = [a.astype(np.float64) for a in x]
            want = c_oracle.fd_trajectory(tab, *x64[:3], G0_, x64[3], 0.01, 1, joint_limits=lim)[:3]
            assert all(np.isfinite(w).all() for w in want)
            for model, tag, (layout, dev_layout) in ((m_, t_, l_) for (m_, t_) in ((gen, "generic"), (spec, "specialised")) for l_ in ROLLOUT_LAYOUTS):
                tag = tag + "/" + layout + ("" if dev_layout is None else ">" + dev_layout)
                got = _rollout(ctx, model, x[0], x[1], x[2], x[3], 0.01, 1, dtype, layout, dev_layout)
                for k, name in enumerate(("positions", "velocities", "accelerations")):
                    assert got[k].dtype == np.float32 and got[k].shape == (B, Nt, 6) and np.isfinite(got[k]).all()
                    scale = float(np.abs(want[k]).max())
                    drift = float(np.abs(got[k].astype(np.float64) - want[k]).max()) / scale
                    report[(np.dtype(dtype).name, tag, name)] = drift
                    assert drift <= tol, (np.dtype(dtype).name, tag, name, drift)
                    # the last lane of the ragged wave and the first lane of the second wave, separately
                    for b in (64, B - 1):
                        assert np.abs(got[k][b] - want[k][b]).max() <= tol * scale
                    ref_scale = float(np.abs(z[name]).max())
                    assert np.abs(got[k][:3] - z[name]).max() <= tol_dump * ref_scale, (tag, name)
                np.testing.assert_array_equal(got[2][:, 0], 0)
                if dtype == np.float32:  # float32 rows ARE the kernel's state; float64 state does not survive the float32 rows
                    defect = _one_step_defect(tab, *got, x64[2], x64[3], 0.01, lim)
                    for name, d in defect.items():
                        report[(np.dtype(dtype).name, tag, "defect " + name)] = d
                        # one float32 step: q' and qd' inherit a few ulps; qdd = M^-1 (...) carries eps * cond(M) (cond up to
                        # 1e3 with xarm6's 8e-5 kg.m^2 wrist), still inside north_star's 1e-4
                        assert d <= {"positions": 2e-6, "velocities": 2e-5, "accelerations": 1e-4}[name], (tag, name, d)
        print("\nc5 horizon (N = 100) max error / column scale:", {" ".join(k): f"{v:.1e}" for k, v in report.items()})
        # a finer step (dt = 0.001, intRes = 2): the float32 roll-out tracks the oracle to 1e-5
        x = [a.astype(np.float32) for a in (th0, dth0, tm, Fm)]
        x64 = [a.astype(np.float64) for a in x]
        want = c_oracle.fd_trajectory(tab, *x64[:3], G0_, x64[3], 0.001, 2, joint_limits=lim)[:3]
        for layout, dev_layout in ROLLOUT_LAYOUTS:
            got = _rollout(ctx, spec, *x[:3], x[3], 0.001, 2, np.float32, layout, dev_layout)
            for k in range(3):
                assert np.abs(got[k] - want[k]).max() <= 1e-5 * np.abs(want[k]).max()
    finally:
        ctx.destroy()


@pytest.mark.parametrize("robot", ["ur5", "iiwa14", "panda", "xarm6"])
def test_rollout_every_lane_every_tile_against_the_oracle(robot, tables):
    """The output rows of a tile leave wave-cooperatively and in whole 64-byte blocks: which lane stores which 16 bytes, and
    in which tile, depends on the trajectory index, the tile index, the row size (n = 6: 96-byte runs, tails of 0 / 32
    bytes; n = 7: 112-byte runs, tails of 0 / 16 / 32 / 48; n = 8: whole lines) and the horizon (last tile, partial tile,
    unaligned rows fall back to lane-by-lane stores).  Every element of every trajectory is compared with the C oracle,
    for horizons that end on a whole tile, on a partial tile and on unaligned rows, full and ragged waves, with and
    without wrenches, generic and robot-specialised kernels."""
    from manipulapy_amd import _hip
    from oracle import c_oracle

    tab = tables[robot]
    n = tab.n
    ctx = _hip.HipContext(0)
    try:
        gen = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
        spec = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
        ctx.specialize(spec)
        for B, Nt in ((130, 8), (65, 12), (64, 16), (130, 20), (63, 36), (1, 24), (67, 10), (66, 11), (129, 40)):
            th0, dth0, tm, Fm = _c5_workload(tab, B, Nt, 4000 + 10 * B + Nt)
            x = [a.astype(np.float32) for a in (th0, dth0, tm, Fm)]
            x64 = [a.astype(np.float64) for a in x]
            for wrench in (True, False):
                want = c_oracle.fd_trajectory(tab, *x64[:3], G0_, x64[3] if wrench else np.zeros_like(x64[3]), 0.01, 1,
                                              joint_limits=tab.joint_limits)[:3]
                assert all(np.isfinite(w).all() for w in want)
                for model in (spec, gen):
                    for layout, dev_layout in ROLLOUT_LAYOUTS:
                        got = _rollout(ctx, model, x[0], x[1], x[2], x[3] if wrench else None, 0.01, 1, np.float32, layout, dev_layout)
                        for k in range(3):
                            assert got[k].shape == (B, Nt, n)
                            scale = float(np.abs(want[k]).max())
                            err = np.abs(got[k].astype(np.float64) - want[k])
                            bad = np.argwhere(err > 1e-4 * scale)
                            assert bad.size == 0, (robot, B, Nt, wrench, layout, dev_layout, k, bad[:5].tolist(), float(err.max() / scale))
    finally:
        ctx.destroy()


def test_rollout_output_pointers_off_the_line_boundary(tables):
    """The whole-line output mode needs the three output arrays to start on a 128-byte boundary; device pointers that do not
    (a caller's sub-buffer) must take the plain path and produce the same bits."""
    from manipulapy_amd import _hip

    tab = tables["xarm6"]
    ctx = _hip.HipContext(0)
    try:
        model = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
        ctx.specialize(model)
        B, Nt, n = 130, 24, 6
        th0, dth0, tm, Fm = (a.astype(np.float32) for a in _c5_workload(tab, B, Nt, 99))
        d_in = [ctx.to_device(a) for a in (th0, dth0, tm, Fm)]
        nb = B * Nt * n * 4
        outs = {}
        for shift in (0, 16, 64, 32):
            bufs = [ctx.alloc(nb + 256) for _ in range(3)]
            for bf in bufs:
                ctx.memset(bf, 0, nb + 256)
            ctx.fd_trajectory(model, d_in[0], d_in[1], d_in[2], d_in[3], B, Nt, G0_, 0.01, 1,
                              bufs[0].offset(shift), bufs[1].offset(shift), bufs[2].offset(shift), dtype=np.float32)
            ctx.synchronize()
            got = []
            for bf in bufs:
                raw = bf.download((nb + 256,), np.uint8)
                got.append(raw[shift:shift + nb].view(np.float32).reshape(B, Nt, n).copy())
                assert not raw[shift + nb:].any() and not raw[:shift].any()   # nothing written outside the arrays
                bf.free()
            outs[shift] = got
        for shift in (16, 64, 32):
            for a, b in zip(outs[0], outs[shift]):
                np.testing.assert_array_equal(a, b)
        assert np.isfinite(outs[0][0]).all() and np.abs(outs[0][2]).max() > 0
    finally:
        ctx.destroy()


def test_stream_bandwidth_probe(ctx):
    """The device-copy microbenchmark behind `roofline.device_copy`: both byte mixes stream at HBM-class rates,
    bad arguments are refused."""
    from manipulapy_amd import _hip

    copy = ctx.stream_bandwidth(256 << 20, reads=1, reps=5)
    mix = ctx.stream_bandwidth(256 << 20, reads=3, reps=5)
    assert 1.5e3 < copy < 8.0e3 and 1.5e3 < mix < 8.0e3, (copy, mix)   # GB/s: above any PCIe / single-stack figure, below the peak
    copy_nt = ctx.stream_bandwidth(256 << 20, reads=1, reps=5, nontemporal=True)
    mix_nt = ctx.stream_bandwidth(256 << 20, reads=3, reps=5, nontemporal=True)
    assert 1.5e3 < copy_nt < 8.0e3 and 1.5e3 < mix_nt < 8.0e3, (copy_nt, mix_nt)
    with pytest.raises(_hip.HipError):
        ctx.stream_bandwidth(256 << 20, reads=2)
    with pytest.raises(_hip.HipError):
        ctx.stream_bandwidth(0, reads=1)


def test_nonfinite_rows_contract(tables):
    """The reference returns a non-finite row wherever an input of that row is NaN / inf and leaves the other rows alone
    (tests/golden/nonfinite.npz, generated by the reference; its try / except only covers exceptions).  The kernels —
    including the robot-specialised ones, which are compiled with -ffinite-math-only — do the same: such a row comes
    back as NaN, the neighbouring rows (same wave, same packed lane) are untouched; a roll-out is NaN from the first
    bad torque / wrench row (or non-finite state) on."""
    from manipulapy_amd import _hip

    z = np.load(golden_path("nonfinite.npz"))
    ctx = _hip.HipContext(0)
    try:
        tab = tables["ur5"]
        for specialise in (False, True):
            m = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, z["joint_limits"])
            if specialise:
                ctx.specialize(m)
            for dtype in (np.float32, np.float64):
                tau = ctx.id_trajectory_host(m, z["id_q"], z["id_qd"], z["id_qdd"], None, z["id_ftip"], dtype=dtype)
                np.testing.assert_array_equal(np.isfinite(tau), np.isfinite(z["id_tau"]))
                ok = np.isfinite(z["id_tau"]).all(axis=1)
                (assert_f32 if dtype == np.float32 else assert_f64)(tau[ok], z["id_tau"][ok].astype(np.float64) if dtype == np.float32 else
                                                                    ref.inverse_dynamics_trajectory(tab, z["id_q"][ok], z["id_qd"][ok], z["id_qdd"][ok], None, z["id_ftip"], dtype=np.float64))
                # a big batch (packed lanes pair row p with row p + rows/2; bad rows on both sides of the split)
                rng = np.random.default_rng(3)
                rows = 1000
                q, qd, qdd = (rng.uniform(-1, 1, (rows, 6)) for _ in range(3))
                clean = ctx.id_trajectory_host(m, q, qd, qdd, None, None, dtype=dtype)
                bad = [0, 1, 63, 64, 499, 500, 501, 777, 999]
                for i, r in enumerate(bad):
                    (q, qd, qdd)[i % 3][r, i % 6] = (np.nan, np.inf, -np.inf)[i % 3]
                dirty = ctx.id_trajectory_host(m, q, qd, qdd, None, None, dtype=dtype)
                mask = np.zeros(rows, bool); mask[bad] = True
                assert np.isnan(dirty[mask]).all()
                np.testing.assert_array_equal(dirty[~mask], clean[~mask])
                T, J, tau3 = ctx.fk_jac_id_host(m, q, qd, qdd) if dtype == np.float64 else (None, None, None)
                if tau3 is not None:
                    assert np.isnan(tau3[mask]).all() and np.isfinite(tau3[~mask]).all()
        tab = tables["xarm6"]
        for specialise in (False, True):
            m = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, z["fd_joint_limits"])
            if specialise:
                ctx.specialize(m)
            for dtype in (np.float32, np.float64):
                B = 70
                th0 = np.tile(z["fd_theta0"], (B, 1)); dth0 = np.tile(z["fd_dtheta0"], (B, 1))
                tm = np.tile(z["fd_taumat"], (B, 1, 1)); Fm = np.tile(z["fd_Ftipmat"], (B, 1, 1))
                tm[1:] = np.nan_to_num(tm[1:], nan=0.01)      # only trajectory 0 keeps the NaN of the fixture (torque row 5)
                Fm[65, 7, 4] = np.inf                          # wrench row 7 of trajectory 65
                dth0[69, 0] = np.nan                           # initial state of trajectory 69
                for layout in ("batch_major", "time_major"):
                    got = _rollout(ctx, m, th0.astype(dtype), dth0.astype(dtype), tm.astype(dtype), Fm.astype(dtype), 0.01, 1, dtype, layout)
                    for k, name in enumerate(("positions", "velocities", "accelerations")):
                        np.testing.assert_array_equal(np.isfinite(got[k][0]), np.isfinite(z["fd_" + name]))
                        np.testing.assert_allclose(got[k][0, :5], z["fd_" + name][:5], rtol=1e-4, atol=1e-4 * np.abs(z["fd_" + name][:5]).max())
                        assert np.isfinite(got[k][65, :7]).all() and np.isnan(got[k][65, 7:]).all()
                        assert np.isfinite(got[k][1:65]).all() and np.isfinite(got[k][66:69]).all()
                        assert np.isnan(got[k][69, 1:]).all()   # row 0 is the initial state as given (acceleration 0), as in the reference
                    np.testing.assert_array_equal(got[0][69, 0], th0[69].astype(dtype).astype(np.float32))
                    assert np.isnan(got[1][69, 0, 0]) and np.isfinite(got[1][69, 0, 1:]).all() and (got[2][69, 0] == 0).all()
    finally:
        ctx.destroy()


def test_planner_profiling_specialised_dispatch_and_model_release(tables):
    """enable_profiling=True (reference planning/trajectory_planning.py:283-296) puts per-call HIP-event kernel time into
    performance_stats; the planner's first big call really ran the robot-specialised kernel (mp_model_is_specialized on
    the planner's own model); destroying a model releases its code object from the context."""
    import manipulapy_amd as mp
    from manipulapy_amd import _hip, registry

    sm, dyn, lim = mp.load_robot("ur5")
    rng = np.random.default_rng(9)
    with mp.use_backend("hip"):
        pl = mp.OptimizedTrajectoryPlanning(sm, None, dyn, lim, enable_profiling=True)
        ctx = registry.get_context()
        B, N = 64, 500
        s_ = rng.uniform(lim[:, 0], lim[:, 1], (B, 6)).astype(np.float32); e_ = rng.uniform(lim[:, 0], lim[:, 1], (B, 6)).astype(np.float32)
        tau = pl.batch_inverse_dynamics_trajectory(s_, e_, 2.0, N, 5)
        assert tau.shape == (B, N, 6)
        assert ctx.is_specialized(pl._hip_model()), "the planner's model was not specialised: the generic kernels ran"
        st = pl.get_performance_stats()
        assert st["gpu_calls"] == 1 and st["gpu_timed_calls"] >= 1
        assert 0.0 < st["gpu_kernel_ms_last"] < 50.0 and st["gpu_kernel_ms_total"] >= st["gpu_kernel_ms_last"]
        r = pl.batch_joint_trajectory(s_, e_, 2.0, N, 5)
        t2 = pl.inverse_dynamics_trajectory(r["positions"].reshape(-1, 6), r["velocities"].reshape(-1, 6), r["accelerations"].reshape(-1, 6))
        # (two different kernels - the fused one keeps two timesteps per lane in packed arithmetic, the plain one a row per lane: the
        # same operations, but the compiler contracts them into FMAs independently, so the float32 results agree to rounding, not
        # to the bit; rows re-evaluated in float64 do agree exactly)
        t2 = t2.reshape(B, N, 6)
        np.testing.assert_allclose(t2, tau, rtol=1e-4, atol=1e-5 * float(np.abs(tau).max(axis=2).max()))
        hard = _hip.cpu_id_row_precision(pl._hip_model(), r["positions"].reshape(-1, 6), r["velocities"].reshape(-1, 6),
                                         r["accelerations"].reshape(-1, 6)).reshape(B, N)
        assert hard.any() and np.array_equal(t2[hard], tau[hard])
        st2 = pl.get_performance_stats()
        assert st2["gpu_timed_calls"] >= st["gpu_timed_calls"] + 2 and st2["gpu_kernel_ms_total"] > st["gpu_kernel_ms_total"]
        assert st2["gpu_kernel_ms_total"] <= st2["total_gpu_time"] * 1e3  # kernel time is inside the wall time of the calls
        pl.reset_performance_stats()
        assert pl.get_performance_stats()["gpu_timed_calls"] == 0
        pl.close()   # gives the context's profiling flag back (reference-counted: registry.acquire_profiling)
        # generic vs specialised: MANIPULAPY_HIP_SPECIALIZE is honoured per process, so check the flag on a fresh model instead
        tab = tables["ur5"]
        m = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
        assert not ctx.is_specialized(m)
        ctx.specialize(m)
        assert ctx.is_specialized(m)
        q = rng.uniform(-1, 1, (1000, 6))
        a = ctx.id_trajectory_host(m, q, q * 0.1, q * -0.2, dtype=np.float32)
        handle = m.handle
        m.destroy()   # releases the context's code object for this model (no leak per planner / per model)
        m2 = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
        assert not ctx.is_specialized(m2)
        b = ctx.id_trajectory_host(m2, q, q * 0.1, q * -0.2, dtype=np.float32)   # generic kernel
        assert np.abs(a - b).max() <= 2e-4 * np.abs(b).max()
        del handle


def test_urdf_to_kernel_for_the_reference_robot_database(monkeypatch):
    """URDF file -> manipulapy_amd.URDFToSerialManipulator -> OptimizedTrajectoryPlanning.inverse_dynamics_trajectory /
    SerialManipulator.forward_kinematics on the GPU, against the REFERENCE's torques and poses for the same URDF
    (tests/golden/urdf_suite.npz: all robots of the reference's database - UR, Panda, iiwa, Gen3, Fanuc, CRX, IRB2400,
    xArm6 + gripper, Robotiq, and the Jaco arms whose hands bring them to 9 / 10 actuated joints (run-time-n kernels,
    csrc/mp_dyn.h) - and its URDF test fixtures incl. the branched tree, the prismatic chain, mimic and continuous joints);
    float64 and float32 kernels."""
    import manipulapy_amd as mp

    monkeypatch.setenv("MANIPULAPY_HIP_SPECIALIZE", "0")   # 34 robots x ~2 s of hiprtc each buys nothing here: generic kernels
    z = np.load(golden_path("urdf_suite.npz"))
    g, F = np.array([0.0, 0.0, -9.81]), np.array([1.0, -2.0, 0.5, 3.0, -1.5, 0.75])
    done, big = 0, []
    with mp.use_backend("hip"):
        for name in [str(n) for n in z["names"]]:
            proc = mp.URDFToSerialManipulator(golden_path(os.path.join("urdf_suite", f"{name}.urdf")), tip_link=str(z[f"{name}__ee"]))
            n = proc.robot_data["actuated_joints_num"]
            th, dth, ddth, want = z[f"{name}__theta"], z[f"{name}__dtheta"], z[f"{name}__ddtheta"], z[f"{name}__tau"]
            lim = np.array([[-10.0, 10.0]] * n)   # wide limits: the planner must not clip this check's configuration
            pl = mp.OptimizedTrajectoryPlanning(proc.serial_manipulator, None, proc.dynamics, lim)
            rows = 130
            q = np.tile(th, (rows, 1)); qd = np.tile(dth, (rows, 1)); qdd = np.tile(ddth, (rows, 1))
            t64 = pl.inverse_dynamics_trajectory(q, qd, qdd, g, F)            # float64 kernel, float32 rows as the reference stores them
            assert t64.dtype == np.float32 and t64.shape == (rows, n)
            np.testing.assert_allclose(t64, np.tile(want, (rows, 1)), rtol=2e-6, atol=2e-6 * max(1.0, float(np.abs(want).max())), err_msg=name)
            t32 = pl.inverse_dynamics_trajectory(q.astype(np.float32), qd.astype(np.float32), qdd.astype(np.float32), g, F)
            assert_f32(t32, np.tile(want, (rows, 1)))
            np.testing.assert_allclose(proc.serial_manipulator.forward_kinematics(th), z[f"{name}__T"], atol=1e-10, err_msg=name)
            np.testing.assert_allclose(proc.dynamics.inverse_dynamics(th, dth, ddth, g, F), want, rtol=1e-6, atol=1e-7, err_msg=name)
            if n > 8:   # every operation of the path on the robots only the looped kernels serve
                big.append(name)
                sm, dyn = proc.serial_manipulator, proc.dynamics
                np.testing.assert_allclose(sm.jacobian(th), z[f"{name}__J"], atol=1e-10)
                np.testing.assert_allclose(dyn.mass_matrix(th), z[f"{name}__mass"], rtol=1e-9, atol=1e-11)
                np.testing.assert_allclose(dyn.gravity_forces(th, g), z[f"{name}__g"], rtol=1e-9, atol=1e-10)
                qdd = dyn.forward_dynamics(th, dth, want, g, F)
                np.testing.assert_allclose(qdd, z[f"{name}__qdd"], rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(z[f"{name}__qdd"]).max())))
                r = pl.batch_forward_dynamics_trajectory(np.tile(th, (70, 1)), np.tile(dth, (70, 1)), np.tile(want, (70, 4, 1)), g,
                                                         np.tile(F, (70, 4, 1)), 1e-4, 1)
                np.testing.assert_allclose(r["accelerations"][69, 1], z[f"{name}__qdd"], rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(z[f"{name}__qdd"]).max())))
                traj = pl.batch_joint_trajectory(np.tile(th, (3, 1)).astype(np.float32), np.tile(th + 0.1, (3, 1)).astype(np.float32), 1.0, 700, 5)
                tau_f = pl.batch_inverse_dynamics_trajectory(np.tile(th, (3, 1)).astype(np.float32), np.tile(th + 0.1, (3, 1)).astype(np.float32), 1.0, 700, 5)
                two = pl.inverse_dynamics_trajectory(traj["positions"].reshape(-1, n), traj["velocities"].reshape(-1, n), traj["accelerations"].reshape(-1, n))
                np.testing.assert_allclose(tau_f.reshape(-1, n), two, rtol=1e-5, atol=1e-5 * float(np.abs(two).max()))
                # inverse kinematics (k_dyn_ik) and the multi-start solve built on it
                rng = np.random.default_rng(4)
                goal = mp.ik_helpers.clip_to_limits(th + rng.uniform(-0.2, 0.2, (40, n)), sm.joint_limits)
                Tg = np.stack([sm.forward_kinematics(x) for x in goal])
                sol, ok, it = sm.batch_inverse_kinematics(Tg, np.tile(th, (40, 1)), max_iterations=2000, adaptive_tuning=True, backtracking=True)
                assert ok.all(), (name, ok, it)
                for a_, T_ in zip(sol, Tg):
                    assert np.abs(sm.forward_kinematics(a_)[:3, 3] - T_[:3, 3]).max() < 2e-6
                t_sol, t_ok, t_s = sm.trac_ik(Tg[0], theta0=th)
                assert t_ok and np.abs(sm.forward_kinematics(t_sol)[:3, 3] - Tg[0][:3, 3]).max() < 2e-4
            done += 1
    assert done >= 32 and sorted(big) == ["jaco_6dof", "jaco_7dof"], (done, big)


def test_transpose_rows_and_time_major_edge_cases(tables):
    """mp_transpose_rows: (outer, inner, row) -> (inner, outer, row) for every row size the roll-out uses (n = 1..8 float32 /
    float64, the 6-value wrench rows), ragged tiles and single rows / columns, bit for bit against NumPy.  And the
    time-major roll-out on its edge shapes: one trajectory, one step (row 0 only), two steps, B not a multiple of 64,
    float64 - against the batch-major kernel's rows, which the oracle tests pin."""
    from manipulapy_amd import _hip

    ctx = _hip.HipContext(0)
    try:
        rng = np.random.default_rng(11)
        for outer, inner, row_bytes in ((1, 1, 4), (33, 65, 24), (64, 100, 24), (130, 7, 28), (5, 257, 32), (100, 64, 48), (31, 33, 56), (40, 50, 64), (1, 300, 24), (300, 1, 24)):
            a = rng.integers(0, 2**32, (outer, inner, row_bytes // 4), dtype=np.uint32)
            d_a = ctx.to_device(a)
            d_b = ctx.alloc(a.nbytes)
            ctx.transpose_rows(d_a, outer, inner, row_bytes, d_b)
            got = d_b.download((inner, outer, row_bytes // 4), np.uint32)
            np.testing.assert_array_equal(got, np.swapaxes(a, 0, 1))
            d_a.free(); d_b.free()
        with pytest.raises(_hip.HipError):
            ctx.transpose_rows(ctx.alloc(64), 2, 2, 6, ctx.alloc(64))
        tab = tables["xarm6"]
        spec = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
        gen = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
        ctx.specialize(spec)
        for B, Nt in ((1, 1), (1, 2), (3, 1), (65, 2), (64, 3), (200, 5), (1, 30)):
            th0, dth0, tm, Fm = _c5_workload(tab, B, Nt, 31 * B + Nt)
            for dtype in (np.float32, np.float64):
                for model in (spec, gen):
                    for wrench in (True, False):
                        F = Fm if wrench else None
                        a = _rollout(ctx, model, th0, dth0, tm, F, 0.01, 1, dtype)
                        b = _rollout(ctx, model, th0, dth0, tm, F, 0.01, 1, dtype, "time_major")
                        for k in range(3):
                            assert b[k].shape == (B, Nt, 6)
                            np.testing.assert_allclose(b[k], a[k], rtol=0, atol=2e-6 * max(1.0, float(np.abs(a[k]).max())))
        # empty batch / empty horizon
        z = ctx.fd_trajectory_host(spec, np.zeros((0, 6)), np.zeros((0, 6)), np.zeros((4, 0, 6)), G0_, None, 0.01, 1, layout="time_major")
        assert z[0].shape == (4, 0, 6)
    finally:
        ctx.destroy()


@pytest.mark.parametrize("robot", ROBOTS)
def test_id_large_random_sample_every_row_inside_the_bound(robot, tables):
    """300 000 rows of config c2's own input distribution (start / end uniform over the joint limits, quintic, Tf = 2: velocities
    up to 12 rad/s) through the generic and the robot-specialised kernels, the fused kernel included, against the pinned C oracle.
    EVERY row inside the suite's element-wise bound - float32 1e-4 |ref| + 5e-6 max|row| with room to spare (<= 0.6 x), float64
    1e-6 |ref| + 1e-7 + 4e-9 |qd|^2 (the oracle's finite-difference noise at these speeds, bench.parity_rows).  Round 3 needed a
    second, fitted allowance here for the rows whose torque is a small difference of large terms; since round 4 the float32
    kernels take joint offsets exactly and evaluate those rows in float64 (csrc/mp_core.h, mp_rnea_row): the rows the product
    itself reports as float64 (mp_id_row_precision_cpu_f32) must land at <= 0.2 x the bound."""
    import bench
    from manipulapy_amd import _hip
    from oracle import c_oracle

    tab = tables[robot]
    n = tab.n
    lim = tab.joint_limits
    rng = np.random.default_rng(20260705 + 2)
    s_ = rng.uniform(lim[:, 0], lim[:, 1], (300, n)).astype(np.float32)
    e_ = rng.uniform(lim[:, 0], lim[:, 1], (300, n)).astype(np.float32)
    o = ref.batch_joint_trajectory(lim, s_, e_, 2.0, 1000, 5)
    q, qd, qdd = (np.ascontiguousarray(o[k].reshape(-1, n), dtype=np.float32) for k in ("positions", "velocities", "accelerations"))
    want = c_oracle.inverse_dynamics_rows(tab, q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64))[0]
    tol = 1e-4 * np.abs(want) + 5e-6 * np.abs(want).max(axis=1, keepdims=True)
    ctx = _hip.HipContext(0)
    try:
        for specialise in (False, True):
            m = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, lim)
            if specialise:
                ctx.specialize(m)
            in_f64 = _hip.cpu_id_row_precision(m, q, qd, qdd)
            for name, tau in (("id", ctx.id_trajectory_host(m, q, qd, qdd, dtype=np.float32)),
                              ("fused", ctx.traj_id_fused_host(m, s_, e_, 2.0, 1000, 5).reshape(-1, n))):
                par = bench.parity_rows(tau, want, "f32")
                assert par["ok"] and par["rows_over_first_bound"] == 0 and par["worst_over_tol"] <= 0.6, (robot, specialise, name, par)
                if in_f64.any():
                    r = (np.abs(tau[in_f64].astype(np.float64) - want[in_f64]) / tol[in_f64]).max()
                    assert r <= 0.2, (robot, specialise, name, float(r), int(in_f64.sum()))
            t64 = ctx.id_trajectory_host(m, q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64), dtype=np.float64)
            par = bench.parity_rows(t64, want, "f64", qd=qd)
            assert par["ok"], (robot, specialise, "f64", par)
    finally:
        ctx.destroy()


@pytest.mark.parametrize("robot", ROBOTS)
def test_looped_kernels_equal_the_unrolled_kernels(robot, tables, dyn_golden, monkeypatch):
    """MANIPULAPY_HIP_LOOPED=1 builds a 6..8-joint model for the run-time-n kernels (k_dyn_*, csrc/mp_dyn.h): inverse dynamics,
    FK + Jacobian, mass matrix, forward dynamics, the roll-out on both device layouts, trajectory generation and the fused
    kernel must reproduce the unrolled kernels (which the goldens pin) - float32 and float64, ragged row counts."""
    from manipulapy_amd import _hip

    tab, z = tables[robot], dyn_golden[robot]
    n = tab.n
    ctx = _hip.HipContext(0)
    try:
        unrolled = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
        monkeypatch.setenv("MANIPULAPY_HIP_LOOPED", "1")
        looped = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
        monkeypatch.delenv("MANIPULAPY_HIP_LOOPED")
        rng = np.random.default_rng(17)
        rows = 333
        q, qd, qdd = (rng.uniform(-1.5, 1.5, (rows, n)) for _ in range(3))
        g, F = z["g"], z["ftips"][0]
        for dtype, tol in ((np.float64, 1e-11), (np.float32, 3e-5)):
            for wrench in (None, F):
                a = ctx.id_trajectory_host(unrolled, q, qd, qdd, g, wrench, dtype=dtype)
                b = ctx.id_trajectory_host(looped, q, qd, qdd, g, wrench, dtype=dtype)
                np.testing.assert_allclose(b, a, rtol=tol, atol=tol * np.abs(a).max())
        Ta, Ja, ta = ctx.fk_jac_id_host(unrolled, q, qd, qdd, g, F)
        Tb, Jb, tb = ctx.fk_jac_id_host(looped, q, qd, qdd, g, F)
        np.testing.assert_allclose(Tb, Ta, rtol=0, atol=1e-12); np.testing.assert_allclose(Jb, Ja, rtol=0, atol=1e-12)
        np.testing.assert_allclose(tb, ta, rtol=1e-11, atol=1e-11 * np.abs(ta).max())
        np.testing.assert_allclose(ctx.mass_matrix_host(looped, q), ctx.mass_matrix_host(unrolled, q), rtol=1e-11, atol=1e-12)
        fa = ctx.forward_dynamics_host(unrolled, q, qd, ta, g, F)
        fb = ctx.forward_dynamics_host(looped, q, qd, ta, g, F)
        np.testing.assert_allclose(fb, fa, rtol=1e-7, atol=1e-8 * max(1.0, float(np.abs(fa).max())))
        B, Nt = 70, 10
        th0, dth0, tm, Fm = _c5_workload(tab, B, Nt, 5 + n)
        for dtype, tol in ((np.float64, 1e-6), (np.float32, 2e-4)):
            ra = _rollout(ctx, unrolled, th0, dth0, tm, Fm, 0.01, 2, dtype)
            for layout in ("batch_major", "time_major"):
                rb = _rollout(ctx, looped, th0, dth0, tm, Fm, 0.01, 2, dtype, layout)
                for k in range(3):
                    np.testing.assert_allclose(rb[k], ra[k], rtol=0, atol=tol * max(1.0, float(np.abs(ra[k]).max())))
        lim = tab.joint_limits
        s_ = rng.uniform(lim[:, 0], lim[:, 1], (5, n)).astype(np.float32); e_ = rng.uniform(lim[:, 0], lim[:, 1], (5, n)).astype(np.float32)
        pa = ctx.batch_trajectory_host(unrolled, s_, e_, 2.0, 77, 5)
        pb = ctx.batch_trajectory_host(looped, s_, e_, 2.0, 77, 5)
        for k in range(3):
            np.testing.assert_array_equal(pb[k], pa[k])
        ua = ctx.traj_id_fused_host(unrolled, s_, e_, 2.0, 77, 5)
        ub = ctx.traj_id_fused_host(looped, s_, e_, 2.0, 77, 5)
        np.testing.assert_allclose(ub, ua, rtol=3e-5, atol=3e-5 * np.abs(ua).max())
        with pytest.raises(_hip.HipError):
            ctx.specialize(looped)
        # inverse kinematics: k_dyn_ik (run-time-n kinematics) against k_ik and against the CPU launcher - one iteration template
        q0 = np.clip(q[:96] * 0.6, lim[:, 0], lim[:, 1])
        goal = np.clip(q0 + rng.uniform(-0.15, 0.15, q0.shape), lim[:, 0], lim[:, 1])
        Tg = ctx.fk_jac_id_host(unrolled, goal)[0]
        for opts in (dict(), dict(adaptive_tuning=True, backtracking=True)):
            ia = ctx.inverse_kinematics_host(unrolled, Tg, q0, lim, max_iterations=1500, **opts)
            ib = ctx.inverse_kinematics_host(looped, Tg, q0, lim, max_iterations=1500, **opts)
            ic = _hip.cpu_inverse_kinematics(looped, Tg, q0, lim, max_iterations=1500, **opts)
            # (the Panda's narrow finger joint leaves plain damped least squares at ~74 % here - on both kernels alike)
            assert ib[1].mean() > 0.7 and abs(ib[1].mean() - ia[1].mean()) <= 0.05, (ia[1].mean(), ib[1].mean())
            same = ia[2] == ib[2]
            assert same.mean() > 0.9           # rounding may move an iteration count on a slow problem; never the answer's quality
            np.testing.assert_allclose(ib[0][same & ia[1]], ia[0][same & ia[1]], rtol=0, atol=1e-6)
            assert (ib[1] == ic[1]).mean() > 0.95
            okb = ib[1]
            Tb_ = ctx.fk_jac_id_host(looped, ib[0][okb])[0]
            assert np.abs(Tb_[:, :3, 3] - Tg[okb][:, :3, 3]).max() < 2e-6
    finally:
        ctx.destroy()


def test_launch_graph_survives_the_models_it_captured(tables):
    """A captured launch graph holds kernel nodes of a model's specialised code object and the address of its device-resident
    copy, and no reference to the model: dropping the model (Python's garbage collector does that at arbitrary times) must
    not unload / free them while the graph is alive - replaying it afterwards has to give the same bits - and they are
    released with the last graph.  Also: destroying a model during an open capture must not invalidate the capture."""
    from manipulapy_amd import _hip

    tab = tables["ur5"]
    ctx = _hip.HipContext(0)
    try:
        rng = np.random.default_rng(3)
        rows = 4096
        q, qd, qdd = (rng.uniform(-1, 1, (rows, 6)).astype(np.float32) for _ in range(3))
        d = [ctx.to_device(a) for a in (q, qd, qdd)]
        out_spec, out_gen = ctx.alloc(q.nbytes), ctx.alloc(q.nbytes)
        spec = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
        gen = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)   # generic path: device-resident model copy
        ctx.specialize(spec)
        ctx.id_trajectory(spec, *d, rows, out_spec); ctx.id_trajectory(gen, *d, rows, out_gen); ctx.synchronize()
        want_spec, want_gen = out_spec.download((rows, 6), np.float32), out_gen.download((rows, 6), np.float32)
        victim = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
        ctx.specialize(victim)
        with ctx.capture() as cap:
            ctx.id_trajectory(spec, *d, rows, out_spec)
            victim.destroy()                  # inside an open capture: retired, not synchronised / freed
            ctx.id_trajectory(gen, *d, rows, out_gen)
        graph = cap.graph
        spec.destroy(); gen.destroy()         # the graph still references their code object / device copy
        for _ in range(3):
            ctx.memset(out_spec, 0, q.nbytes); ctx.memset(out_gen, 0, q.nbytes)
            # allocations in between would recycle a freed device model's buffer
            junk = [ctx.to_device(np.full(256, 7.0, np.float32)) for _ in range(8)]
            graph.launch(); ctx.synchronize()
            np.testing.assert_array_equal(out_spec.download((rows, 6), np.float32), want_spec)
            np.testing.assert_array_equal(out_gen.download((rows, 6), np.float32), want_gen)
            for j in junk:
                j.free()
        graph.destroy()                        # last graph: the retired objects go now
        m2 = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
        ctx.specialize(m2)
        ctx.id_trajectory(m2, *d, rows, out_spec); ctx.synchronize()
        np.testing.assert_array_equal(out_spec.download((rows, 6), np.float32), want_spec)
    finally:
        ctx.destroy()


def test_profiling_is_reference_counted_and_bounded():
    """enable_profiling planners share the context's flag: the last one to close() switches it off, each reports what was timed
    since ITS start, and a caller that never reads the figures does not accumulate events without bound."""
    import manipulapy_amd as mp
    from manipulapy_amd import registry

    sm, dyn, lim = mp.load_robot("ur5")
    rng = np.random.default_rng(2)
    s_ = rng.uniform(lim[:, 0], lim[:, 1], (16, 6)).astype(np.float32); e_ = rng.uniform(lim[:, 0], lim[:, 1], (16, 6)).astype(np.float32)
    with mp.use_backend("hip"):
        ctx = registry.get_context()
        p1 = mp.OptimizedTrajectoryPlanning(sm, None, dyn, lim, enable_profiling=True)
        p1.batch_inverse_dynamics_trajectory(s_, e_, 2.0, 400, 5)
        p2 = mp.OptimizedTrajectoryPlanning(sm, None, dyn, lim, enable_profiling=True)
        assert p2.get_performance_stats()["gpu_timed_calls"] == 0          # p1's calls are not p2's
        p2.batch_inverse_dynamics_trajectory(s_, e_, 2.0, 400, 5)
        assert p2.get_performance_stats()["gpu_timed_calls"] >= 1
        a = p1.get_performance_stats()["gpu_timed_calls"]
        p1.reset_performance_stats()
        assert p1.get_performance_stats()["gpu_timed_calls"] == 0 and p2.get_performance_stats()["gpu_timed_calls"] >= 1 and a >= 1
        p1.close()
        plain = mp.OptimizedTrajectoryPlanning(sm, None, dyn, lim)
        before = ctx.profile()["timed_calls"]
        plain.batch_inverse_dynamics_trajectory(s_, e_, 2.0, 400, 5)       # p2 still holds the flag: timed
        assert ctx.profile()["timed_calls"] > before
        for _ in range(200):                                                # nobody reads: pending pairs are folded in on the way
            plain.batch_inverse_dynamics_trajectory(s_[:2], e_[:2], 2.0, 64, 5)
        p2.close()
        before = ctx.profile()["timed_calls"]
        plain.batch_inverse_dynamics_trajectory(s_, e_, 2.0, 400, 5)       # last reference gone: not timed any more
        assert ctx.profile()["timed_calls"] == before
        p2.close()                                                          # idempotent


def test_uneven_allgather_on_a_one_rank_communicator():
    """mp_comm_allgatherv / mp_comm_exchange_chunk_v with nranks = 1 (all a one-GPU box can run): the local shard lands at
    its offset of the gathered buffer, in place or from a separate send buffer; argument checks.  The peer traffic itself
    is exercised by the driver's multi-GPU run (bench.py)."""
    from manipulapy_amd import _hip, sharding

    ctx = _hip.HipContext(0)
    try:
        comm = ctx.comm_create(_hip.HipContext.comm_unique_id(), 1, 0)
        counts, offsets = sharding.shard_layout(10, 1, 792)
        a = np.arange(counts[0] // 4, dtype=np.float32)
        d_a, d_all = ctx.to_device(a), ctx.alloc(counts[0])
        ctx.memset(d_all, 0, counts[0])
        comm.allgatherv(d_a, d_all, counts)
        ctx.synchronize()
        np.testing.assert_array_equal(d_all.download(a.shape, np.float32), a)
        comm.allgatherv(d_all, d_all, counts)          # in place: nothing to copy
        comm.exchange_chunk_v(d_all, offsets, [0], [counts[0] // 2])
        comm.join()
        ctx.synchronize()
        np.testing.assert_array_equal(d_all.download(a.shape, np.float32), a)
        with pytest.raises(ValueError):
            comm.allgatherv(d_a, d_all, [1, 2])
        comm.destroy()
    finally:
        ctx.destroy()


def test_planner_rollout_layouts_on_the_gpu():
    """batch_forward_dynamics_trajectory under the "hip" backend: (B, N, *) arrays through the batch-major kernel, the same
    arrays through device transposes + the time-major kernel (device_layout), and (N, B, *) arrays through the time-major
    kernel directly - the same roll-outs to float32 rounding, against the CPU launcher's."""
    import manipulapy_amd as mp

    sm, dyn, lim = mp.load_robot("xarm6")
    rng = np.random.default_rng(8)
    B, N, n = 200, 24, 6
    th0, dth0 = rng.uniform(-0.5, 0.5, (B, n)), rng.uniform(-0.2, 0.2, (B, n))
    tm, Fm = rng.uniform(-0.5, 0.5, (B, N, n)), rng.uniform(-0.02, 0.02, (B, N, 6))
    sw = lambda x: np.ascontiguousarray(np.swapaxes(x, 0, 1))
    cpu = mp.OptimizedTrajectoryPlanning(sm, None, dyn, lim, use_cuda=False).batch_forward_dynamics_trajectory(th0, dth0, tm, None, Fm, 0.005, 1)
    with mp.use_backend("hip"):
        pl = mp.OptimizedTrajectoryPlanning(sm, None, dyn, lim)
        a = pl.batch_forward_dynamics_trajectory(th0, dth0, tm, None, Fm, 0.005, 1)
        b = pl.batch_forward_dynamics_trajectory(th0, dth0, tm, None, Fm, 0.005, 1, device_layout="time_major")
        c = pl.batch_forward_dynamics_trajectory(th0, dth0, sw(tm), None, sw(Fm), 0.005, 1, layout="time_major")
    for k in ("positions", "velocities", "accelerations"):
        scale = max(1.0, float(np.abs(cpu[k]).max()))
        assert c[k].shape == (N, B, n) and a[k].shape == b[k].shape == (B, N, n)
        for got in (a[k], b[k], sw(c[k])):
            np.testing.assert_allclose(got, cpu[k], rtol=0, atol=1e-6 * scale)


def test_gain_sweep_is_one_launch_and_matches_the_reference(tables, monkeypatch):
    """ManipulatorController.find_ultimate_gain_and_period under "hip": the whole gain ladder is one k_pd_regulation launch and
    gives the reference's own ultimate gain / period / error histories (tests/golden/gain_sweep_ur5.npz); kernel = CPU launcher;
    the run-time-n kernel (k_dyn_pd_regulation) = the unrolled one; a 10-joint arm runs."""
    import manipulapy_amd as mp
    from manipulapy_amd import _hip
    from test_round3_host import _jaco, check_gain_sweep

    sm, dyn, lim = mp.load_robot("ur5")
    with mp.use_backend("hip"):
        check_gain_sweep(mp.ManipulatorController(dyn))
        zj, proc = _jaco("jaco_7dof")
        th = zj["jaco_7dof__theta"]
        Ku, Tu, gains, errs = mp.ManipulatorController(proc.dynamics).find_ultimate_gain_and_period(th, th + 0.05, 0.002, 20)
        with mp.use_backend("numpy"):
            Ku2, Tu2, gains2, errs2 = mp.ManipulatorController(proc.dynamics).find_ultimate_gain_and_period(th, th + 0.05, 0.002, 20)
        assert (Ku, Tu, gains) == (Ku2, Tu2, gains2)
        np.testing.assert_allclose(np.stack(errs), np.stack(errs2), rtol=1e-9)
    tab = tables["xarm6"]
    ctx = _hip.HipContext(0)
    try:
        unrolled = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
        monkeypatch.setenv("MANIPULAPY_HIP_LOOPED", "1")
        looped = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
        monkeypatch.delenv("MANIPULAPY_HIP_LOOPED")
        rng = np.random.default_rng(6)
        K = 150                                                      # three waves, the last one ragged
        th0, des = rng.uniform(-1, 1, (K, 6)), rng.uniform(-1, 1, (K, 6))
        Kp, Kd = rng.uniform(0.01, 3.0, K), rng.uniform(0.0, 1e-4, K)
        a = ctx.pd_regulation_host(unrolled, th0, des, Kp, Kd, [0.0, 0.0, -9.81], 0.004, 50)
        b = ctx.pd_regulation_host(looped, th0, des, Kp, Kd, [0.0, 0.0, -9.81], 0.004, 50)
        c = _hip.cpu_pd_regulation(unrolled, th0, des, Kp, Kd, [0.0, 0.0, -9.81], 0.004, 50)
        np.testing.assert_array_equal(a[1], c[1]); np.testing.assert_array_equal(b[1], c[1])
        live = np.isfinite(c[0])
        assert live.mean() > 0.9
        # closed-loop runs amplify rounding differences: compare where the host run is still of ordinary size
        tame = live & (np.abs(c[0]) < 1e3)
        np.testing.assert_allclose(a[0][tame], c[0][tame], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(b[0][tame], c[0][tame], rtol=1e-6, atol=1e-9)
        e0, c0 = ctx.pd_regulation_host(unrolled, th0[:3], des[:3], Kp[:3], Kd[:3], None, 0.004, 0)
        assert e0.shape == (3, 0) and (c0 == 0).all()
    finally:
        ctx.destroy()


def test_urdf_processor_surface_under_the_hip_backend():
    """The processor's convenience surface (tests/golden/urdf_api.npz) with the tip batches of plain serial chains routed to the
    kinematics kernel."""
    import manipulapy_amd as mp
    from test_round3_host import check_urdf_processor_surface

    with mp.use_backend("hip"):
        check_urdf_processor_surface()


def test_mass_matrix_store_paths_every_n_and_tail():
    """The mass-matrix kernel's three row-store paths (whole 128-byte lines / whole 16-byte chunks / odd rows as 16-byte chunks across
    the row boundaries, csrc/mp_bodies.h mp_wave_store_auto): every joint count 1..8, float32 and float64, row counts around the
    16-row staging passes and the 64-row wave, against the CPU launcher; a guard band behind the output must stay untouched."""
    from manipulapy_amd import _hip
    from test_random_robots import FLAVOURS, random_robot

    ctx = _hip.HipContext(0)
    try:
        rng = np.random.default_rng(77)
        for n in range(1, 9):
            tab = random_robot(rng, n, FLAVOURS[n % len(FLAVOURS)])
            model = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
            for rows in (1, 15, 16, 17, 31, 33, 63, 64, 65, 130, 333):
                q = rng.uniform(-2, 2, (rows, n))
                want = _hip.cpu_mass_matrix(model, q)
                for dtype, tol in ((np.float64, 1e-11), (np.float32, 3e-5)):
                    item = np.dtype(dtype).itemsize
                    nb = rows * n * n * item
                    guard = 4096
                    d_q = ctx.to_device(q.astype(dtype))
                    d_M = ctx.alloc(nb + guard)
                    ctx.memset(d_M, 0xA5, nb + guard)
                    ctx.mass_matrix(model, d_q, rows, d_M, dtype=dtype)
                    ctx.synchronize()
                    raw = d_M.download((nb + guard,), np.uint8)
                    assert (raw[nb:] == 0xA5).all(), (n, rows, dtype)              # nothing written past the last valid row
                    got = raw[:nb].view(dtype).reshape(rows, n, n)
                    np.testing.assert_allclose(got, want, rtol=tol, atol=tol * np.abs(want).max(), err_msg=f"n={n} rows={rows} {dtype}")
                    d_q.free(); d_M.free()
            model.destroy()
    finally:
        ctx.destroy()


@pytest.mark.parametrize("robot", ["ur5", "panda7", "panda"])
def test_whole_line_inverse_dynamics_kernel_edges(robot):
    """mp_spec_id_co (rows >= 64 of a specialised float32 model move as whole lines, non-temporal, through LDS; the last < 64 rows
    take the per-lane kernel): n = 6 / 7 / 8, row counts around the 64-row wave and the 256-row block, against the PINNED C ORACLE
    (oracle/oracle.c - not the product's own CPU launcher, which shares the kernels' templates); rows on either side of the
    hand-over; a NaN / inf row inside a full wave poisons only itself; device pointers at odd multiples of 16 bytes; nothing is
    written past the end."""
    import bench
    import manipulapy_amd as mp
    from manipulapy_amd import _hip
    from oracle import c_oracle

    t = mp.robot_tables(robot)
    tab = bench.oracle_tables(ref, robot)
    n = t["S_list"].shape[1]

    def oracle(q, qd, qdd, wrench=None):
        return c_oracle.inverse_dynamics_rows(tab, q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64), None, wrench)[0]

    ctx = _hip.HipContext(0)
    try:
        model = _hip.HipModel(t["S_list"], t["Mlist_per_link"], t["Glist"], t["M_ee"], t["joint_limits"])
        ctx.specialize(model)
        rng = np.random.default_rng(31)
        F = np.array([1.0, -2.0, 0.5, 3.0, -1.5, 0.75])
        for rows in (1, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 4096 + 77):
            q, qd, qdd = (rng.uniform(-1.5, 1.5, (rows, n)).astype(np.float32) for _ in range(3))
            for wrench in (None, F):
                got = ctx.id_trajectory_host(model, q, qd, qdd, None, wrench, dtype=np.float32)
                assert_f32(got, oracle(q, qd, qdd, wrench))
        # a non-finite row inside a full wave, and one in the per-lane tail
        rows = 64 * 3 + 20
        q, qd, qdd = (rng.uniform(-1, 1, (rows, n)).astype(np.float32) for _ in range(3))
        q[70, 0] = np.nan; qd[130, n - 1] = np.inf; qdd[rows - 3, 1] = -np.inf
        got = ctx.id_trajectory_host(model, q, qd, qdd, dtype=np.float32)
        bad = np.zeros(rows, bool); bad[[70, 130, rows - 3]] = True
        assert np.isnan(got[bad]).all() and np.isfinite(got[~bad]).all()
        # device pointers 16 bytes into their buffers (the chunks are 16-byte aligned, not line aligned), guard band behind tau
        rows = 64 * 5 + 9
        q, qd, qdd = (rng.uniform(-1, 1, (rows, n)).astype(np.float32) for _ in range(3))
        nb = rows * n * 4
        bufs = []
        for a in (q, qd, qdd):
            d = ctx.alloc(nb + 64)
            _hip._check(ctx.lib.mp_memcpy_h2d(ctx.handle, d.offset(16), a.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(nb)))
            bufs.append(d)
        d_tau = ctx.alloc(nb + 4096)
        ctx.memset(d_tau, 0x5A, nb + 4096)
        ctx.id_trajectory(model, bufs[0].offset(16), bufs[1].offset(16), bufs[2].offset(16), rows, d_tau.offset(16), dtype=np.float32)
        ctx.synchronize()
        raw = d_tau.download((nb + 4096,), np.uint8)
        assert (raw[:16] == 0x5A).all() and (raw[16 + nb:] == 0x5A).all()
        got = raw[16:16 + nb].view(np.float32).reshape(rows, n)
        assert_f32(got, oracle(q, qd, qdd))
    finally:
        ctx.destroy()


def test_planner_benchmark_helpers_on_the_gpu():
    """benchmark_all_kernels: the reference's five kernel names all run (one kernel here); benchmark_performance reports the GPU
    route and a measured CPU-launcher comparison."""
    import manipulapy_amd as mp

    sm, dyn, lim = mp.load_robot("ur5")
    with mp.use_backend("hip"):
        pl = mp.OptimizedTrajectoryPlanning(sm, mp.robot_urdf("ur5"), dyn, lim, cuda_threshold=1)
        r = pl.benchmark_all_kernels(N=10000, num_joints=6, num_runs=2)   # (N x n above the planner's GPU threshold of ~42 700 elements)
        assert sorted(r) == ["cache_friendly", "memory_optimized", "standard", "vectorized", "warp_optimized"]
        assert all(e["success_rate"] == 1.0 and e["trajectory_shape"] == (10000, 6) and len(e["all_times"]) == 2 for e in r.values())
        b = pl.benchmark_performance([{"N": 10000, "joints": 6, "name": "case"}])
        assert b["case"]["used_gpu"] and b["case"]["cpu_time"] > 0 and b["case"]["actual_speedup"] > 0 and b["case"]["stats"]["gpu_calls"] == 3
        pl.close()


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5])
def test_whole_line_kernel_small_joint_counts(n):
    """mp_spec_id_co for 1..5 joints (4..20-byte rows: 16..80 chunks per wave and array, partly filled chunk instructions) on random
    chains: rows through the whole-line kernel and the per-lane tail against the pinned C oracle, with and without a tip wrench."""
    from manipulapy_amd import _hip
    from oracle import c_oracle
    from test_random_robots import FLAVOURS, random_robot

    rng = np.random.default_rng(500 + n)
    tab = random_robot(rng, n, FLAVOURS[(n + 2) % len(FLAVOURS)])
    ctx = _hip.HipContext(0)
    try:
        model = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
        ctx.specialize(model)
        for rows in (64, 64 * 7 + 5, 2048 + 63):
            q, qd, qdd = (rng.uniform(-1.5, 1.5, (rows, n)).astype(np.float32) for _ in range(3))
            for wrench in (None, np.array([0.5, -1.0, 0.25, 2.0, -1.5, 0.75])):
                want = c_oracle.inverse_dynamics_rows(tab, q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64), None, wrench)[0]
                got = ctx.id_trajectory_host(model, q, qd, qdd, None, wrench, dtype=np.float32)
                # (a one- or two-value row has no "row scale" to hold its cancellations against: the batch's scale is the floor here)
                np.testing.assert_allclose(got, want, rtol=1e-4, atol=5e-6 * float(np.abs(want).max()))
    finally:
        ctx.destroy()


@pytest.mark.parametrize("specialise", [False, True])
def test_float64_pass_is_parked_and_run_before_anything_can_see_the_difference(specialise, tables):
    """Round 4: a device-pointer float32 inverse-dynamics launch leaves its ill-conditioned rows to a float64 pass that the
    context parks (csrc/mp_capi.cpp, hard_defer / hard_flush) and runs - several launches' worth in one kernel - before the next entry
    point that could read the torques.  Whatever is launched in between, a download must return exactly what the host-buffer entry
    point returns (which flushes at once): launches on other buffers (parked side by side, more than four of them), a launch that
    overwrites a parked launch's INPUT (the pass re-reads its rows: it must run first), a launch onto a parked launch's OUTPUT, a
    captured graph (its passes are nodes of the graph) and an empty launch."""
    from manipulapy_amd import _hip
    from oracle import c_oracle

    tab = tables["ur5"]
    lim = tab.joint_limits
    rng = np.random.default_rng(77)
    ctx = _hip.HipContext(0)
    try:
        m = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, lim)
        if specialise:
            ctx.specialize(m)
        sets = []
        for k in range(6):   # six independent (q, qd, qdd) histories of fast trajectories: ~1 % of their rows are ill-conditioned
            s_ = rng.uniform(lim[:, 0], lim[:, 1], (40, 6)).astype(np.float32)
            e_ = rng.uniform(lim[:, 0], lim[:, 1], (40, 6)).astype(np.float32)
            o = ref.batch_joint_trajectory(lim, s_, e_, 2.0, 1000 + k, 5)     # row counts that are not multiples of 64: the tail kernel joins in
            q, qd, qdd = (np.ascontiguousarray(o[key].reshape(-1, 6), dtype=np.float32) for key in ("positions", "velocities", "accelerations"))
            want = ctx.id_trajectory_host(m, q, qd, qdd, dtype=np.float32)
            assert _hip.cpu_id_row_precision(m, q, qd, qdd).sum() > 50
            sets.append({"q": q, "qd": qd, "qdd": qdd, "want": want, "rows": len(q),
                         "d": [ctx.to_device(a) for a in (q, qd, qdd)], "d_tau": ctx.alloc(q.nbytes)})
        get = lambda st: st["d_tau"].download(st["q"].shape, np.float32)
        run = lambda st: ctx.id_trajectory(m, *st["d"], st["rows"], st["d_tau"], dtype=np.float32)
        # (1) six launches back to back on disjoint buffers (more than the context parks), then the downloads
        for st in sets:
            ctx.memset(st["d_tau"], 0, st["q"].nbytes)
        for st in sets:
            run(st)
        for st in sets:
            np.testing.assert_array_equal(get(st), st["want"])
        # every row inside the suite's bound against the oracle, the float64-evaluated ones with room to spare
        w64 = c_oracle.inverse_dynamics_rows(tab, *(sets[0][k].astype(np.float64) for k in ("q", "qd", "qdd")))[0]
        assert_f32(sets[0]["want"], w64)
        # (2) a launch parked, then its INPUT buffer overwritten through the API and another launch: each result is its own inputs'
        a, b = sets[0], sets[1]
        run(a)
        n = min(a["rows"], b["rows"])
        for d, key in zip(a["d"], ("q", "qd", "qdd")):
            d.upload(np.ascontiguousarray(np.concatenate([b[key][:n], a[key][n:]])))
        mixed = ctx.id_trajectory_host(m, *(np.concatenate([b[k][:n], a[k][n:]]) for k in ("q", "qd", "qdd")), dtype=np.float32)
        np.testing.assert_array_equal(get(a), a["want"])          # the parked pass ran on the OLD rows before the upload replaced them
        run(a)
        np.testing.assert_array_equal(get(a), mixed)
        for d, key in zip(a["d"], ("q", "qd", "qdd")):
            d.upload(a[key])
        # (3) two launches onto the SAME output buffer from different inputs: the second one's result stands
        c, d_ = sets[2], sets[3]
        n = min(c["rows"], d_["rows"])
        ctx.id_trajectory(m, *c["d"], n, c["d_tau"], dtype=np.float32)
        ctx.id_trajectory(m, *d_["d"], n, c["d_tau"], dtype=np.float32)
        np.testing.assert_array_equal(get(c)[:n], d_["want"][:n])
        # (4) captured launches carry their float64 passes as nodes of the graph (round 5; lists and counters belong to the graph):
        # SIX launches on disjoint buffers in one capture - more than a pool parks, so a pass is captured in between and the rest at
        # mp_graph_end - replayed on the captured inputs and then on exchanged ones; every download bit-equal to the host entry point
        # (the in-place re-evaluation a capture used to fall back to is correct but NOT bit-equal to the pass)
        ctx.synchronize()
        with ctx.capture() as cap:
            for st in sets:
                run(st)
        for rep in range(3):
            want = [st["want"] for st in sets]
            if rep == 2:   # fresh contents in the same buffers: every history reversed in time
                for st in sets:
                    for d, key in zip(st["d"], ("q", "qd", "qdd")):
                        d.upload(np.ascontiguousarray(st[key][::-1]))
                want = [ctx.id_trajectory_host(m, *(np.ascontiguousarray(st[k][::-1]) for k in ("q", "qd", "qdd")), dtype=np.float32) for st in sets]
            for st in sets:
                ctx.memset(st["d_tau"], 0, st["q"].nbytes)
            cap.graph.launch()
            if rep == 0:
                run(sets[0])   # an eager launch right behind a replay: the context's own pool, not the graph's
            for st, w in zip(sets, want):
                np.testing.assert_array_equal(get(st), w)
        cap.graph.destroy()
        for st in sets:
            for d, key in zip(st["d"], ("q", "qd", "qdd")):
                d.upload(st[key])
        # (5) nothing parked, nothing to run: an empty launch and a synchronise are fine
        ctx.id_trajectory(m, *sets[5]["d"], 0, sets[5]["d_tau"], dtype=np.float32)
        ctx.synchronize()
        # (6) the fused generation + inverse dynamics parks its pass too (specialised kernels; it re-reads the end points and the
        # context's time table): five launches back to back on their own outputs, then one with another N - which rewrites the table,
        # so the parked passes run first - then the end points of a parked launch replaced through the API
        B, N = 9, 700
        pairs = [rng.uniform(lim[:, 0], lim[:, 1], (2, B, 6)).astype(np.float32) for _ in range(6)]
        want = [ctx.traj_id_fused_host(m, se[0], se[1], 2.0, N, 5) for se in pairs]
        d_se = [(ctx.to_device(se[0]), ctx.to_device(se[1])) for se in pairs]
        d_out = [ctx.alloc(B * N * 6 * 4) for _ in pairs]
        for (ds, de), do in zip(d_se[:5], d_out[:5]):
            ctx.traj_id_fused(m, ds, de, B, N, 2.0, 5, do)
        other = ctx.traj_id_fused_host(m, pairs[5][0], pairs[5][1], 1.5, N - 43, 5)       # host entry: flushes, rewrites the table
        ctx.traj_id_fused(m, *d_se[5], B, N - 43, 1.5, 5, d_out[5])
        for k in range(5):
            np.testing.assert_array_equal(d_out[k].download((B, N, 6), np.float32), want[k])
        np.testing.assert_array_equal(d_out[5].download((B, N - 43, 6), np.float32), other)
        ctx.traj_id_fused(m, *d_se[0], B, N, 2.0, 5, d_out[0])
        d_se[0][1].upload(pairs[1][1])                                   # an entry point: the parked pass ran on the OLD end points
        np.testing.assert_array_equal(d_out[0].download((B, N, 6), np.float32), want[0])
        ctx.traj_id_fused(m, *d_se[0], B, N, 2.0, 5, d_out[0])
        np.testing.assert_array_equal(d_out[0].download((B, N, 6), np.float32), ctx.traj_id_fused_host(m, pairs[0][0], pairs[1][1], 2.0, N, 5))
        w64 = c_oracle.inverse_dynamics_rows(tab, *(ref.batch_joint_trajectory(lim, pairs[2][0], pairs[2][1], 2.0, N, 5)[key].reshape(-1, 6).astype(np.float64)
                                                    for key in ("positions", "velocities", "accelerations")))[0]
        assert_f32(want[2].reshape(-1, 6), w64)
    finally:
        ctx.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ur10", "iiwa7", "gen3", "crx10ia", "fanuc_m16ib", "abb_irb2400"])
def test_adaptive_rows_on_arms_the_rule_was_not_fitted_on(name):
    """Round 5: the float32 rows' conditioning rule (csrc/mp_core.h, MpRowScale) was chosen on the UR5.  Six arms
    the benchmark does not use, 60 000 c2-distributed rows each (joint speeds up to ~10 rad/s), generic and robot-specialised
    KERNELS against the pinned C oracle: no row over the suite's element-wise float32 bound, the worst at <= 0.6 x.  The CPU
    launcher's side of the statement, on all 34 suite robots: tests/test_round5_rule.py."""
    import bench
    from manipulapy_amd import _hip
    from oracle import c_oracle
    from test_round5_rule import c2_rows, suite_robot

    tab, model, lim = suite_robot(name)
    q, qd, qdd = c2_rows(lim, 60, 4242)
    want = c_oracle.inverse_dynamics_rows(tab, q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64))[0]
    ctx = _hip.HipContext(0)
    try:
        for tag in ("generic", "specialised"):
            if tag == "specialised":
                ctx.specialize(model)
                assert ctx.is_specialized(model)
            tau = ctx.id_trajectory_host(model, q, qd, qdd, dtype=np.float32)
            par = bench.parity_rows(tau, want, "f32")
            assert par["ok"] and par["rows_over_first_bound"] == 0 and par["worst_over_tol"] <= 0.6, (name, tag, par)
    finally:
        ctx.destroy()


class _RawHip:
    """The HIP runtime itself through ctypes: what a C-ABI caller with arrays of its own (hipMalloc, a framework tensor) uses next
    to the library.  Test helper only."""

    def __init__(self):
        import ctypes
        self.c = ctypes
        self.rt = ctypes.CDLL("libamdhip64.so")
        self.rt.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        self.rt.hipFree.argtypes = [ctypes.c_void_p]
        self.rt.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        self.rt.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        self.rt.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
        self.rt.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]

    def malloc(self, nbytes):
        p = self.c.c_void_p()
        assert self.rt.hipMalloc(self.c.byref(p), max(nbytes, 16)) == 0
        return p

    def upload(self, p, a):
        assert self.rt.hipMemcpy(p, a.ctypes.data_as(self.c.c_void_p), a.nbytes, 1) == 0   # hipMemcpyHostToDevice, blocking

    def download_on(self, stream, p, shape, dtype):
        out = np.empty(shape, dtype)
        assert self.rt.hipMemcpyAsync(out.ctypes.data_as(self.c.c_void_p), p, out.nbytes, 2, self.c.c_void_p(stream)) == 0
        assert self.rt.hipStreamSynchronize(self.c.c_void_p(stream)) == 0
        return out


@pytest.mark.gpu
@pytest.mark.parametrize("specialise", [False, True])
def test_caller_owned_device_arrays_are_complete_in_stream_order(specialise, tables):
    """Round 5: the float64 pass over a float32 launch's ill-conditioned rows is parked only when
    every array of the launch comes from the context's pool.  For arrays the CALLER allocated (hipMalloc here) it is enqueued at
    once behind the kernel: a raw hipMemcpyAsync on the compute stream - no mp_* call between the launch and the read - returns
    exactly what the host-buffer entry point returns, as the reference's launchers return finished arrays
    (cuda_kernels/trajectory_kernels.py:1043-1081).  Given rows and generated rows (the fused launch), and a mixed launch (inputs
    from the pool, output the caller's)."""
    from manipulapy_amd import _hip

    tab = tables["ur5"]
    lim = tab.joint_limits
    rng = np.random.default_rng(91)
    hip = _RawHip()
    ctx = _hip.HipContext(0)
    raw = []
    try:
        m = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, lim)
        if specialise:
            ctx.specialize(m)
        B, N = 70, 1001
        s_ = rng.uniform(lim[:, 0], lim[:, 1], (B, 6)).astype(np.float32)
        e_ = rng.uniform(lim[:, 0], lim[:, 1], (B, 6)).astype(np.float32)
        o = ref.batch_joint_trajectory(lim, s_, e_, 2.0, N, 5)
        q, qd, qdd = (np.ascontiguousarray(o[key].reshape(-1, 6), dtype=np.float32) for key in ("positions", "velocities", "accelerations"))
        rows = len(q)
        assert _hip.cpu_id_row_precision(m, q, qd, qdd).sum() > 100          # there ARE rows the pass has to rewrite
        want = ctx.id_trajectory_host(m, q, qd, qdd, dtype=np.float32)
        want_fused = ctx.traj_id_fused_host(m, s_, e_, 2.0, N, 5)
        stream = ctx.stream()
        assert stream
        d = [hip.malloc(a.nbytes) for a in (q, qd, qdd)]
        d_tau = hip.malloc(q.nbytes)
        raw += d + [d_tau]
        for p_, a in zip(d, (q, qd, qdd)):
            hip.upload(p_, a)
        for _ in range(3):   # (three times: a parked pass of an earlier launch would show up as a stale row of a later read)
            assert hip.rt.hipMemsetAsync(d_tau, 0, q.nbytes, hip.c.c_void_p(stream)) == 0
            ctx.id_trajectory(m, *d, rows, d_tau, dtype=np.float32)
            np.testing.assert_array_equal(hip.download_on(stream, d_tau, q.shape, np.float32), want)
        # inputs from the pool, the output the caller's: still at once
        pool_in = [ctx.to_device(a) for a in (q, qd, qdd)]
        assert hip.rt.hipMemsetAsync(d_tau, 0, q.nbytes, hip.c.c_void_p(stream)) == 0
        ctx.id_trajectory(m, *pool_in, rows, d_tau, dtype=np.float32)
        np.testing.assert_array_equal(hip.download_on(stream, d_tau, q.shape, np.float32), want)
        # the fused generation + inverse dynamics on the caller's start / end / tau arrays
        ds, de = hip.malloc(s_.nbytes), hip.malloc(e_.nbytes)
        raw += [ds, de]
        hip.upload(ds, s_); hip.upload(de, e_)
        ctx.traj_id_fused_host(m, s_[:2], e_[:2], 2.0, N, 5)                 # sizes the time table for this N
        assert hip.rt.hipMemsetAsync(d_tau, 0, q.nbytes, hip.c.c_void_p(stream)) == 0
        ctx.traj_id_fused(m, ds, de, B, N, 2.0, 5, d_tau)
        np.testing.assert_array_equal(hip.download_on(stream, d_tau, (B, N, 6), np.float32), want_fused)
        # Pool arrays (mp_malloc: all four) read by a RAW copy on the handed-out stream, no mp_* call between launch and read (ADVICE
        # r5): mp_malloc returns plain device pointers, so once mp_ctx_get_stream has been called nothing stays parked any more
        # (mp_ctx::stream_exported, sticky) - given rows and generated rows, twice each
        d_pool_tau = ctx.alloc(q.nbytes)
        pool_s, pool_e = ctx.to_device(s_), ctx.to_device(e_)
        for _ in range(2):
            ctx.memset(d_pool_tau, 0, q.nbytes)
            ctx.id_trajectory(m, *pool_in, rows, d_pool_tau, dtype=np.float32)
            np.testing.assert_array_equal(hip.download_on(stream, d_pool_tau.ptr, q.shape, np.float32), want)
            ctx.memset(d_pool_tau, 0, q.nbytes)
            ctx.traj_id_fused(m, pool_s, pool_e, B, N, 2.0, 5, d_pool_tau)
            np.testing.assert_array_equal(hip.download_on(stream, d_pool_tau.ptr, (B, N, 6), np.float32), want_fused)
        np.testing.assert_array_equal(d_pool_tau.download((B, N, 6), np.float32), want_fused)
    finally:
        ctx.synchronize()
        for p_ in raw:
            hip.rt.hipFree(p_)
        ctx.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("specialise", [False, True])
def test_more_ill_conditioned_rows_than_the_list_holds(specialise, tables):
    """An arm balanced upright: joint forces carry its weight, every torque is ~0.1 N.m - EVERY row is ill-conditioned in float32.  The
    list a launch hands its rows to the float64 pass in holds one row in eight (at least 65 536): past that the count tells the pass
    to evaluate EVERY row of the launch (csrc/mp_bodies.h, mp_push_hard_rows / mp_body_id_hard), so that the result does not depend on
    which waves found room.  Every row inside a tenth of the float32 bound, and twice the same launch gives the same bits."""
    from manipulapy_amd import _hip
    from oracle import c_oracle

    tab = tables["ur5"]
    rng = np.random.default_rng(5)
    rows = 600_000 + 37
    q = (np.array([0.0, -np.pi / 2, 0.0, -np.pi / 2, 0.0, 0.0]) + rng.uniform(-2e-3, 2e-3, (rows, 6))).astype(np.float32)
    z = np.zeros_like(q)
    # non-finite rows in a launch whose list overflows: the pass then walks EVERY row and must leave these as the float32 kernel
    # stored them (NaN rows, reference planning/trajectory_dynamics.py:345-358) - with torque limits, so that a clip could launder them
    qd = z.copy()
    bad_rows = [100, 70_001, rows - 5]
    q[bad_rows[0], 2] = np.nan
    qd[bad_rows[1], 0] = np.inf
    qd[bad_rows[2], 5] = -np.inf
    ctx = _hip.HipContext(0)
    try:
        m = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits, np.tile([-150.0, 150.0], (6, 1)))
        if specialise:
            ctx.specialize(m)
        assert _hip.cpu_id_row_precision(m, q[:4096], z[:4096], z[:4096])[np.r_[0:100, 101:4096]].all()
        d = [ctx.to_device(a) for a in (q, qd, z)]
        d_tau = ctx.alloc(q.nbytes)
        got = []
        for _ in range(2):
            ctx.memset(d_tau, 0, q.nbytes)
            ctx.id_trajectory(m, *d, rows, d_tau, dtype=np.float32)
            got.append(d_tau.download(q.shape, np.float32))
        np.testing.assert_array_equal(got[0], got[1])
        assert np.isnan(got[0][bad_rows]).all(), got[0][bad_rows]
        neighbours = np.array([r + k for r in bad_rows for k in (-1, 1)])
        assert np.isfinite(got[0][neighbours]).all()
        sample = np.setdiff1d(np.r_[0:3000, rows - 3000:rows], bad_rows)
        want = c_oracle.inverse_dynamics_rows(tab, q[sample].astype(np.float64), z[sample].astype(np.float64), z[sample].astype(np.float64))[0]
        err = np.abs(got[0][sample].astype(np.float64) - want)
        tol = 1e-4 * np.abs(want) + 5e-6 * np.abs(want).max(axis=1, keepdims=True)
        assert (err / tol).max() < 0.1, float((err / tol).max())   # float32 rows of this kind sit at 1 - 5 x the bound
        f64 = ctx.id_trajectory_host(m, q[sample], z[sample], z[sample], dtype=np.float64)
        np.testing.assert_allclose(got[0][sample], f64, rtol=0, atol=2e-7)
    finally:
        ctx.destroy()



@pytest.mark.gpu
@pytest.mark.parametrize("foreign", [False, True])
def test_parked_pass_stress_against_a_context_that_never_parks(foreign):
    """Round 6: the parked / carried / captured float64 passes are the most intricate host logic of the library
    (csrc/mp_capi.cpp: attach_hard_list, hard_defer, hard_park_or_run, hard_flush_if_overlapping, pick_rider); their randomised
    coverage was a manual tool.  A bounded run of it, fixed seed: ~300 operations - float32 launches on overlapping sub-ranges of shared
    arrays of three models (two specialised programs, one generic), outputs landing in other launches' inputs, fused launches with
    table rewrites, memsets, uploads, float64 launches, synchronisations, launches on arrays that are FREED while their pass is parked
    and allocated again at once ("recycle": the pool hands the same blocks back), and launch graphs of two or three launches replayed
    once or twice.  Every download must equal, bit for bit, a host mirror advanced with what a SECOND context's host entry point
    returns for the same rows (it runs its pass at once, never parks).  foreign=False: all arrays from the pool, the stream never
    handed out - passes stay parked as long as the library allows.  foreign=True: two models' arrays are the caller's (hipMalloc) and
    are touched with raw HIP calls on the handed-out compute stream only."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import stress_passes

    summary, bad = stress_passes.run(seed=20260706, ops=300, foreign=foreign, verbose=False)
    assert not bad, f"{len(bad)} mismatching downloads, first: {bad[:3]} ({summary})"
    assert summary["float32_launches"] >= 150 and summary["downloads"] >= 18, summary
    assert summary["recycles"] >= 5 and summary["graph_replays"] >= 5, summary
    assert sum(summary["flagged_rows_per_model"]) >= 50, summary   # there ARE rows the passes have to rewrite


@pytest.mark.gpu
def test_in_place_float64_rows_without_a_list(tables):
    """The float32 kernels hand their ill-conditioned rows to the float64 pass through a list; a launch of 2^32 rows or more, or a
    list that cannot be allocated, has none and re-evaluates such rows INSIDE the float32 kernel (mp_cold_rows: per-joint state in
    the wave's LDS slice, csrc/mp_bodies.h).  MANIPULAPY_HIP_HARD_PASS=0 takes that path for every launch (read once per process,
    hence the child): given rows and generated rows, generic and specialised kernels, a row count that ends in a partial wave -
    every row inside the float32 bound, the flagged ones at <= 0.2 x (they are float64 results), against the pinned C oracle."""
    import subprocess
    import sys
    import textwrap

    code = textwrap.dedent("""
        import os, sys
        import numpy as np
        sys.path.insert(0, %r)
        from manipulapy_amd import _hip
        from oracle import ref_numpy as ref, c_oracle
        tab = ref.load_tables(os.path.join(%r, "manipulapy_amd", "data", "model_ur5.npz"))
        lim = tab.joint_limits
        rng = np.random.default_rng(17)
        B, N = 70, 1001
        s_ = rng.uniform(lim[:, 0], lim[:, 1], (B, 6)).astype(np.float32)
        e_ = rng.uniform(lim[:, 0], lim[:, 1], (B, 6)).astype(np.float32)
        o = ref.batch_joint_trajectory(lim, s_, e_, 2.0, N, 5)
        q, qd, qdd = (np.ascontiguousarray(o[k].reshape(-1, 6), dtype=np.float32) for k in ("positions", "velocities", "accelerations"))
        want = c_oracle.inverse_dynamics_rows(tab, q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64))[0]
        tol = 1e-4 * np.abs(want) + 5e-6 * np.abs(want).max(axis=1, keepdims=True) + 1e-12
        ctx = _hip.HipContext(0)
        worst = {}
        for tag in ("generic", "specialised"):
            m = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, lim)
            if tag == "specialised":
                ctx.specialize(m)
            flagged = _hip.cpu_id_row_precision(m, q, qd, qdd).astype(bool)
            assert flagged.sum() > 100
            for name, got in (("rows", ctx.id_trajectory_host(m, q, qd, qdd, dtype=np.float32)),
                              ("fused", ctx.traj_id_fused_host(m, s_, e_, 2.0, N, 5).reshape(-1, 6))):
                r = np.abs(got - want) / tol
                # (the fused kernel regenerates its rows - within an ulp of these - so the flagged set is exact for the given rows only)
                assert np.isfinite(got).all() and r.max() <= 0.6 and (name == "fused" or r[flagged].max() <= 0.2), (tag, name, float(r.max()), float(r[flagged].max()))
                worst[tag + " " + name] = (round(float(r.max()), 3), round(float(r[flagged].max()), 4))
        ctx.destroy()
        print("OK", worst)
    """ % (ROOT, ROOT))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MANIPULAPY_HIP_HARD_PASS="0"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_bench_n2_line_end_to_end_with_two_ranks_sharing_the_gpu():
    """The N > 1 bench line has never met a multi-GPU node (the pool gives one GPU; the driver's 8-GPU run decides).  What CAN run
    here is its whole flow with two ranks sharing this GPU (MANIPULAPY_BENCH_SHARE_DEVICE=1; the figures mean nothing, the flow
    does): launcher -> gloo rendezvous -> weak-scaled headline with rank 0's host legs (CPU oracle timed, parity on its own shard,
    streaming probe) while rank 1 waits in the barrier -> both strong-scaled entries with THEIR oracle samples -> the RCCL phases,
    which either run or - RCCL refuses two ranks on one device - are reported in `collectives` without costing the line or the exit
    code (an N > 1 line without cpu_baseline / parity is unmeasured)."""
    import json
    import subprocess
    import sys

    env = dict(os.environ, PYTHONPATH=ROOT, MANIPULAPY_BENCH_SHARE_DEVICE="1", MANIPULAPY_BENCH_GATHER_TIMEOUT="90")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2"], env=env,
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and res.stdout.strip(), f"rc {res.returncode}\n{res.stdout[-1500:]}\n{res.stderr[-3000:]}"
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0 and list(line)[-1] == "summary"
    # the headline's host legs: rank 0, its own shard
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0 and "rank 0 of 2" in line["cpu_baseline"]["where"]
    assert line["parity_sample"]["ok"] and line["roofline"]["parity"]["rows_checked"] > 0 and line["roofline"]["parity"]["sets_checked"] == 1
    assert line["roofline"]["frac_of_probe"] > 0 and "elements_over_pure_rel" in line["roofline"]["parity"]
    # the strong-scaled entries: compute-only figures + an oracle sample each, whatever became of the collectives
    for name, key in (("c4_strong", "rows"), ("c5_strong", "trajectories")):
        e = line["configs"][name]
        assert "error" not in e, e.get("error")
        assert e["scaling"] == "strong" and e["n_gpus"] == 2 and e["value"] > 0 and sum(e["shard"]["trajectories_of_rank"]) == e["shard"]["B_total"]
        assert e["parity_sample"]["ok"] and e["roofline"]["parity"]["ok"] and e["roofline"]["parity"]["rows_checked"] > 0
        assert e["cpu_baseline"]["value"] > 0 and e["roofline"]["frac_of_probe"] > 0
        assert line["summary"][name]["parity_ok"] is True
    c = line["collectives"]
    assert c["ran"] or set(c["failed_in"]) <= {"c2", "c4_strong", "c5_strong"}
    if c["ran"]:   # a node where RCCL accepts the two ranks: then the reassembly must have been verified
        assert line["verified"] and all(line["configs"][k]["verified"] for k in ("c4_strong", "c5_strong"))


@pytest.mark.gpu
@pytest.mark.parametrize("specialise", [False, True])
def test_arrays_beyond_2_31_elements(specialise, tables):
    """Maximum sizes: rows x n beyond 2^31 ELEMENTS (8.7 GB per float32 array; 288 GB of HBM hold far more), where a 32-bit index,
    element offset or byte offset anywhere in the generation kernel, the inverse-dynamics kernels, the fused kernel, the row lists of
    the float64 pass or the copies would wrap: UR5, B = 362 000 x N = 1000 = 3.62e8 rows = 2.17e9 elements per array.  Slices at the
    start, on both sides of the 2^31st element, at 2^32 BYTES, and at the very end - of the generated histories against the oracle's
    time scaling, and of tau (two-step and fused) against the pinned C oracle under the float32 bound; the flagged rows among them at
    <= 0.2 x (their float64 pass found them through a list index beyond 2^28)."""
    from manipulapy_amd import _hip
    from oracle import c_oracle

    tab = tables["ur5"]
    lim = tab.joint_limits
    n, B, N = 6, 362_000, 1000
    rows = B * N
    assert rows * n > 2**31 and rows < 2**32
    ctx = _hip.HipContext(0)
    bufs = []
    try:
        m = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, lim)
        if specialise:
            ctx.specialize(m)
        rng = np.random.default_rng(2031)
        s_ = rng.uniform(lim[:, 0], lim[:, 1], (B, n)).astype(np.float32)
        e_ = rng.uniform(lim[:, 0], lim[:, 1], (B, n)).astype(np.float32)
        d_s, d_e = ctx.to_device(s_), ctx.to_device(e_)
        nb = rows * n * 4
        d_q, d_qd, d_qdd, d_tau, d_fused = (ctx.alloc(nb) for _ in range(5))
        bufs += [d_s, d_e, d_q, d_qd, d_qdd, d_tau, d_fused]
        ctx.batch_trajectory(m, d_s, d_e, B, N, 2.0, 5, d_q, d_qd, d_qdd)
        ctx.id_trajectory(m, d_q, d_qd, d_qdd, rows, d_tau, dtype=np.float32)
        ctx.traj_id_fused(m, d_s, d_e, B, N, 2.0, 5, d_fused)
        ctx.synchronize()
        K = 2048   # rows per slice
        firsts = [0, (2**31 // n) - K // 2, (2**32 // (n * 4)) - K // 2, rows // 2 + 12345, rows - K]
        for r0 in firsts:
            def rows_of(buf):
                out = np.empty((K, n), np.float32)
                _hip._check(ctx.lib.mp_memcpy_d2h(ctx.handle, out.ctypes.data, buf.offset(r0 * n * 4), out.nbytes))
                return out
            q, qd, qdd, tau, fused = (rows_of(b_) for b_ in (d_q, d_qd, d_qdd, d_tau, d_fused))
            # the generated rows: trajectory b, timestep t of every row of the slice, from the oracle's generator
            b_idx, t_idx = np.divmod(np.arange(r0, r0 + K), N)
            ub = np.unique(b_idx)
            o = ref.batch_joint_trajectory(lim, s_[ub], e_[ub], 2.0, N, 5)
            pick = (np.searchsorted(ub, b_idx), t_idx)
            for got, key in ((q, "positions"), (qd, "velocities"), (qdd, "accelerations")):
                want = o[key][pick]
                assert np.abs(got - want).max() <= 2e-6 * max(1.0, float(np.abs(want).max())), (r0, key)
            want = c_oracle.inverse_dynamics_rows(tab, q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64))[0]
            tol = 1e-4 * np.abs(want) + 5e-6 * np.abs(want).max(axis=1, keepdims=True) + 1e-12
            flagged = _hip.cpu_id_row_precision(m, q, qd, qdd).astype(bool)
            for name, got in (("two-step", tau), ("fused", fused)):
                ratio = np.abs(got - want) / tol
                assert np.isfinite(got).all() and ratio.max() <= 0.6, (r0, name, float(ratio.max()))
                if name == "two-step" and flagged.any():
                    assert ratio[flagged].max() <= 0.2, (r0, float(ratio[flagged].max()))
        # rows that went through the float64 pass, far beyond the 2^31st element: the ill-conditioned rows of the last 2^18 rows
        W = 1 << 18
        r0 = rows - W
        def window(buf):
            out = np.empty((W, n), np.float32)
            _hip._check(ctx.lib.mp_memcpy_d2h(ctx.handle, out.ctypes.data, buf.offset(r0 * n * 4), out.nbytes))
            return out
        q, qd, qdd, tau, fused = (window(b_) for b_ in (d_q, d_qd, d_qdd, d_tau, d_fused))
        hard = np.flatnonzero(_hip.cpu_id_row_precision(m, q, qd, qdd))[:4096]
        assert len(hard) > 100, len(hard)
        want = c_oracle.inverse_dynamics_rows(tab, q[hard].astype(np.float64), qd[hard].astype(np.float64), qdd[hard].astype(np.float64))[0]
        tol = 1e-4 * np.abs(want) + 5e-6 * np.abs(want).max(axis=1, keepdims=True) + 1e-12
        assert (np.abs(tau[hard] - want) / tol).max() <= 0.2 and (np.abs(fused[hard] - want) / tol).max() <= 0.6
    finally:
        ctx.synchronize()
        for b_ in bufs:
            b_.free()
        ctx.destroy()


@pytest.mark.gpu
def test_jacobian_array_beyond_2_31_elements(tables):
    """Maximum sizes, the long output rows: the space Jacobian of 5.2e7 iiwa14 rows is 2.18e9 float32 ELEMENTS (8.7 GB) and leaves
    through the wave-cooperative flat stores (mp_wave_store_flat: 16 rows staged, streamed out as 16-byte chunks across the row
    boundaries) - element and byte offsets beyond 2^31 / 2^32 in J.  FK + Jacobian + ID fused, float32,
    slices at the start, around J's 2^32nd byte and its 2^31st element, and at the very end (a partial wave: the
    row count is not a multiple of 64) against the NumPy oracle's FK / Jacobian and the C oracle's torques."""
    from manipulapy_amd import _hip
    from oracle import c_oracle

    tab = tables["iiwa14"]
    lim = tab.joint_limits
    n, B, N = 7, 52_013, 1000
    rows = B * N - 29                      # ends in a partial wave
    assert rows * 6 * n > 2**31 and rows % 64 != 0
    ctx = _hip.HipContext(0)
    bufs = []
    try:
        m = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, lim)
        rng = np.random.default_rng(2032)
        s_ = rng.uniform(lim[:, 0], lim[:, 1], (B, n)).astype(np.float32)
        e_ = rng.uniform(lim[:, 0], lim[:, 1], (B, n)).astype(np.float32)
        d_s, d_e = ctx.to_device(s_), ctx.to_device(e_)
        d_q, d_qd, d_qdd = (ctx.alloc(B * N * n * 4) for _ in range(3))
        d_T, d_J, d_tau = ctx.alloc(rows * 16 * 4), ctx.alloc(rows * 6 * n * 4), ctx.alloc(rows * n * 4)
        bufs += [d_s, d_e, d_q, d_qd, d_qdd, d_T, d_J, d_tau]
        ctx.batch_trajectory(m, d_s, d_e, B, N, 2.0, 5, d_q, d_qd, d_qdd)
        ctx.fk_jac_id(m, d_q, d_qd, d_qdd, rows, d_T, d_J, d_tau, dtype=np.float32)
        ctx.synchronize()
        K = 192
        for r0 in (0, 2**32 // (6 * n * 4) - K // 2, 2**31 // (6 * n) - K // 2, rows - K):   # J's 2^32nd byte, its 2^31st element, the tail
            def rows_of(buf, width):
                out = np.empty((K, width), np.float32)
                _hip._check(ctx.lib.mp_memcpy_d2h(ctx.handle, out.ctypes.data, buf.offset(r0 * width * 4), out.nbytes))
                return out
            q, qd, qdd = (rows_of(b_, n) for b_ in (d_q, d_qd, d_qdd))
            T, J, tau = rows_of(d_T, 16), rows_of(d_J, 6 * n), rows_of(d_tau, n)
            Tw = np.stack([ref.fk_space(tab, q[i].astype(np.float64)) for i in range(K)]).reshape(K, 16)
            Jw = np.stack([ref.jacobian_space(tab, q[i].astype(np.float64)) for i in range(K)]).reshape(K, 6 * n)
            assert np.abs(T - Tw).max() <= 2e-5 and np.abs(J - Jw).max() <= 2e-5, (r0, float(np.abs(T - Tw).max()), float(np.abs(J - Jw).max()))
            want = c_oracle.inverse_dynamics_rows(tab, q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64))[0]
            assert_f32(tau, want)
    finally:
        ctx.synchronize()
        for b_ in bufs:
            b_.free()
        ctx.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["batch_major", "time_major"])
def test_rollout_arrays_beyond_2_31_elements(layout, tables):
    """Maximum sizes, the roll-out: 3.7e6 xarm6 trajectories x 100 steps x 6 joints = 2.22e9 elements per array (8.9 GB), on both
    device layouts (batch-major: trajectory b's rows start at element 600 b - beyond 2^31 from b = 3.58e6; time-major: step t's
    rows start at 6 t B - beyond 2^31 from t = 97).  Every trajectory is driven by the torques that hold its start pose (uploaded
    step by step into the time-major array, transposed on the device for the batch-major run), no wrench; the first and the LAST 64
    trajectories over all 100 steps against the C roll-out oracle, float32, within 1e-4 of each array's scale."""
    from manipulapy_amd import _hip
    from oracle import c_oracle

    tab = tables["xarm6"]
    n, B, N = 6, 3_700_037, 100            # (not a multiple of 64: a ragged last wave)
    assert B * N * n > 2**31
    g = np.array([0.0, 0.0, -9.81])
    tmaj = layout == "time_major"
    ctx = _hip.HipContext(0)
    bufs = []
    try:
        m = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, tab.joint_limits)
        ctx.specialize(m)
        rng = np.random.default_rng(2033)
        th0 = rng.uniform(-0.5, 0.5, (B, n)).astype(np.float32)
        dth0 = np.zeros((B, n), np.float32)
        hold = ctx.id_trajectory_host(m, th0, dth0, dth0, g, None, dtype=np.float32)
        nb = B * N * n * 4
        d_th0, d_dth0, d_tau = ctx.to_device(th0), ctx.to_device(dth0), ctx.alloc(nb)
        outs = [ctx.alloc(nb) for _ in range(3)]
        bufs += [d_th0, d_dth0, d_tau] + outs
        for t in range(N):                 # time-major (N, B, n): step t's B rows are contiguous
            _hip._check(ctx.lib.mp_memcpy_h2d(ctx.handle, d_tau.offset(t * B * n * 4), hold.ctypes.data, hold.nbytes))
        if not tmaj:
            d_bm = ctx.alloc(nb)
            bufs.append(d_bm)
            ctx.transpose_rows(d_tau, N, B, n * 4, d_bm)      # -> (B, N, n)
            d_tau = d_bm
        ctx.fd_trajectory(m, d_th0, d_dth0, d_tau, None, B, N, g, 0.01, 1, *outs, dtype=np.float32, time_major=tmaj)
        ctx.synchronize()

        def trajectories(buf, b0, k):      # (k, N, n) of trajectories b0 .. b0 + k - 1
            if not tmaj:
                out = np.empty((k, N, n), np.float32)
                _hip._check(ctx.lib.mp_memcpy_d2h(ctx.handle, out.ctypes.data, buf.offset(b0 * N * n * 4), out.nbytes))
                return out
            out = np.empty((N, k, n), np.float32)
            for t in range(N):
                _hip._check(ctx.lib.mp_memcpy_d2h(ctx.handle, out[t].ctypes.data, buf.offset((t * B + b0) * n * 4), out[t].nbytes))
            return np.ascontiguousarray(np.swapaxes(out, 0, 1))

        for b0 in (0, B - 64):
            got = [trajectories(o, b0, 64) for o in outs]
            tm = np.repeat(hold[b0:b0 + 64, None, :].astype(np.float64), N, axis=1)
            want = c_oracle.fd_trajectory(tab, th0[b0:b0 + 64].astype(np.float64), dth0[b0:b0 + 64].astype(np.float64), tm, g, None, 0.01, 1)[:3]
            for name, x, y in zip(("positions", "velocities", "accelerations"), got, want):
                assert np.isfinite(x).all() and np.isfinite(y).all(), (b0, name)
                scale = max(float(np.abs(y).max()), 1e-3)
                drift = np.abs(x.astype(np.float64) - y).reshape(64, -1).max(axis=1) / scale
                assert drift.max() <= 1e-4, (b0, name, float(np.median(drift)), float(drift.max()))
            np.testing.assert_array_equal(got[0][:, 0], th0[b0:b0 + 64])   # row 0 = the initial state, bit for bit
    finally:
        ctx.synchronize()
        for b_ in bufs:
            b_.free()
        ctx.destroy()


@pytest.mark.gpu
def test_fused_launch_beyond_2_32_rows(tables):
    """Maximum sizes, the row count itself: the fused generation + inverse dynamics writes tau only, so ONE launch can hold more than
    2^32 rows in HBM - UR5, B = 4 300 000 x N = 1000 = 4.3e9 rows, a 103 GB torque array.  Such a launch has no row list (list
    entries are 32-bit: csrc/mp_capi.cpp, attach_hard_list) and re-evaluates its ill-conditioned rows in place; every index on the
    way (trajectory, timestep, row, byte offset) passes 2^32.  Slices at the start, on both sides of row 2^32, and at the very end
    against the pinned C oracle under the float32 bound."""
    from manipulapy_amd import _hip
    from oracle import c_oracle

    tab = tables["ur5"]
    lim = tab.joint_limits
    n, B, N = 6, 4_300_000, 1000
    rows = B * N
    assert rows > 2**32
    ctx = _hip.HipContext(0)
    bufs = []
    try:
        if ctx.properties()["total_memory"] < 150 * 2**30:
            pytest.skip("needs a 103 GB array")
        m = _hip.HipModel(tab.S, tab.Mcom, tab.G, tab.M_ee, lim)
        ctx.specialize(m)
        rng = np.random.default_rng(2034)
        s_ = rng.uniform(lim[:, 0], lim[:, 1], (B, n)).astype(np.float32)
        e_ = rng.uniform(lim[:, 0], lim[:, 1], (B, n)).astype(np.float32)
        d_s, d_e, d_tau = ctx.to_device(s_), ctx.to_device(e_), ctx.alloc(rows * n * 4)
        bufs += [d_s, d_e, d_tau]
        ctx.traj_id_fused(m, d_s, d_e, B, N, 2.0, 5, d_tau)
        ctx.synchronize()
        K = 2000                                                   # two whole trajectories per slice
        for b0 in (0, 2**32 // N - 1, B - 2):                      # trajectories; the middle pair straddles row 2^32
            got = np.empty((K, n), np.float32)
            _hip._check(ctx.lib.mp_memcpy_d2h(ctx.handle, got.ctypes.data, d_tau.offset(b0 * N * n * 4), got.nbytes))
            o = ref.batch_joint_trajectory(lim, s_[b0:b0 + 2], e_[b0:b0 + 2], 2.0, N, 5)
            q, qd, qdd = (o[k].reshape(-1, n).astype(np.float64) for k in ("positions", "velocities", "accelerations"))
            assert_f32(got, c_oracle.inverse_dynamics_rows(tab, q, qd, qdd)[0])
    finally:
        ctx.synchronize()
        for b_ in bufs:
            b_.free()
        ctx.destroy()


@pytest.mark.gpu
def test_the_ctypes_stub_of_integration_md_runs_as_written(tables):
    """INTEGRATION.md section 2 shows the binding a ManipulaPy maintainer would add (`hip_kernels/_ffi.py`).  The block is executed
    here exactly as printed - only the library's file name becomes its in-tree path - on a stand-in for the reference's
    ManipulatorDynamics (the attributes the stub reads: S_list, M_list, Glist, Mlist_per_link), and its two functions are held to the
    oracle: the documentation is under test."""
    import re
    import types
    from oracle import c_oracle

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. ctypes stub"):text.index("## 3. Hooking the seams")]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    assert 'ctypes.CDLL("libmanipula_hip.so")' in code
    code = code.replace('ctypes.CDLL("libmanipula_hip.so")', 'ctypes.CDLL(%r)' % os.path.join(ROOT, "manipulapy_amd", "libmanipula_hip.so"))
    ns = {}
    exec(compile(code, "INTEGRATION.md#2", "exec"), ns)
    tab = tables["ur5"]
    dyn = types.SimpleNamespace(S_list=tab.S, M_list=tab.M_ee, Glist=tab.G, Mlist_per_link=tab.Mcom)
    model = ns["model_from_dynamics"](dyn, tab.joint_limits, None)
    rng = np.random.default_rng(77)
    q, qd, qdd = (rng.uniform(-1, 1, (1000, 6)).astype(np.float32) for _ in range(3))
    g, F = np.array([0.0, 0.0, -9.81]), np.array([1.0, -2.0, 0.5, 3.0, -1.5, 0.75])
    tau = ns["inverse_dynamics_trajectory"](model, q, qd, qdd, g, F)
    assert_f32(tau, c_oracle.inverse_dynamics_rows(tab, q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64), g, F)[0])
    Bn, Nn = 5, 20
    th0, dth0 = rng.uniform(-0.5, 0.5, (Bn, 6)), np.zeros((Bn, 6))
    hold = c_oracle.inverse_dynamics_rows(tab, th0, dth0, dth0, g, np.zeros(6))[0]
    tm = np.repeat(hold[:, None, :], Nn, axis=1) + rng.uniform(-1e-3, 1e-3, (Bn, Nn, 6))
    Fm = rng.uniform(-0.02, 0.02, (Bn, Nn, 6))
    out = ns["forward_dynamics_trajectories"](model, th0, dth0, tm, g, Fm, 0.01, 1)
    want = c_oracle.fd_trajectory(tab, th0, dth0, tm, g, Fm, 0.01, 1)
    for k, key in enumerate(("positions", "velocities", "accelerations")):
        assert out[key].shape == (Bn, Nn, 6) and out[key].dtype == np.float32
        assert np.abs(out[key] - want[k]).max() <= 1e-4 * max(1.0, float(np.abs(want[k]).max())), key
