#!/usr/bin/env python3
"""Generate the committed golden fixtures by IMPORTING the reference (ManipulaPy v1.4.1).

Runs only in the build container, where /root/reference exists; the GPU box never
sees the reference, only the small .npz files this script writes next to itself.

    PYTHONHASHSEED=0 python tests/golden/make_golden.py

What it does:
  * creates a throw-away `numba` stub in a temp dir (numba is absent here; only
    ManipulaPy.planning / ManipulaPy.cuda_kernels import it, dynamics/kinematics do not);
  * re-executes itself with PYTHONHASHSEED=0 so the reference's set-ordered
    end-effector choice (urdf/core.py:461-462, :388-392) is pinned;
  * builds URDF -> SerialManipulator / ManipulatorDynamics through the reference's public
    consumer path (urdf_processor.py:82-138) for ur5 / iiwa14 / panda / xarm6 and dumps
      model_<robot>.npz    : the constant tables the hot path consumes
      dynamics_<robot>.npz : inputs + reference outputs at 25 configurations
                             (recipe of the reference's tests/test_dynamics_golden.py:107-182,
                             extended with random Ftip, FK, Jacobians, forward dynamics)
      trajectory_ur5.npz   : joint_trajectory / batch_joint_trajectory /
                             inverse_dynamics_trajectory / forward_dynamics_trajectory dumps
      cartesian_ur5.npz    : cartesian_trajectory dumps (`make_golden.py cartesian` regenerates only this)
      ik.npz               : iterative_inverse_kinematics dumps, 10 problems per robot (`make_golden.py ik`)
      control_ur5.npz      : ManipulatorController laws on UR5 (`make_golden.py control`)
      plan_ur5.npz         : OptimizedTrajectoryPlanning.plan_trajectory on the mesh-less UR5 (`make_golden.py plan`)
      urdf_api.npz         : URDFToSerialManipulator's convenience surface on ten urdf_suite files (`make_golden.py urdf_api`)
      gain_sweep_ur5.npz   : ManipulatorController.find_ultimate_gain_and_period on UR5 (`make_golden.py gain_sweep`)
      utils.npz            : every public ManipulaPy.utils function on generic and branch-switching inputs (`make_golden.py utils`)
      (manipulapy_amd/data/) model_<robot>.npz, urdf/<robot>.urdf : the four benchmark robots' tables and URDF skeletons
      legacy_dynamics.npz  : ManipulatorDynamics without Mlist_per_link + truncated / body-frame kinematics (`make_golden.py legacy`)
      urdf_suite.npz + urdf_suite/*.urdf : reference tables for all 28 database robots + the reference's URDF test fixtures
      reference_cpu_timings.json : cold-cache per-point timings of the reference

Nothing from /root/reference is copied: the fixtures hold numbers only.
dynamics_golden_{ur5,panda}.npz in this directory are the reference's OWN test data
files (tests/data/), kept byte-for-byte as the primary known-answer pin.
"""
import json
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
PKG_DATA = os.path.join(os.path.dirname(os.path.dirname(HERE)), "manipulapy_amd", "data")  # what the product ships
REF = "/root/reference"
SEED = 20260705  # same seed the reference's golden test uses (tests/test_dynamics_golden.py:53)

NUMBA_STUB = '''
def _ident(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f
njit = jit = vectorize = guvectorize = _ident
prange = range
class _Cfg: pass
config = _Cfg()
float32 = int32 = float64 = int64 = None
'''


def _reexec_pinned():
    if os.environ.get("_MP_GOLDEN_CHILD") == "1":
        return
    stub = tempfile.mkdtemp(prefix="mp_numba_stub_")
    os.makedirs(os.path.join(stub, "numba"))
    with open(os.path.join(stub, "numba", "__init__.py"), "w") as f:
        f.write(NUMBA_STUB)
    env = dict(os.environ)
    env.update(
        _MP_GOLDEN_CHILD="1",
        PYTHONHASHSEED="0",
        NUMBA_DISABLE_CUDA="1",
        MPLBACKEND="Agg",
        MANIPULAPY_QUIET="1",
        PYTHONPATH=os.pathsep.join([stub, REF, env.get("PYTHONPATH", "")]),
    )
    os.execve(sys.executable, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env)


_reexec_pinned()

import warnings  # noqa: E402

import numpy as np  # noqa: E402

warnings.simplefilter("ignore")

from ManipulaPy.ManipulaPy_data import get_robot_urdf  # noqa: E402
from ManipulaPy.urdf_processor import URDFToSerialManipulator  # noqa: E402

ROBOTS = ["ur5", "iiwa14", "panda", "xarm6"]
G_VEC = np.array([0.0, 0.0, -9.81])
FTIP_REF = np.array([1.0, -2.0, 0.5, 3.0, -1.5, 0.75])  # tests/test_dynamics_golden.py:145


def build(robot):
    proc = URDFToSerialManipulator(get_robot_urdf(robot), load_meshes=False)
    return proc, proc.serial_manipulator, proc.dynamics


def finite_limits(sm, n):
    lims = getattr(sm, "joint_limits", None) or [(None, None)] * n
    out = np.empty((n, 2))
    for i in range(n):
        lo, hi = lims[i] if i < len(lims) else (None, None)
        out[i, 0] = -np.pi if lo is None else float(lo)
        out[i, 1] = np.pi if hi is None else float(hi)
    return out


def clear_caches(dyn):
    dyn._mass_matrix_cache.clear()
    dyn._mass_matrix_derivative_cache.clear()


def dump_model(robot, proc, sm, dyn):
    n = sm.S_list.shape[1]
    ee = getattr(proc.robot, "end_effector_link", None)
    ee_name = getattr(ee, "name", str(ee))
    np.savez(
        os.path.join(PKG_DATA, f"model_{robot}.npz"),
        n=np.int64(n),
        S_list=np.asarray(sm.S_list, dtype=np.float64),
        B_list=np.asarray(sm.B_list, dtype=np.float64),
        M_ee=np.asarray(sm.M_list, dtype=np.float64),
        Glist=np.asarray(dyn.Glist, dtype=np.float64),
        Mlist_per_link=np.asarray(dyn.Mlist_per_link, dtype=np.float64),
        joint_limits=finite_limits(sm, n),
        ee_name=np.array(ee_name),
    )
    return n


def dump_dynamics(robot, sm, dyn, n, seed_off):
    rng = np.random.default_rng(SEED + seed_off)
    lims = finite_limits(sm, n)
    lo, hi = lims[:, 0], lims[:, 1]
    thetas = [np.zeros(n), lo.copy(), hi.copy(), np.full(n, 0.02)]
    while len(thetas) < 25:
        thetas.append(rng.uniform(lo, hi))
    thetas = np.array(thetas)
    K = len(thetas)
    dthetas = rng.uniform(-1, 1, (K, n))
    ddthetas = rng.uniform(-1, 1, (K, n))
    ftips = rng.uniform(-3, 3, (K, 6))
    ftips[:5] = 0.0
    ftips[5] = FTIP_REF
    out = {k: [] for k in ("mass_matrix", "inverse_dynamics", "gravity_forces",
                            "velocity_quadratic_forces", "forward_dynamics",
                            "fk_space", "fk_body", "jac_space", "jac_body")}
    for i in range(K):
        clear_caches(dyn)
        th, dth, ddth, ft = thetas[i], dthetas[i], ddthetas[i], ftips[i]
        out["mass_matrix"].append(np.asarray(dyn.mass_matrix(th)))
        out["velocity_quadratic_forces"].append(np.asarray(dyn.velocity_quadratic_forces(th, dth)))
        out["gravity_forces"].append(np.asarray(dyn.gravity_forces(th, G_VEC)))
        tau = np.asarray(dyn.inverse_dynamics(th, dth, ddth, G_VEC, ft))
        out["inverse_dynamics"].append(tau)
        # forward dynamics of the torques just computed -> should give ddth back
        out["forward_dynamics"].append(np.asarray(dyn.forward_dynamics(th, dth, tau, G_VEC, ft)))
        out["fk_space"].append(np.asarray(sm.forward_kinematics(th, frame="space")))
        out["fk_body"].append(np.asarray(sm.forward_kinematics(th, frame="body")))
        out["jac_space"].append(np.asarray(sm.jacobian(th, frame="space")))
        out["jac_body"].append(np.asarray(sm.jacobian(th, frame="body")))
    np.savez(
        os.path.join(HERE, f"dynamics_{robot}.npz"),
        thetas=thetas, dthetas=dthetas, ddthetas=ddthetas, g=G_VEC, ftips=ftips,
        **{k: np.array(v, dtype=np.float64) for k, v in out.items()},
    )


def dump_trajectories():
    """Planner-level dumps (UR5 + xarm6) through OptimizedTrajectoryPlanning(use_cuda=False)."""
    from ManipulaPy.cuda_kernels.trajectory_kernels import trajectory_cpu_fallback
    from ManipulaPy.planning import OptimizedTrajectoryPlanning

    proc, sm, dyn = build("ur5")
    n = sm.S_list.shape[1]
    lims = finite_limits(sm, n)
    planner = OptimizedTrajectoryPlanning(
        sm, get_robot_urdf("ur5"), dyn, lims.tolist(), use_cuda=False)
    rng = np.random.default_rng(SEED + 100)
    lo, hi = lims[:, 0], lims[:, 1]
    start = rng.uniform(lo, hi).astype(np.float32)
    end = rng.uniform(lo, hi).astype(np.float32)
    d = {"joint_limits": lims, "start": start, "end": end}

    # C1: UR5 quintic N=1000 (BASELINE.json configs[0]) + cubic N=500 + N=3 known answer
    for tag, (N, method, Tf) in {"q1000": (1000, 5, 2.0), "c500": (500, 3, 1.5)}.items():
        r = planner.joint_trajectory(start, end, Tf, N, method)
        for k in ("positions", "velocities", "accelerations"):
            d[f"jt_{tag}_{k}"] = np.asarray(r[k])
        d[f"jt_{tag}_args"] = np.array([N, method, Tf], dtype=np.float64)
        # NumPy twin (float32 linspace math), cuda_kernels/trajectory_kernels.py:20-88
        p, v, a = trajectory_cpu_fallback(start, end, Tf, N, method)
        d[f"np_{tag}_positions"], d[f"np_{tag}_velocities"], d[f"np_{tag}_accelerations"] = p, v, a
    # exceeding the limits on purpose -> exercises the clip (planning/trajectory.py:311-313)
    far = (hi + 0.5).astype(np.float32)
    r = planner.joint_trajectory(start, far, 1.0, 32, 5)
    d["jt_clip_end"] = far
    for k in ("positions", "velocities", "accelerations"):
        d[f"jt_clip_{k}"] = np.asarray(r[k])
    # linear / unsupported method on the NumPy twin (the numba one yields zeros for it)
    p, v, a = trajectory_cpu_fallback(start, end, 2.0, 16, 1)
    d["np_lin16_positions"], d["np_lin16_velocities"], d["np_lin16_accelerations"] = p, v, a

    # batch (B=5, N=16), planning/trajectory.py:335-502 (sequential CPU path)
    B = 5
    bs = rng.uniform(lo, hi, (B, n)).astype(np.float32)
    be = rng.uniform(lo, hi, (B, n)).astype(np.float32)
    r = planner.batch_joint_trajectory(bs, be, 2.0, 16, 5)
    d["batch_start"], d["batch_end"] = bs, be
    for k in ("positions", "velocities", "accelerations"):
        d[f"batch_{k}"] = np.asarray(r[k])

    # inverse_dynamics_trajectory, N=64 quintic, float64 inputs
    N = 64
    r = planner.joint_trajectory(start, end, 2.0, N, 5)
    q = np.asarray(r["positions"], dtype=np.float64)
    qd = np.asarray(r["velocities"], dtype=np.float64)
    qdd = np.asarray(r["accelerations"], dtype=np.float64)
    clear_caches(dyn)
    tau32 = np.asarray(planner.inverse_dynamics_trajectory(q, qd, qdd))
    tau64 = np.array([np.asarray(dyn.inverse_dynamics(q[i], qd[i], qdd[i], G_VEC, np.zeros(6)))
                      for i in range(N)])
    d["idt_q"], d["idt_qd"], d["idt_qdd"] = q, qd, qdd
    d["idt_tau_f32"], d["idt_tau_f64"] = tau32, tau64
    # with a wrench and tight torque limits -> clip behaviour (planning/trajectory_dynamics.py:369-373)
    tl = np.stack([-np.full(n, 20.0), np.full(n, 15.0)], axis=1)
    planner2 = OptimizedTrajectoryPlanning(
        sm, get_robot_urdf("ur5"), dyn, lims.tolist(), torque_limits=tl.tolist(), use_cuda=False)
    d["idt_torque_limits"] = tl
    d["idt_ftip"] = FTIP_REF
    d["idt_tau_f32_clip_ftip"] = np.asarray(
        planner2.inverse_dynamics_trajectory(q[:16], qd[:16], qdd[:16], G_VEC, FTIP_REF))
    np.savez(os.path.join(HERE, "trajectory_ur5.npz"), **d)

    # forward_dynamics_trajectory roll-out: xarm6, N=8, intRes=2
    proc, sm, dyn = build("xarm6")
    n = sm.S_list.shape[1]
    lims = finite_limits(sm, n)
    planner = OptimizedTrajectoryPlanning(
        sm, get_robot_urdf("xarm6"), dyn, lims.tolist(), use_cuda=False)
    rng = np.random.default_rng(SEED + 200)
    th0 = rng.uniform(-0.5, 0.5, n)
    dth0 = rng.uniform(-0.2, 0.2, n)
    N = 8
    taumat = rng.uniform(-1, 1, (N, n)) * 0.5
    Ftipmat = np.tile(FTIP_REF, (N, 1)) * rng.uniform(0.5, 1.0, (N, 1))
    r = planner.forward_dynamics_trajectory(th0, dth0, taumat, G_VEC, Ftipmat, 0.01, 2)
    np.savez(
        os.path.join(HERE, "fd_trajectory_xarm6.npz"),
        joint_limits=lims, theta0=th0, dtheta0=dth0, taumat=taumat, g=G_VEC,
        Ftipmat=Ftipmat, dt=np.float64(0.01), intRes=np.int64(2),
        positions=np.asarray(r["positions"]), velocities=np.asarray(r["velocities"]),
        accelerations=np.asarray(r["accelerations"]),
    )


def dump_cartesian():
    """cartesian_trajectory (planning/trajectory.py:504-594) dumps: generic pose pair, near-identity, near-pi and
    exact half-turn rotations, cubic + quintic + 'other' method."""
    from ManipulaPy.planning import OptimizedTrajectoryPlanning

    proc, sm, dyn = build("ur5")
    lims = finite_limits(sm, sm.S_list.shape[1])
    planner = OptimizedTrajectoryPlanning(sm, get_robot_urdf("ur5"), dyn, lims.tolist(), use_cuda=False)
    rng = np.random.default_rng(SEED + 400)

    def rot(axis, ang):
        axis = np.asarray(axis, float) / np.linalg.norm(axis)
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)

    Xs = np.asarray(sm.forward_kinematics(rng.uniform(-2, 2, 6)))
    cases = {"generic": np.asarray(sm.forward_kinematics(rng.uniform(-2, 2, 6)))}
    for tag, (axis, ang) in {"tiny": ([1, 2, 3], 1e-5), "small": ([0, 1, 1], 5e-3), "nearpi": ([1, -1, 0.5], np.pi - 1e-3),
                             "pi_z": ([0, 0, 1], np.pi), "pi_x": ([1, 0, 0], np.pi), "pi_gen": ([1, 2, -1], np.pi)}.items():
        X = np.eye(4)
        X[:3, :3] = Xs[:3, :3] @ rot(axis, ang)
        X[:3, 3] = Xs[:3, 3] + rng.uniform(-0.3, 0.3, 3)
        cases[tag] = X
    d = {"Xstart": Xs}
    for tag, Xe in cases.items():
        d[f"{tag}_Xend"] = Xe
        for method in (3, 5, 1):
            r = planner.cartesian_trajectory(Xs, Xe, 2.0, 21, method)
            for k in ("positions", "velocities", "accelerations", "orientations"):
                d[f"{tag}_m{method}_{k}"] = np.asarray(r[k])
    np.savez(os.path.join(HERE, "cartesian_ur5.npz"), **d)


def dump_ik():
    """iterative_inverse_kinematics (kinematics/ik.py:39-311) with its default flags, for the four robots: targets are
    FK of random in-limit configurations, initial guesses are that configuration plus a perturbation (small: converges in
    a few steps; large: tens of iterations, step cap + joint-limit clip active), plus one unreachable target per robot
    that exhausts a small iteration budget without ever stalling for 20 iterations (so the reference's random
    restart, which draws from NumPy's global stream, never fires and the dump is deterministic)."""
    d = {}
    for r, robot in enumerate(ROBOTS):
        proc, sm, dyn = build(robot)
        n = sm.S_list.shape[1]
        lims = finite_limits(sm, n)
        rng = np.random.default_rng(SEED + 500 + r)
        T_des, th0, th, ok, it, params = [], [], [], [], [], []
        for case in range(10):
            q_true = rng.uniform(0.6 * lims[:, 0], 0.6 * lims[:, 1])
            T = np.asarray(sm.forward_kinematics(q_true), dtype=np.float64)
            spread = (0.05, 0.3, 0.8)[case % 3]
            q0 = np.clip(q_true + rng.uniform(-spread, spread, n), lims[:, 0], lims[:, 1])
            kw = dict(eomg=1e-6, ev=1e-6, max_iterations=400, damping=2e-2, step_cap=0.3)
            if case == 9:  # unreachable: 3 m away, tiny budget
                T = T.copy(); T[:3, 3] += np.array([3.0, 0.0, 0.0])
                kw["max_iterations"] = 15
            if case in (4, 7):  # non-default weights / damping / cap
                kw.update(damping=5e-2, step_cap=0.15, weight_orientation=0.5, weight_position=2.0)
            if case in (2, 5, 9):  # the two options robust_inverse_kinematics switches on
                kw.update(adaptive_tuning=True, backtracking=True)
            if case == 3:
                kw.update(adaptive_tuning=True)
            if case == 6:
                kw.update(backtracking=True)
            np.random.seed(1234)
            sol, success, iters = sm.iterative_inverse_kinematics(T, q0, **kw)
            T_des.append(T); th0.append(q0); th.append(np.asarray(sol, dtype=np.float64)); ok.append(bool(success)); it.append(int(iters))
            params.append([kw["eomg"], kw["ev"], kw["max_iterations"], kw["damping"], kw["step_cap"],
                           kw.get("weight_orientation", 1.0), kw.get("weight_position", 1.0),
                           float(kw.get("adaptive_tuning", False)), float(kw.get("backtracking", False))])
        d[f"{robot}_T_desired"] = np.stack(T_des); d[f"{robot}_theta0"] = np.stack(th0); d[f"{robot}_theta"] = np.stack(th)
        d[f"{robot}_success"] = np.array(ok); d[f"{robot}_iterations"] = np.array(it); d[f"{robot}_params"] = np.array(params)
        d[f"{robot}_joint_limits"] = np.array([[-np.inf if lo is None else lo, np.inf if hi is None else hi]
                                              for lo, hi in sm.joint_limits], dtype=np.float64)
        print(robot, "ik:", ok, it, flush=True)
        # robust_inverse_kinematics (kinematics/ik.py:477-598): multi-start, adaptive tuning + backtracking, np.random guesses
        Tr, thr, okr, itr, names = [], [], [], [], []
        for case in range(4):
            q_true = rng.uniform(0.8 * lims[:, 0], 0.8 * lims[:, 1])
            T = np.asarray(sm.forward_kinematics(q_true), dtype=np.float64)
            if case == 3:
                T = T.copy(); T[:3, 3] += np.array([2.5, 0.0, 0.0])  # unreachable: every attempt fails
            np.random.seed(4321 + case)
            sol, success, iters, name = sm.robust_inverse_kinematics(T, max_attempts=10 if case < 3 else 4, max_iterations=300)
            Tr.append(T); thr.append(np.asarray(sol, dtype=np.float64)); okr.append(bool(success)); itr.append(int(iters)); names.append(name)
        d[f"{robot}_robust_T_desired"] = np.stack(Tr); d[f"{robot}_robust_theta"] = np.stack(thr)
        d[f"{robot}_robust_success"] = np.array(okr); d[f"{robot}_robust_iterations"] = np.array(itr)
        d[f"{robot}_robust_strategy"] = np.array(names)
        print(robot, "robust ik:", okr, itr, names, flush=True)
        if robot == "ur5":  # smart_inverse_kinematics + its helpers (kinematics/ik.py:327-475, kinematics/ik_helpers.py)
            from ManipulaPy import utils as ref_utils
            from ManipulaPy.kinematics import ik_helpers as ref_h

            def rot(axis, ang):
                axis = np.asarray(axis, float) / np.linalg.norm(axis)
                K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
                return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)

            Ts = []
            for axis, ang in (([1, 2, 3], 0.7), ([0, 1, 1], 1e-5), ([1, -1, 0.5], 0.05), ([0, 0, 1], 2.5), ([1, 0, 0], np.pi - 1e-3),
                              ([1, 2, -1], np.pi), ([0, 0, 1], 0.0)):
                T = np.eye(4); T[:3, :3] = rot(axis, ang); T[:3, 3] = rng.uniform(-0.5, 0.5, 3)
                Ts.append(T)
            d["log6_T"] = np.stack(Ts)
            d["log6_V"] = np.stack([np.asarray(ref_utils.se3ToVec(ref_utils.MatrixLog6(T))) for T in Ts])
            qc = rng.uniform(0.5 * lims[:, 0], 0.5 * lims[:, 1], (4, n))
            qn = qc + rng.uniform(-0.15, 0.15, (4, n))
            Tc = np.stack([np.asarray(sm.forward_kinematics(q)) for q in qc]); Tn = np.stack([np.asarray(sm.forward_kinematics(q)) for q in qn])
            d["ext_theta"], d["ext_Tc"], d["ext_Tn"] = qc, Tc, Tn
            d["ext_guess"] = np.stack([np.asarray(ref_h.extrapolate_from_current(qc[i], Tc[i], Tn[i], lambda th: sm.jacobian(th, frame="space"),
                                                                             sm.joint_limits, alpha=0.5)) for i in range(4)])
            cache = ref_h.IKInitialGuessCache(max_size=3)
            for i in range(4):  # the first entry is evicted
                cache.add(Tc[i], qc[i], residual=[None, 0.5, 1e-4, 0.02][i])
            d["cache_query"] = Tn
            d["cache_out"] = np.stack([np.asarray(cache.get_nearest(Tn[i], k=3, joint_limits=sm.joint_limits)) for i in range(4)])
            smart = {}
            for j, strat in enumerate(("workspace_heuristic", "midpoint", "extrapolate", "cached", "random")):
                np.random.seed(777 + j)
                kw = dict(max_iterations=250)
                if strat == "extrapolate":
                    kw.update(theta_current=qc[1], T_current=Tc[1])
                if strat == "cached":
                    kw.update(cache=cache)
                th_s, ok_s, it_s = sm.smart_inverse_kinematics(Tn[1], strategy=strat, **kw)
                smart[strat] = (np.asarray(th_s, dtype=np.float64), bool(ok_s), int(it_s))
                print("smart", strat, ok_s, it_s, flush=True)
            d["smart_target"] = Tn[1]
            for k_, (th_s, ok_s, it_s) in smart.items():
                d[f"smart_{k_}_theta"], d[f"smart_{k_}_success"], d[f"smart_{k_}_iterations"] = th_s, np.array(ok_s), np.array(it_s)
            np.random.seed(4242)
            Tbad = Tn[2].copy(); Tbad[:3, 3] += np.array([2.5, 0, 0])
            th_s, ok_s, it_s = sm.smart_inverse_kinematics(Tbad, max_iterations=40)
            d["smart_unreachable_target"], d["smart_unreachable_theta"] = Tbad, np.asarray(th_s, dtype=np.float64)
            d["smart_unreachable_success"], d["smart_unreachable_iterations"] = np.array(bool(ok_s)), np.array(int(it_s))
    np.savez(os.path.join(HERE, "ik.npz"), **d)


def dump_control():
    """ManipulatorController laws (control/pid.py, computed_torque.py, robust_adaptive.py) on UR5, two consecutive calls each
    where the controller carries state (integral, parameter estimate)."""
    from ManipulaPy.control import ManipulatorController

    proc, sm, dyn = build("ur5")
    rng = np.random.default_rng(SEED + 600)
    n = 6
    d = {}
    u = lambda lo, hi, *shape: rng.uniform(lo, hi, shape if shape else (n,))
    st = {k: u(-1, 1) for k in ("qd_des", "q", "dq_des", "dq", "ddq_des", "ddq")}
    st.update(Kp=u(5, 50), Ki=u(0.1, 2), Kd=u(0.5, 5), x_des=u(-0.5, 0.5, 3), Kp3=u(5, 50, 3), Kd3=u(0.5, 5, 3),
              Kp33=u(-5, 5, 3, 3), Kd33=u(-1, 1, 3, 3), dist=u(-1, 1), gain=u(0.1, 2), merr=u(-0.1, 0.1),
              jl=np.stack([u(-2, -0.2), u(0.2, 2)], axis=1), tl=np.stack([u(-30, -5), u(5, 30)], axis=1), tau_in=u(-60, 60))
    d.update({f"in_{k}": v for k, v in st.items()})
    c = ManipulatorController(dyn)
    d["pd"] = np.asarray(c.pd_control(st["qd_des"], st["dq_des"], st["q"], st["dq"], st["Kp"], st["Kd"]))
    d["pid_1"] = np.asarray(c.pid_control(st["qd_des"], st["dq_des"], st["q"], st["dq"], 0.01, st["Kp"], st["Ki"], st["Kd"]))
    d["pid_2"] = np.asarray(c.pid_control(st["qd_des"], st["dq_des"], st["q"], st["dq"], 0.01, st["Kp"], st["Ki"], st["Kd"], i_clamp=0.015))
    d["pdff"] = np.asarray(c.pd_feedforward_control(st["qd_des"], st["dq_des"], st["ddq_des"], st["q"], st["dq"], st["Kp"], st["Kd"], G_VEC, FTIP_REF))
    lim = c.enforce_limits(st["q"], st["dq"], st["tau_in"], st["jl"], st["tl"])
    d["lim_q"], d["lim_dq"], d["lim_tau"] = (np.asarray(x) for x in lim)
    d["jsc"] = np.asarray(c.joint_space_control(st["qd_des"], st["q"], st["dq"], st["Kp"], st["Kd"]))
    d["csc_vec"] = np.asarray(c.cartesian_space_control(st["x_des"], st["q"], st["dq"], st["Kp3"], st["Kd3"]))
    d["csc_mat"] = np.asarray(c.cartesian_space_control(st["x_des"], st["q"], st["dq"], st["Kp33"], st["Kd33"]))
    d["robust"] = np.asarray(c.robust_control(st["q"], st["dq"], st["ddq"], G_VEC, FTIP_REF, st["dist"], st["gain"]))
    c2 = ManipulatorController(dyn)
    d["adaptive_1"] = np.asarray(c2.adaptive_control(st["q"], st["dq"], st["ddq"], G_VEC, FTIP_REF, st["merr"], 0.7))
    d["adaptive_2"] = np.asarray(c2.adaptive_control(st["q"], st["dq"], st["ddq"], G_VEC, FTIP_REF, st["merr"], 0.7))
    c3 = ManipulatorController(dyn)
    d["ctc_1"] = np.asarray(c3.computed_torque_control(st["qd_des"], st["dq_des"], st["ddq_des"], st["q"], st["dq"], G_VEC, 0.01, st["Kp"], st["Ki"], st["Kd"]))
    d["ctc_2"] = np.asarray(c3.computed_torque_control(st["qd_des"], st["dq_des"], st["ddq_des"], st["q"], st["dq"], G_VEC, 0.01, st["Kp"], st["Ki"], st["Kd"]))
    d["ff"] = np.asarray(c3.feedforward_control(st["qd_des"], st["dq_des"], st["ddq_des"], G_VEC, FTIP_REF))
    # Kalman filter (control/kalman.py) and response metrics / Ziegler-Nichols (control/metrics.py)
    ck = ManipulatorController(dyn)
    Qn, Rn = np.eye(12) * 1e-3, np.eye(12) * 1e-2
    tau_k = u(-5, 5)
    kal = []
    for step in range(3):
        qf, dqf = ck.kalman_filter_control(st["qd_des"], st["dq_des"], st["q"] + 0.01 * step, st["dq"], tau_k, G_VEC, FTIP_REF, 0.01, Qn, Rn)
        kal.append(np.concatenate((np.asarray(qf), np.asarray(dqf))))
    d["kalman_tau"], d["kalman_x"], d["kalman_P"] = tau_k, np.stack(kal), np.asarray(ck.P)
    tt_ = np.linspace(0, 5, 251)
    resp = 1.0 - np.exp(-1.2 * tt_) * np.cos(4.0 * tt_)
    d["metric_t"], d["metric_y"] = tt_, resp
    d["metric_values"] = np.array([c.calculate_rise_time(tt_, resp, 1.0), c.calculate_percent_overshoot(resp, 1.0),
                                   c.calculate_settling_time(tt_, resp, 1.0), c.calculate_settling_time(tt_, resp, 1.0, 0.2),
                                   c.calculate_steady_state_error(resp, 1.0), c.calculate_rise_time(tt_, resp, 5.0),
                                   c.calculate_settling_time(tt_, resp * 0 + 3, 1.0)])
    d["zn"] = np.array([list(c.ziegler_nichols_tuning(8.0, 0.5, k)) for k in ("P", "PI", "PID")])
    d["zn_vec"] = np.stack([np.asarray(x) for x in c.ziegler_nichols_tuning(np.array([8.0, 4.0]), np.array([0.5, 0.25]), "PID")])
    # small kinematics helpers of SerialManipulator (kinematics/fk.py:88-104, kinematics/velocity.py:65-89)
    qs = rng.uniform(-2, 2, (6, n))
    Vs = rng.uniform(-1, 1, (6, 6))
    d["kin_q"], d["kin_V"] = qs, Vs
    d["kin_pose"] = np.stack([np.asarray(sm.end_effector_pose(q)) for q in qs])
    d["kin_jvel_space"] = np.stack([np.asarray(sm.joint_velocity(q, V)) for q, V in zip(qs, Vs)])
    d["kin_jvel_body"] = np.stack([np.asarray(sm.joint_velocity(q, V, frame="body")) for q, V in zip(qs, Vs)])
    np.savez(os.path.join(HERE, "control_ur5.npz"), **d)


def dump_plan():
    """OptimizedTrajectoryPlanning.plan_trajectory (planning/collision_host.py:90-152) on the UR5 skeleton URDF (no meshes: the collision
    checker never reports a collision, so every waypoint takes exactly one potential-field step when obstacles are given), and
    PotentialField's own values at a few points."""
    from ManipulaPy.path_planning import OptimizedTrajectoryPlanning
    from ManipulaPy.potential_field import PotentialField

    proc, sm, dyn = build("ur5")
    urdf = os.path.join(PKG_DATA, "urdf", "ur5.urdf")
    lim = finite_limits(sm, 6)
    pl = OptimizedTrajectoryPlanning(sm, urdf, dyn, lim, use_cuda=False)
    assert pl.collision_checker is not None and pl.potential_field is not None
    rng = np.random.default_rng(SEED + 950)
    d = {}
    start, target = rng.uniform(-1, 1, 6), rng.uniform(-1, 1, 6)
    obstacles = [rng.uniform(-1, 1, 6) for _ in range(4)] + [0.5 * (start + target) + 0.05]
    d["start"], d["target"], d["obstacles"] = start, target, np.stack(obstacles)
    d["with_obstacles"] = np.asarray(pl.plan_trajectory(start.tolist(), target.tolist(), obstacles))
    d["without_obstacles"] = np.asarray(pl.plan_trajectory(start.tolist(), target.tolist(), []))
    d["collision_free"] = np.array([bool(pl.collision_checker.check_collision(q)) for q in (start, target, np.zeros(6))])
    pl2 = OptimizedTrajectoryPlanning(sm, "nonexistent.urdf", dyn, lim, use_cuda=False)
    d["no_checker"] = np.array([pl2.collision_checker is None, pl2.potential_field is None])
    d["no_checker_plan"] = np.asarray(pl2.plan_trajectory(start.tolist(), target.tolist(), obstacles))
    np.savez(os.path.join(HERE, "plan_ur5.npz"), **d)


def dump_urdf_api():
    """The convenience surface of URDFToSerialManipulator (urdf_processor.py:363-617) on a few of the urdf_suite skeletons
    (the SAME files the build reads): link / joint names, all-link forward kinematics at one configuration and for a batch,
    transforms between links, the validation report."""
    d = {}
    names = ["ur5", "panda", "xarm6_gripper", "robotiq_2f_85", "jaco_6dof", "fixture_branched", "fixture_mimic_joints", "fixture_multi_root",
             "fixture_prismatic_joint", "fixture_continuous_joints"]
    rng = np.random.default_rng(SEED + 900)
    for name in names:
        proc = URDFToSerialManipulator(os.path.join(HERE, "urdf_suite", f"{name}.urdf"), load_meshes=False)
        n = proc.num_dofs
        info = proc.print_joint_info()
        d[f"{name}__num_dofs"] = np.array(n); d[f"{name}__joint_names"] = np.array(proc.joint_names)
        d[f"{name}__link_names"] = np.array(proc.link_names); d[f"{name}__ee"] = np.array(proc.end_effector_name)
        d[f"{name}__all_joint_names"] = np.array(info["joint_names"]); d[f"{name}__limits"] = np.asarray(proc.joint_limits_array, dtype=np.float64)
        cfg = rng.uniform(-1.0, 1.0, n)
        cfgs = rng.uniform(-1.0, 1.0, (4, n))
        fk = proc.link_fk(cfg)
        order = list(fk.keys())
        d[f"{name}__cfg"], d[f"{name}__cfgs"] = cfg, cfgs
        d[f"{name}__fk_links"] = np.array(order); d[f"{name}__fk"] = np.stack([np.asarray(fk[k]) for k in order])
        fkb = proc.batch_forward_kinematics(cfgs)
        d[f"{name}__fkb"] = np.stack([np.asarray(fkb[k]) for k in order])
        d[f"{name}__ee_batch"] = np.asarray(proc.get_end_effector_transforms(cfgs))
        a, b = order[-1], order[len(order) // 2]
        d[f"{name}__tf_pair"] = np.array([a, b])
        d[f"{name}__tf"] = np.asarray(proc.get_transform(a, b, cfg)); d[f"{name}__tf_world"] = np.asarray(proc.get_transform(a, "world", cfg))
        d[f"{name}__fk_current"] = np.stack([np.asarray(v) for v in proc.link_fk(None).values()])   # the configuration persists
        v = proc.validate()
        d[f"{name}__valid"] = np.array(bool(v["valid"]))
        d[f"{name}__issues"] = np.array([f"{i['severity']}|{i['message']}" for i in v["issues"]] or ["<none>"])
        d[f"{name}__fk_ee"] = np.asarray(proc.forward_kinematics(cfg)); d[f"{name}__jac"] = np.asarray(proc.jacobian(cfg))
        print(name, n, len(order), bool(v["valid"]), len(v["issues"]), flush=True)
    d["w_p_in_w"], d["w_p_in_p"] = rng.uniform(-1, 1, (4, 3)), rng.uniform(-1, 1, (4, 3))
    d["w_p_out"] = np.asarray(URDFToSerialManipulator.w_p_to_slist(d["w_p_in_w"], d["w_p_in_p"], 4))
    np.savez(os.path.join(HERE, "urdf_api.npz"), **d)


def dump_gain_sweep():
    """ManipulatorController.find_ultimate_gain_and_period (control/metrics.py:280-366) on UR5: the closed-loop P-control
    simulations of the gain ladder 0.01 * 1.1^k - ultimate gain / period, the gains visited and every run's error history."""
    from ManipulaPy.control import ManipulatorController

    proc, sm, dyn = build("ur5")
    d = {}
    cases = {"a": (np.full(6, 0.1), np.full(6, 0.5), 0.01, 40),
             "b": (np.array([0.3, -1.2, 1.0, -0.4, 0.2, 0.1]), np.array([0.35, -1.25, 1.05, -0.45, 0.25, 0.05]), 0.002, 25),
             "c": (np.array([0.0, -1.5707963, 0.0, 0.0, 0.0, 0.0]), np.array([0.2, -1.4, 0.1, 0.1, -0.1, 0.05]), 0.005, 30)}
    for tag, (th, des, dt, steps) in cases.items():
        clear_caches(dyn)
        t0 = time.time()
        Ku, Tu, gains, errs = ManipulatorController(dyn).find_ultimate_gain_and_period(th.copy(), des.copy(), dt, steps)
        d[f"{tag}_theta"], d[f"{tag}_des"], d[f"{tag}_dt"], d[f"{tag}_steps"] = th, des, dt, steps
        d[f"{tag}_Ku"], d[f"{tag}_Tu"], d[f"{tag}_gains"] = Ku, Tu, np.asarray(gains)
        d[f"{tag}_errors"] = np.stack([np.asarray(e) for e in errs])
        print(tag, "gains visited", len(gains), "Ku", Ku, "Tu", Tu, f"{time.time() - t0:.1f}s", flush=True)
    np.savez(os.path.join(HERE, "gain_sweep_ur5.npz"), **d)


def dump_utils():
    """ManipulaPy.utils (so3 / se3 / screw / time_scaling): every public function on generic inputs and on the inputs where its
    branches switch (identity, tiny angles on both sides of each Taylor band, near-pi, exact half turns about several axes,
    Euler gimbal lock, prismatic screws)."""
    from ManipulaPy import utils as U

    rng = np.random.default_rng(SEED + 700)

    def rot(axis, ang):
        axis = np.asarray(axis, float) / np.linalg.norm(axis)
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)

    angles = [0.0, 1e-9, 5e-7, 2e-6, 1e-4, 9e-3, 1.1e-2, 0.09, 0.11, 0.7, 2.5, np.pi - 2e-2, np.pi - 5e-3, np.pi - 1e-7, np.pi]
    axes = [[0, 0, 1], [1, 0, 0], [0, 1, 0], [1, 2, 3], [1, -1, 0.5], [-2, 0.1, 0.3]]
    Rs = np.stack([rot(axes[(i + j) % len(axes)], a) for i, a in enumerate(angles) for j in range(2)])
    d = {"R": Rs}
    d["MatrixLog3"] = np.stack([np.asarray(U.MatrixLog3(R)) for R in Rs])
    ax_ang = [U.rotation_logm(R) for R in Rs]
    d["rotation_logm_axis"] = np.stack([np.asarray(a) for a, _ in ax_ang]); d["rotation_logm_angle"] = np.array([float(t) for _, t in ax_ang])
    d["euler"] = np.stack([np.asarray(U.rotation_matrix_to_euler_angles(R)) for R in Rs])
    gimbal = rot([0, 1, 0], np.pi / 2) @ rot([1, 0, 0], 0.3)
    d["R_gimbal"], d["euler_gimbal"] = gimbal, np.asarray(U.rotation_matrix_to_euler_angles(gimbal))
    eul = rng.uniform(-170, 170, (5, 3))
    d["euler_deg"], d["euler_to_R"] = eul, np.stack([np.asarray(U.euler_to_rotation_matrix(e)) for e in eul])
    ws = np.stack([np.asarray(ax, float) / np.linalg.norm(ax) * a for ax in axes[:4] for a in (0.0, 5e-3, 1.5e-2, 0.8, 3.0)])
    d["w"] = ws
    d["MatrixExp3"] = np.stack([np.asarray(U.MatrixExp3(U.VecToso3(w))) for w in ws])
    d["skew"] = np.stack([np.asarray(U.skew_symmetric(w)) for w in ws])
    Vs = np.concatenate([ws, rng.uniform(-1, 1, ws.shape)], axis=1)
    d["V"] = Vs
    d["VecTose3"] = np.stack([np.asarray(U.VecTose3(V)) for V in Vs])
    d["MatrixExp6"] = np.stack([np.asarray(U.MatrixExp6(U.VecTose3(V))) for V in Vs])
    Ts = []
    for R in Rs:
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = rng.uniform(-0.6, 0.6, 3)
        Ts.append(T)
    Ts = np.stack(Ts)
    d["T"] = Ts
    d["MatrixLog6"] = np.stack([np.asarray(U.MatrixLog6(T)) for T in Ts])
    d["logm"] = np.stack([np.asarray(U.logm(T)) for T in Ts])
    d["se3ToVec"] = np.stack([np.asarray(U.se3ToVec(U.MatrixLog6(T))) for T in Ts])
    d["logm_to_twist"] = np.stack([np.asarray(U.logm_to_twist(U.MatrixLog6(T))) for T in Ts])
    d["TransInv"] = np.stack([np.asarray(U.TransInv(T)) for T in Ts])
    d["adjoint"] = np.stack([np.asarray(U.adjoint_transform(T)) for T in Ts])
    S = np.stack([np.concatenate([np.asarray(ax, float) / np.linalg.norm(ax), rng.uniform(-1, 1, 3)]) for ax in axes] +
                 [np.array([0, 0, 0, 0.6, 0, 0.8])])
    th = rng.uniform(-3, 3, len(S))
    d["S"], d["theta"] = S, th
    d["transform_from_twist"] = np.stack([np.asarray(U.transform_from_twist(s, t)) for s, t in zip(S, th)])
    Slist = S.T  # (6, n)
    d["extract_r_list"] = np.asarray(U.extract_r_list(Slist))
    d["extract_omega_list"] = np.asarray(U.extract_omega_list(Slist))
    om, rr = Slist[:3, :], np.asarray(U.extract_r_list(Slist)).T
    d["screw_in_omega"], d["screw_in_r"] = om, rr
    d["extract_screw_list"] = np.asarray(U.extract_screw_list(om, rr))
    d["extract_screw_list_flat"] = np.asarray(U.extract_screw_list(om.reshape(-1), rr.reshape(-1)))
    d["extract_screw_list_bcast"] = np.asarray(U.extract_screw_list(om, rr[:, :1]))
    # joint-space potential field (potential_field/fields.py:35-160)
    from ManipulaPy.potential_field import PotentialField
    pf = PotentialField(attractive_gain=1.3, repulsive_gain=80.0, influence_distance=0.6)
    qpf, goal = rng.uniform(-1, 1, 6), rng.uniform(-1, 1, 6)
    obstacles = [qpf + rng.uniform(-0.2, 0.2, 6), qpf + rng.uniform(-1.5, 1.5, 6), qpf.copy(), qpf + rng.uniform(-0.1, 0.1, 6)]
    d["pf_q"], d["pf_goal"], d["pf_obstacles"] = qpf, goal, np.stack(obstacles)
    d["pf_attractive"] = np.asarray(pf.compute_attractive_potential(qpf, goal))
    d["pf_repulsive"] = np.asarray(pf.compute_repulsive_potential(qpf, obstacles))
    d["pf_gradient"] = np.asarray(pf.compute_gradient(qpf, goal, obstacles))
    d["pf_gradient_free"] = np.asarray(pf.compute_gradient(qpf, goal, []))
    tt = np.linspace(0, 2.0, 9)
    d["t"] = tt
    d["cubic"] = np.array([U.CubicTimeScaling(2.0, t) for t in tt]); d["quintic"] = np.array([U.QuinticTimeScaling(2.0, t) for t in tt])
    np.savez(os.path.join(HERE, "utils.npz"), **d)


def write_skeleton(src_path, dst_path, label):
    """The kinematic + inertial skeleton of a URDF (robot description DATA; number strings kept verbatim so the tables stay
    bit-identical).  Visual / collision geometry, materials, mesh references, transmissions and gazebo blocks are dropped:
    manipulapy_amd.urdf never reads them."""
    import xml.etree.ElementTree as ET

    src = ET.parse(src_path).getroot()
    out = ET.Element("robot", {"name": src.get("name", label)})
    out.append(ET.Comment(f" kinematic + inertial skeleton of the {label} description; generated by tests/golden/make_golden.py "))
    for link in src.findall("link"):
        le = ET.SubElement(out, "link", {"name": link.get("name")})
        ine = link.find("inertial")
        if ine is not None:
            ie = ET.SubElement(le, "inertial")
            for tag in ("origin", "mass", "inertia"):
                e = ine.find(tag)
                if e is not None:
                    ET.SubElement(ie, tag, dict(e.attrib))
    for joint in src.findall("joint"):
        je = ET.SubElement(out, "joint", {"name": joint.get("name"), "type": joint.get("type", "fixed")})
        for tag in ("origin", "parent", "child", "axis", "limit", "mimic"):
            e = joint.find(tag)
            if e is not None:
                ET.SubElement(je, tag, dict(e.attrib))
    ET.indent(out, space="  ")
    ET.ElementTree(out).write(dst_path, encoding="utf-8", xml_declaration=True)


def dump_urdfs():
    """manipulapy_amd/data/urdf/<robot>.urdf: skeletons of the four benchmark robots' URDFs (shipped with the package)."""
    os.makedirs(os.path.join(PKG_DATA, "urdf"), exist_ok=True)
    for robot in ROBOTS:
        write_skeleton(get_robot_urdf(robot), os.path.join(PKG_DATA, "urdf", f"{robot}.urdf"), robot)


def dump_urdf_suite():
    """Reference tables for EVERY robot of the reference's database (ManipulaPy_data/__init__.py:44-310, 28 entries: UR3/5/10
    + e-series, Panda, iiwa7/14, Gen3, Jaco, Fanuc, CRX, IRB2400, xArm6 with and without gripper, Robotiq grippers) and for the
    URDF fixtures of the reference's own tests (tests/urdf_fixtures/: simple_arm, prismatic_joint, branched,
    continuous_joints, mimic_joints, multi_root, primitives, transmissions): urdf_suite/<name>.urdf skeletons +
    urdf_suite.npz with S_list, B_list, M, Glist, Mlist_per_link, joint limits, end-effector name, for the default
    (seed-pinned) end effector and, for branching trees, for every other leaf as tip_link."""
    from ManipulaPy.ManipulaPy_data import ROBOT_DATABASE
    from ManipulaPy.urdf import URDF

    out_dir = os.path.join(HERE, "urdf_suite")
    os.makedirs(out_dir, exist_ok=True)
    d, names = {}, []
    sources = {name: get_robot_urdf(name) for name in ROBOT_DATABASE}
    fx = os.path.join(REF, "tests", "urdf_fixtures")
    for f in ("simple_arm", "prismatic_joint", "branched", "continuous_joints", "mimic_joints", "multi_root", "primitives", "transmissions"):
        sources["fixture_" + f] = os.path.join(fx, f + ".urdf")
    for name, path in sources.items():
        try:
            proc = URDFToSerialManipulator(path, load_meshes=False)
        except Exception as exc:  # the reference itself rejects the file: record that
            d[f"{name}__error"] = np.array(type(exc).__name__ + ": " + str(exc)[:160])
            write_skeleton(path, os.path.join(out_dir, f"{name}.urdf"), name)
            names.append(name)
            print(f"urdf suite: {name}: reference raises {type(exc).__name__}", flush=True)
            continue
        sm, dyn = proc.serial_manipulator, proc.dynamics
        n = sm.S_list.shape[1]
        ee = proc.robot.end_effector_link.name
        d[f"{name}__S"] = np.asarray(sm.S_list, dtype=np.float64); d[f"{name}__B"] = np.asarray(sm.B_list, dtype=np.float64)
        d[f"{name}__M"] = np.asarray(sm.M_list, dtype=np.float64); d[f"{name}__G"] = np.asarray(dyn.Glist, dtype=np.float64)
        d[f"{name}__Mcom"] = np.asarray(dyn.Mlist_per_link, dtype=np.float64)
        d[f"{name}__limits"] = np.array([[np.nan if lo is None else lo, np.nan if hi is None else hi] for lo, hi in proc.robot_data["joint_limits"]], dtype=np.float64)
        d[f"{name}__ee"] = np.array(ee)
        d[f"{name}__joint_names"] = np.array([j.name for j in proc.robot.actuated_joints])
        # every other leaf link as an explicit tip (branching trees: grippers, the branched fixture)
        robot = URDF.load(path, backend="builtin", load_meshes=False)
        leaves = [l for l in getattr(robot, "_end_link_names", []) if l != ee]
        for k, leaf in enumerate(sorted(leaves)[:3]):
            prm = robot.extract_screw_axes(tip_link=leaf)
            d[f"{name}__tip{k}_name"] = np.array(leaf)
            d[f"{name}__tip{k}_M"] = np.asarray(prm["M"], dtype=np.float64)
            d[f"{name}__tip{k}_B"] = np.asarray(prm["B_list"], dtype=np.float64)
        # a few reference dynamics values, so the suite also pins tables -> physics for robots outside the benchmark four
        rng = np.random.default_rng(SEED + 1000 + len(names))
        lim = finite_limits(sm, n)
        th = rng.uniform(lim[:, 0], lim[:, 1]); dth = rng.uniform(-1, 1, n); ddth = rng.uniform(-1, 1, n)
        d[f"{name}__theta"], d[f"{name}__dtheta"], d[f"{name}__ddtheta"] = th, dth, ddth
        clear_caches(dyn)
        d[f"{name}__tau"] = np.asarray(dyn.inverse_dynamics(th, dth, ddth, G_VEC, FTIP_REF))
        d[f"{name}__T"] = np.asarray(sm.forward_kinematics(th))
        if n > 8:  # the robots only the run-time-n kernels serve: pin every operation of the path, not only tau and T
            d[f"{name}__mass"] = np.asarray(dyn.mass_matrix(th))
            d[f"{name}__J"] = np.asarray(sm.jacobian(th))
            d[f"{name}__qdd"] = np.asarray(dyn.forward_dynamics(th, dth, d[f"{name}__tau"], G_VEC, FTIP_REF))
            d[f"{name}__c"] = np.asarray(dyn.velocity_quadratic_forces(th, dth))
            d[f"{name}__g"] = np.asarray(dyn.gravity_forces(th, G_VEC))
        write_skeleton(path, os.path.join(out_dir, f"{name}.urdf"), name)
        names.append(name)
        print(f"urdf suite: {name}: n={n} ee={ee} leaves={len(leaves) + 1}", flush=True)
    d["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "urdf_suite.npz"), **d)


def time_reference():
    """Cold-cache single-thread timings of the reference's inverse_dynamics."""
    res = {"host": "build container", "cores_visible": os.cpu_count(), "threads_used": 1,
           "numpy": np.__version__, "note": "caches cleared before every call; float64 inputs"}
    for robot in ROBOTS:
        proc, sm, dyn = build(robot)
        n = sm.S_list.shape[1]
        lims = finite_limits(sm, n)
        rng = np.random.default_rng(SEED + 300)
        ts = []
        for _ in range(12):
            th = rng.uniform(lims[:, 0], lims[:, 1])
            dth, ddth = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
            clear_caches(dyn)
            t0 = time.perf_counter()
            dyn.inverse_dynamics(th, dth, ddth, G_VEC, np.zeros(6))
            ts.append(time.perf_counter() - t0)
        med = float(np.median(ts))
        res[robot] = {"dof": n, "inverse_dynamics_ms_per_point": med * 1e3,
                      "joint_timesteps_per_s_per_core": n / med}
    with open(os.path.join(HERE, "reference_cpu_timings.json"), "w") as f:
        json.dump(res, f, indent=1)



def rollout_workload(rng, sm, dyn, B, N, n):
    """A roll-out that stays finite for a full second: torques that hold the start configuration against gravity plus a
    small disturbance, small per-step wrenches.  (SURVEY §8d's c5 inputs - a free-falling arm with a 3 N / 0.75 N.m tip
    wrench on a 8e-5 kg.m^2 wrist - overflow to inf within ~20 steps of dt = 0.01 in the reference itself.)"""
    th0 = rng.uniform(-0.5, 0.5, (B, n))
    dth0 = rng.uniform(-0.2, 0.2, (B, n))
    hold = np.array([np.asarray(dyn.inverse_dynamics(th0[b], np.zeros(n), np.zeros(n), G_VEC, np.zeros(6))) for b in range(B)])
    taumat = hold[:, None, :] + rng.uniform(-1, 1, (B, N, n)) * 0.001
    Ftipmat = rng.uniform(-1, 1, (B, N, 6)) * 0.02
    return th0, dth0, taumat, Ftipmat


def dump_rollout100():
    """forward_dynamics_trajectory at config c5's own horizon: xarm6, N = 100, dt = 0.01, intRes = 1, per-step wrench,
    float64 inputs, 3 trajectories; plus one at dt = 0.001 / intRes = 2."""
    from ManipulaPy.planning import OptimizedTrajectoryPlanning

    proc, sm, dyn = build("xarm6")
    n = sm.S_list.shape[1]
    lims = finite_limits(sm, n)
    planner = OptimizedTrajectoryPlanning(sm, get_robot_urdf("xarm6"), dyn, lims.tolist(), use_cuda=False)
    rng = np.random.default_rng(SEED + 700)
    B, N = 3, 100
    th0, dth0, taumat, Ftipmat = rollout_workload(rng, sm, dyn, B, N, n)
    d = {"joint_limits": lims, "theta0": th0, "dtheta0": dth0, "taumat": taumat, "Ftipmat": Ftipmat, "g": G_VEC,
         "dt": np.float64(0.01), "intRes": np.int64(1)}
    out = {k: [] for k in ("positions", "velocities", "accelerations")}
    for b in range(B):
        clear_caches(dyn)
        r = planner.forward_dynamics_trajectory(th0[b], dth0[b], taumat[b], G_VEC, Ftipmat[b], 0.01, 1)
        for k in out:
            out[k].append(np.asarray(r[k]))
        print("rollout100", b, float(np.abs(out["velocities"][-1]).max()), flush=True)
    for k in out:
        d[k] = np.stack(out[k])
    clear_caches(dyn)
    r = planner.forward_dynamics_trajectory(th0[0], dth0[0], taumat[0, :40], G_VEC, Ftipmat[0, :40], 0.001, 2)
    for k in out:
        d["fine_" + k] = np.asarray(r[k])
    np.savez(os.path.join(HERE, "fd_rollout100_xarm6.npz"), **d)


def dump_nonfinite():
    """What the reference returns for rows that contain NaN / inf (planning/trajectory_dynamics.py:345-358 catches
    exceptions only - NumPy raises none here, so such rows come back non-finite, not zero)."""
    from ManipulaPy.planning import OptimizedTrajectoryPlanning

    proc, sm, dyn = build("ur5")
    n = sm.S_list.shape[1]
    lims = finite_limits(sm, n)
    planner = OptimizedTrajectoryPlanning(sm, get_robot_urdf("ur5"), dyn, lims.tolist(), use_cuda=False)
    rng = np.random.default_rng(SEED + 800)
    R = 12
    q = rng.uniform(-1, 1, (R, n)); qd = rng.uniform(-1, 1, (R, n)); qdd = rng.uniform(-1, 1, (R, n))
    q[1, 0] = np.nan; q[2, n - 1] = np.inf; qd[3, 2] = np.nan; qd[4, 0] = -np.inf; qdd[5, n - 1] = np.nan; qdd[6, 3] = np.inf
    q[7, 3] = np.nan; qd[7, 1] = np.inf
    clear_caches(dyn)
    tau = np.asarray(planner.inverse_dynamics_trajectory(q, qd, qdd, G_VEC, FTIP_REF))
    d = {"id_q": q, "id_qd": qd, "id_qdd": qdd, "id_ftip": FTIP_REF, "id_tau": tau, "joint_limits": lims}
    print("nonfinite ID rows:", (~np.isfinite(tau)).all(axis=1).astype(int), (~np.isfinite(tau)).any(axis=1).astype(int))
    # roll-out whose torque row 5 holds a NaN (xarm6): rows 0..4 finite, rows 5.. non-finite
    proc, sm, dyn = build("xarm6")
    n = sm.S_list.shape[1]
    lims6 = finite_limits(sm, n)
    planner = OptimizedTrajectoryPlanning(sm, get_robot_urdf("xarm6"), dyn, lims6.tolist(), use_cuda=False)
    th0, dth0, taumat, Ftipmat = rollout_workload(rng, sm, dyn, 1, 10, n)
    taumat[0, 5, 2] = np.nan
    clear_caches(dyn)
    r = planner.forward_dynamics_trajectory(th0[0], dth0[0], taumat[0], G_VEC, Ftipmat[0], 0.01, 1)
    d.update(fd_joint_limits=lims6, fd_theta0=th0[0], fd_dtheta0=dth0[0], fd_taumat=taumat[0], fd_Ftipmat=Ftipmat[0],
             fd_positions=np.asarray(r["positions"]), fd_velocities=np.asarray(r["velocities"]),
             fd_accelerations=np.asarray(r["accelerations"]))
    print("nonfinite FD rows (pos / vel / acc):", *[np.isfinite(np.asarray(r[k])).all(axis=1).astype(int) for k in ("positions", "velocities", "accelerations")])
    np.savez(os.path.join(HERE, "nonfinite.npz"), **d)


def dump_field():
    """potential_field_cpu_fallback (cuda_kernels/field_kernels.py:113-161): 1200 points x 37 obstacles, including points
    that coincide with an obstacle (zero distance: skipped), points on / outside the influence sphere, a point at the
    goal, and the hand-checked case of tests/test_cuda_kernels_cpu.py:104-118."""
    from ManipulaPy.cuda_kernels.field_kernels import potential_field_cpu_fallback

    rng = np.random.default_rng(SEED + 900)
    P, K = 1200, 37
    obstacles = rng.uniform(-1.0, 1.0, (K, 3)).astype(np.float32)
    goal = np.array([0.4, -0.3, 0.6], dtype=np.float32)
    pos = rng.uniform(-1.2, 1.2, (P, 3)).astype(np.float32)
    pos[:K] = obstacles                                   # zero distance to one obstacle each
    pos[K] = goal                                         # at the goal
    infl = 0.35
    for i in range(20):                                   # exactly on / just inside / just outside the influence sphere
        u = rng.normal(size=3); u /= np.linalg.norm(u)
        pos[K + 1 + i] = (obstacles[i].astype(np.float64) + u * infl * (1.0, 1.0 - 1e-6, 1.0 + 1e-6, 0.999)[i % 4]).astype(np.float32)
    for i in range(20):                                   # very close to an obstacle (large 1/d^3 factors)
        u = rng.normal(size=3); u /= np.linalg.norm(u)
        pos[K + 21 + i] = (obstacles[i].astype(np.float64) + u * (1e-3, 1e-2, 5e-2, 1e-4)[i % 4]).astype(np.float32)
    d = {"positions": pos, "goal": goal, "obstacles": obstacles}
    for tag, dist in (("d035", infl), ("d100", 1.0), ("d000", 0.0)):
        pot, grad = potential_field_cpu_fallback(pos, goal, obstacles, dist)
        d[f"{tag}_influence"] = np.float64(dist)
        d[f"{tag}_potential"], d[f"{tag}_gradient"] = pot, grad
    pot, grad = potential_field_cpu_fallback(pos, goal, np.zeros((0, 3), dtype=np.float32), 0.5)   # no obstacles
    d["noobs_potential"], d["noobs_gradient"] = pot, grad
    hp = np.array([[0.0, 0.0, 0.0], [2.0, 0.0, 0.0]], dtype=np.float32)
    pot, grad = potential_field_cpu_fallback(hp, np.array([1.0, 0.0, 0.0], dtype=np.float32), np.array([[0.5, 0.0, 0.0]], dtype=np.float32), 1.0)
    d["hand_positions"], d["hand_potential"], d["hand_gradient"] = hp, pot, grad
    np.savez(os.path.join(HERE, "potential_field.npz"), **d)


def dump_legacy():
    """legacy_dynamics.npz: ManipulatorDynamics WITHOUT Mlist_per_link (the reference's legacy approximation,
    dynamics/mass_matrix.py:101-132, forces.py:136-154) and truncated / body-frame kinematics (fk.py:59-80, jacobian.py:62-91)
    on (A) the hand-built 6-DOF model of the reference's tests/test_public_api_freeze.py:104-158 - whose B_list is NOT
    Ad(M^-1) S_list - and (B) URDF(ur5).to_manipulator_dynamics() (urdf/core.py:795-817)."""
    from math import pi

    from ManipulaPy.dynamics import ManipulatorDynamics
    from ManipulaPy.path_planning import OptimizedTrajectoryPlanning

    out = {}
    S = np.array([[0, 0, 1, 0, 0, 0], [0, -1, 0, -0.089, 0, 0], [0, -1, 0, -0.089, 0, 0.425], [0, -1, 0, -0.089, 0, 0.817],
                  [1, 0, 0, 0, 0.109, 0], [0, -1, 0, -0.089, 0, 0.817]], dtype=float).T
    M = np.array([[1, 0, 0, 0.817], [0, 1, 0, 0], [0, 0, 1, 0.191], [0, 0, 0, 1]], dtype=float)
    G = np.stack([np.eye(6) * (1.0 + 0.1 * i) for i in range(6)])
    dynA = ManipulatorDynamics(M_list=M, omega_list=None, r_list=None, b_list=None, S_list=S, B_list=S.copy(), Glist=G)
    proc, sm, _ = build("ur5")
    dynB = proc.robot.to_manipulator_dynamics()
    rng = np.random.default_rng(SEED + 77)
    for tag, dyn in (("A", dynA), ("B", dynB)):
        n = dyn.S_list.shape[1]
        K = 5
        th = rng.uniform(-1.5, 1.5, (K, n)); dth = rng.uniform(-1, 1, (K, n)); ddth = rng.uniform(-2, 2, (K, n))
        F = rng.uniform(-2, 2, (K, 6))
        out[f"{tag}_S"], out[f"{tag}_B"], out[f"{tag}_M"], out[f"{tag}_G"] = (np.asarray(dyn.S_list), np.asarray(dyn.B_list),
                                                                               np.asarray(dyn.M_list), np.asarray(dyn.Glist))
        out[f"{tag}_theta"], out[f"{tag}_dtheta"], out[f"{tag}_ddtheta"], out[f"{tag}_ftip"] = th, dth, ddth, F
        mm, cc, gg, idd, fdd = [], [], [], [], []
        for k in range(K):
            clear_caches(dyn)
            mm.append(dyn.mass_matrix(th[k])); cc.append(dyn.velocity_quadratic_forces(th[k], dth[k]))
            gg.append(dyn.gravity_forces(th[k], G_VEC)); idd.append(dyn.inverse_dynamics(th[k], dth[k], ddth[k], G_VEC, F[k]))
            fdd.append(dyn.forward_dynamics(th[k], dth[k], idd[-1], G_VEC, F[k]))
        out[f"{tag}_mass"], out[f"{tag}_c"], out[f"{tag}_g"], out[f"{tag}_id"], out[f"{tag}_fd"] = map(np.array, (mm, cc, gg, idd, fdd))
        # truncated chains and both frames
        for k in range(0, n + 1):
            out[f"{tag}_fk_space_{k}"] = dyn.forward_kinematics(th[0][:k], "space")
            out[f"{tag}_fk_body_{k}"] = dyn.forward_kinematics(th[0][:k], "body")
            out[f"{tag}_jac_space_{k}"] = dyn.jacobian(th[0][:k], "space")
            if k >= 1:
                out[f"{tag}_jac_body_{k}"] = dyn.jacobian(th[0][:k], "body")
        # planner level: the reference's CPU loops over the legacy object
        lim = [(-pi, pi)] * n
        pl = OptimizedTrajectoryPlanning(dyn, "nonexistent.urdf", dyn, lim, use_cuda=False)
        traj = pl.joint_trajectory(th[0], th[1], 1.0, 8, 5)
        out[f"{tag}_traj_pos"], out[f"{tag}_traj_vel"], out[f"{tag}_traj_acc"] = traj["positions"], traj["velocities"], traj["accelerations"]
        clear_caches(dyn)
        out[f"{tag}_traj_tau"] = pl.inverse_dynamics_trajectory(traj["positions"], traj["velocities"], traj["accelerations"])
        taum = np.array(idd)[:, :] * 0.5
        clear_caches(dyn)
        r = pl.forward_dynamics_trajectory(th[2], dth[2] * 0.1, np.tile(taum[2], (6, 1)), G_VEC, np.tile(F[2] * 0.1, (6, 1)), 0.01, 2)
        out[f"{tag}_roll_tau"], out[f"{tag}_roll_F"] = np.tile(taum[2], (6, 1)), np.tile(F[2] * 0.1, (6, 1))
        out[f"{tag}_roll_pos"], out[f"{tag}_roll_vel"], out[f"{tag}_roll_acc"] = r["positions"], r["velocities"], r["accelerations"]
    np.savez(os.path.join(HERE, "legacy_dynamics.npz"), **out)


def main():
    assert os.environ.get("PYTHONHASHSEED") == "0"
    if "urdf" in sys.argv[1:]:  # only (re)generate the URDF skeletons
        dump_urdfs()
        print("urdf skeletons dumped")
        return
    if "utils" in sys.argv[1:]:  # only (re)generate the utils dump
        dump_utils()
        print("utils dumped")
        return
    if "control" in sys.argv[1:]:  # only (re)generate the controller dump
        dump_control()
        print("control dumped")
        return
    if "ik" in sys.argv[1:]:  # only (re)generate the inverse-kinematics dump
        dump_ik()
        print("ik dumped")
        return
    for name, fn in (("plan", dump_plan), ("urdf_api", dump_urdf_api), ("gain_sweep", dump_gain_sweep), ("rollout100", dump_rollout100), ("nonfinite", dump_nonfinite), ("field", dump_field), ("urdf_suite", dump_urdf_suite),
                     ("legacy", dump_legacy)):
        if name in sys.argv[1:]:
            fn()
            print(name, "dumped")
            return
    if "cartesian" in sys.argv[1:]:  # only (re)generate the Cartesian-trajectory dump
        dump_cartesian()
        print("cartesian dumped")
        return
    for i, robot in enumerate(ROBOTS):
        proc, sm, dyn = build(robot)
        n = dump_model(robot, proc, sm, dyn)
        dump_dynamics(robot, sm, dyn, n, i)
        print(f"{robot}: n={n} dumped", flush=True)
    dump_trajectories()
    dump_cartesian()
    dump_ik()
    dump_control()
    dump_utils()
    dump_urdfs()
    dump_rollout100()
    dump_nonfinite()
    dump_field()
    dump_urdf_suite()
    dump_legacy()
    dump_gain_sweep()
    dump_urdf_api()
    dump_plan()
    print("trajectories dumped", flush=True)
    time_reference()
    print("timings dumped")


if __name__ == "__main__":
    main()
