"""The reference's frozen public-API contract as a pin for the mirrored classes.

tests/golden/api_contract_golden.json is the reference's own tests/data/api_contract_golden.json (data, byte for byte): for
every frozen symbol the rendered signature and the type / shape / dtype of what a fixed call returns
(reference tests/test_public_api_freeze.py:170-352 builds it).  Here the SAME calls are made on manipulapy_amd's classes
and, for every symbol this package mirrors, the test asserts

  * parameter names, their order and their defaults (parsed out of the frozen signature string), and
  * the return container type, array shapes and dtypes, recursively,

under the default NumPy backend (CPU suite) and under the "hip" backend (-m gpu).  The fixture robot is the hand-built
6-DOF model of the reference's test - a ManipulatorDynamics WITHOUT Mlist_per_link, i.e. the legacy approximation.
All 22 frozen symbols are mirrored; NOT_MIRRORED (empty) is checked to be exactly the rest of the contract.
"""
import inspect
import json
import os
import warnings
from math import pi

import numpy as np
import pytest

import manipulapy_amd as mp

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "api_contract_golden.json")))

# every frozen symbol is mirrored (the TRAC-IK three since round 3: manipulapy_amd/trac_ik.py)
NOT_MIRRORED = set()


def split_top(text, sep=","):
    """Split at `sep` outside any bracket."""
    out, depth, cur = [], 0, ""
    for ch in text:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [p.strip() for p in out]


def frozen_parameters(signature):
    """[(name, default source or None)] from a rendered signature '(a: T, b: U = 1, *, c=2) -> R'."""
    depth, end = 0, None
    for i, ch in enumerate(signature):
        depth += ch == "("
        depth -= ch == ")"
        if depth == 0 and ch == ")":
            end = i
            break
    params = []
    for part in split_top(signature[1:end]):
        if part in ("*", "/"):
            continue
        name = part.split(":")[0].split("=")[0].strip()
        default = None
        pieces = split_top(part, "=")
        if len(pieces) == 2:
            default = pieces[1].strip()
        params.append((name, default))
    return params


def live_parameters(fn):
    out = []
    for p in inspect.signature(fn).parameters.values():
        out.append((("*" if p.kind is p.VAR_POSITIONAL else "**" if p.kind is p.VAR_KEYWORD else "") + p.name,
                    None if p.default is p.empty else repr(p.default)))
    return out


def describe(value):
    """type / shape / dtype, recursively - the reference's describe() (tests/test_public_api_freeze.py:69-88) restated."""
    tp = type(value)
    entry = {"type": tp.__qualname__ if tp.__module__ == "builtins" else f"{tp.__module__}.{tp.__qualname__}"}
    if isinstance(value, np.ndarray):
        entry["shape"] = list(value.shape)
        entry["dtype"] = str(value.dtype)
    elif isinstance(value, np.generic):
        entry["dtype"] = str(value.dtype)
    elif isinstance(value, (tuple, list)):
        entry["elements"] = [describe(v) for v in value]
    elif isinstance(value, dict):
        entry["items"] = {str(k): describe(v) for k, v in sorted(value.items())}
    return entry


def build_calls():
    """The reference's fixture and calls (tests/test_public_api_freeze.py:104-352), on this package's classes."""
    S = np.array([[0, 0, 1, 0, 0, 0], [0, -1, 0, -0.089, 0, 0], [0, -1, 0, -0.089, 0, 0.425], [0, -1, 0, -0.089, 0, 0.817],
                  [1, 0, 0, 0, 0.109, 0], [0, -1, 0, -0.089, 0, 0.817]], dtype=float).T
    M = np.array([[1, 0, 0, 0.817], [0, 1, 0, 0], [0, 0, 1, 0.191], [0, 0, 0, 1]], dtype=float)
    robot = mp.SerialManipulator(M_list=M, omega_list=S[:3, :], S_list=S, B_list=S.copy(), joint_limits=[(-pi, pi)] * 6)
    G = np.stack([np.eye(6) * (1.0 + 0.1 * i) for i in range(6)])
    dyn = mp.ManipulatorDynamics(M_list=M, omega_list=None, r_list=None, b_list=None, S_list=S, B_list=S.copy(), Glist=G)
    sing = mp.Singularity(robot)
    theta = np.array([0.1, 0.2, -0.3, 0.4, -0.5, 0.6])
    theta2 = theta + 0.2
    dtheta, ddtheta = np.full(6, 0.05), np.full(6, 0.01)
    g_vec, ftip = np.array([0.0, 0.0, -9.81]), np.zeros(6)
    T_desired = robot.forward_kinematics(theta)
    planner = mp.OptimizedTrajectoryPlanning(robot, "nonexistent.urdf", dyn, [(-pi, pi)] * 6, use_cuda=False)
    traj = planner.joint_trajectory(theta, theta2, 1.0, 8, 5)
    pos, vel, acc = traj["positions"], traj["velocities"], traj["accelerations"]
    taumat, ftipmat = np.zeros((pos.shape[0], 6)), np.zeros((pos.shape[0], 6))
    trac_solver = mp.TracIKSolver(fk_func=lambda th: robot.forward_kinematics(th, frame="space"),
                                  jacobian_func=lambda th: robot.jacobian(th, frame="space"), joint_limits=robot.joint_limits, n_joints=6)
    return {
        "SerialManipulator.trac_ik": (robot.trac_ik, lambda: robot.trac_ik(T_desired, theta0=theta.copy(), timeout=0.05)),
        "trac_ik.TracIKSolver.solve": (trac_solver.solve, lambda: trac_solver.solve(T_desired, theta0=theta.copy(), timeout=0.05)),
        "trac_ik.trac_ik_solve": (mp.trac_ik_solve, lambda: mp.trac_ik_solve(robot, T_desired, theta0=theta.copy(), timeout=0.05)),
        "SerialManipulator.forward_kinematics": (robot.forward_kinematics, lambda: robot.forward_kinematics(theta)),
        "SerialManipulator.jacobian": (robot.jacobian, lambda: robot.jacobian(theta)),
        "SerialManipulator.end_effector_velocity": (robot.end_effector_velocity, lambda: robot.end_effector_velocity(theta, dtheta)),
        "SerialManipulator.iterative_inverse_kinematics": (robot.iterative_inverse_kinematics,
                                                           lambda: robot.iterative_inverse_kinematics(T_desired, theta.copy(), max_iterations=50)),
        "SerialManipulator.robust_inverse_kinematics": (robot.robust_inverse_kinematics,
                                                        lambda: robot.robust_inverse_kinematics(T_desired, max_attempts=1, max_iterations=100)),
        "SerialManipulator.smart_inverse_kinematics": (robot.smart_inverse_kinematics,
                                                       lambda: robot.smart_inverse_kinematics(T_desired, max_iterations=100, auto_fallback=False)),
        "ManipulatorDynamics.mass_matrix": (dyn.mass_matrix, lambda: dyn.mass_matrix(theta)),
        "ManipulatorDynamics.velocity_quadratic_forces": (dyn.velocity_quadratic_forces, lambda: dyn.velocity_quadratic_forces(theta, dtheta)),
        "ManipulatorDynamics.gravity_forces": (dyn.gravity_forces, lambda: dyn.gravity_forces(theta, g_vec)),
        "ManipulatorDynamics.inverse_dynamics": (dyn.inverse_dynamics, lambda: dyn.inverse_dynamics(theta, dtheta, ddtheta, g_vec, ftip)),
        "ManipulatorDynamics.forward_dynamics": (dyn.forward_dynamics, lambda: dyn.forward_dynamics(theta, dtheta, ddtheta, g_vec, ftip)),
        "Singularity.singularity_analysis": (sing.singularity_analysis, lambda: sing.singularity_analysis(theta)),
        "Singularity.near_singularity_detection": (sing.near_singularity_detection, lambda: sing.near_singularity_detection(theta)),
        "Singularity.condition_number": (sing.condition_number, lambda: sing.condition_number(theta)),
        "OptimizedTrajectoryPlanning.joint_trajectory": (planner.joint_trajectory, lambda: planner.joint_trajectory(theta, theta2, 1.0, 8, 5)),
        "OptimizedTrajectoryPlanning.cartesian_trajectory": (planner.cartesian_trajectory,
                                                             lambda: planner.cartesian_trajectory(robot.forward_kinematics(theta), robot.forward_kinematics(theta2), 1.0, 8, 5)),
        "OptimizedTrajectoryPlanning.calculate_derivatives": (planner.calculate_derivatives, lambda: planner.calculate_derivatives(pos, 0.1)),
        "OptimizedTrajectoryPlanning.inverse_dynamics_trajectory": (planner.inverse_dynamics_trajectory,
                                                                    lambda: planner.inverse_dynamics_trajectory(pos, vel, acc)),
        "OptimizedTrajectoryPlanning.forward_dynamics_trajectory": (planner.forward_dynamics_trajectory,
                                                                    lambda: planner.forward_dynamics_trajectory(theta, dtheta, taumat, g_vec, ftipmat, 0.1, 1)),
    }


def leaf_diffs(path, golden, live, out):
    if isinstance(golden, dict) and isinstance(live, dict):
        for key in sorted(set(golden) | set(live)):
            if key not in golden or key not in live:
                out.append(f"{path}.{key}: {golden.get(key, '<missing>')!r} != {live.get(key, '<missing>')!r}")
            else:
                leaf_diffs(f"{path}.{key}", golden[key], live[key], out)
    elif isinstance(golden, list) and isinstance(live, list):
        if len(golden) != len(live):
            out.append(f"{path}.length: {len(golden)} != {len(live)}")
        for i, (a, b) in enumerate(zip(golden, live)):
            leaf_diffs(f"{path}[{i}]", a, b, out)
    elif golden != live:
        out.append(f"{path}: {golden!r} != {live!r}")


def check_contract():
    calls = build_calls()
    assert set(GOLDEN) - set(calls) == NOT_MIRRORED, sorted(set(GOLDEN) - set(calls) ^ NOT_MIRRORED)
    problems = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, (fn, call) in sorted(calls.items()):
            want, got = frozen_parameters(GOLDEN[name]["signature"]), live_parameters(fn)
            # a mirrored method may ADD trailing keyword parameters with defaults (batch extensions); the frozen ones must
            # come first, in order, with the frozen defaults
            if [p for p, _ in got[: len(want)]] != [p for p, _ in want]:
                problems.append(f"{name}: parameters {[p for p, _ in got]} != frozen {[p for p, _ in want]}")
            else:
                for (p, d_want), (_, d_got) in zip(want, got):
                    if (d_want is None) != (d_got is None) or (d_want is not None and str(d_want) != str(d_got)):
                        problems.append(f"{name}: default of {p}: {d_got} != frozen {d_want}")
                for p, d in got[len(want):]:
                    if d is None and not p.startswith("*"):
                        problems.append(f"{name}: extra parameter {p} without a default")
            diffs = []
            leaf_diffs("return", GOLDEN[name]["return"], describe(call()), diffs)
            problems += [f"{name}: {d}" for d in diffs]
    assert not problems, "\n".join(problems)


def test_public_api_contract_numpy_backend():
    """Default backend: the CPU launchers (C ABI *_cpu twins) and the host paths."""
    with mp.use_backend("numpy"):
        check_contract()


@pytest.mark.gpu
def test_public_api_contract_hip_backend():
    """"hip" backend: the same calls route to the HIP kernels; types, shapes and dtypes must not move."""
    with mp.use_backend("hip"):
        check_contract()
