#!/usr/bin/env python3
"""The reference's planner workflow on the HIP backend: URDF -> tables -> trajectory -> torques -> roll-out.

    python examples/quickstart.py            (needs an MI355X; the GPU path is an explicit opt-in, as in the reference)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import manipulapy_amd as mp  # noqa: E402

proc = mp.URDFToSerialManipulator(mp.robot_urdf("ur5"))
robot, dynamics = proc.serial_manipulator, proc.dynamics
mp.set_backend("hip")
planner = mp.OptimizedTrajectoryPlanning(robot, proc.urdf_name, dynamics, proc.robot_data["joint_limits"])

# one trajectory, N = 1000 (reference: joint_trajectory + inverse_dynamics_trajectory)
traj = planner.joint_trajectory(np.zeros(6), np.array([0.8, -0.6, 0.9, -0.4, 0.5, 0.3]), Tf=2.0, N=1000, method=5)
tau = planner.inverse_dynamics_trajectory(traj["positions"], traj["velocities"], traj["accelerations"])
print("single trajectory: tau", tau.shape, "peak |tau| per joint", np.abs(tau).max(axis=0).round(2))

# 4096 trajectories at once, generation fused into the dynamics (only the torques cross PCIe)
rng = np.random.default_rng(0)
lim = np.asarray(proc.robot_data["joint_limits"], dtype=np.float64)
start, end = rng.uniform(lim[:, 0], lim[:, 1], (2, 4096, 6))
tau_b = planner.batch_inverse_dynamics_trajectory(start, end, Tf=2.0, N=1000, method=5)
print("batch:", tau_b.shape, tau_b.dtype)

# forward-dynamics roll-outs of 256 torque histories
roll = planner.batch_forward_dynamics_trajectory(start[:256] * 0.2, np.zeros((256, 6)), tau_b[:256, :200] * 0.01, [0, 0, -9.81], None, 0.005, 1)
print("roll-outs:", roll["positions"].shape)

# kinematics: FK / Jacobian of a batch, inverse kinematics back to the same poses
q = rng.uniform(0.5 * lim[:, 0], 0.5 * lim[:, 1], (1000, 6))
T = robot.forward_kinematics(q)
theta, ok, iters = robot.batch_inverse_kinematics(T, q + rng.uniform(-0.2, 0.2, q.shape), max_iterations=200)
print(f"IK: {ok.mean() * 100:.1f} % of 1000 targets solved, {iters.mean():.1f} iterations on average")
print(planner.get_performance_stats())
