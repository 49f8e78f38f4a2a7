#!/usr/bin/env python3
"""Dev probe: one specialised IK launch (UR5, 262144 problems) for rocprofv3 --pmc runs."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import manipulapy_amd as mp
from manipulapy_amd import _hip
ctx = _hip.HipContext(0)
rng = np.random.default_rng(0)
sm, dyn, lim = mp.load_robot("ur5")
model = _hip.HipModel(dyn.S_list, dyn.Mlist_per_link, dyn.Glist, sm.M_list, lim)
if os.environ.get("SPEC", "1") == "1":
    ctx.specialize(model)
n, B = model.n, 1 << 18
lim = np.asarray(lim, dtype=np.float64)
q_true = rng.uniform(0.6 * lim[:, 0], 0.6 * lim[:, 1], (B, n))
dq, dT = ctx.to_device(q_true), ctx.alloc(B * 128)
ctx.fk_jac_id(model, dq, None, None, B, dT, None, None)
q0 = np.clip(q_true + rng.uniform(-0.3, 0.3, (B, n)), lim[:, 0], lim[:, 1])
d0, dth = ctx.to_device(q0), ctx.alloc(B * n * 8)
dok, dit, drs = ctx.alloc(B * 4), ctx.alloc(B * 4), ctx.alloc(B * 4)
for _ in range(3):
    ctx.inverse_kinematics(model, dT, d0, B, dth, dok, dit, drs, joint_limits=lim, max_iterations=200)
ctx.synchronize()
it = dit.download((B,), np.int32)
print("total iterations", int(it.sum()), "mean", float(it.mean()))
