#!/usr/bin/env python3
"""Dev A/B: the batched inverse-kinematics kernel, generic and specialised, one measurement per process (env knobs decide the variant)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import manipulapy_amd as mp
from manipulapy_amd import _hip

def main():
    tag = sys.argv[1]
    ctx = _hip.HipContext(0)
    rng = np.random.default_rng(1)
    for robot in ("ur5", "iiwa14"):
        sm, dyn, lim = mp.load_robot(robot)
        model = _hip.HipModel(dyn.S_list, dyn.Mlist_per_link, dyn.Glist, sm.M_list, lim)
        n, B = model.n, 1 << 18
        lim = np.asarray(lim, dtype=np.float64)
        q_true = rng.uniform(0.6 * lim[:, 0], 0.6 * lim[:, 1], (B, n))
        dq, dT = ctx.to_device(q_true), ctx.alloc(B * 128)
        ctx.fk_jac_id(model, dq, None, None, B, dT, None, None)
        q0 = np.clip(q_true + rng.uniform(-0.3, 0.3, (B, n)), lim[:, 0], lim[:, 1])
        d0, dth = ctx.to_device(q0), ctx.alloc(B * n * 8)
        dok, dit, drs = ctx.alloc(B * 4), ctx.alloc(B * 4), ctx.alloc(B * 4)
        res = {}
        for kind in ("generic", "specialised"):
            if kind == "specialised":
                ctx.specialize(model)
            run = lambda: ctx.inverse_kinematics(model, dT, d0, B, dth, dok, dit, drs, joint_limits=lim, max_iterations=200)
            run(); ctx.synchronize()
            best = 1e9
            for _ in range(4):
                t0 = time.perf_counter(); run(); ctx.synchronize(); best = min(best, time.perf_counter() - t0)
            res[kind] = round(best * 1e3, 3)
        print(json.dumps({"variant": tag, "robot": robot, **res}), flush=True)

main()
