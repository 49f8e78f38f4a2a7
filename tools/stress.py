#!/usr/bin/env python3
"""Dev stress run (not part of the suite): random sizes around block / wave / tile boundaries for every device entry point,
generic against specialised kernels, fp32 against fp64.  Prints one line per robot; exits non-zero on a mismatch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import manipulapy_amd as mp
from manipulapy_amd import _hip

ctx = _hip.HipContext(0)
rng = np.random.default_rng(int(os.environ.get("SEED", "0")))
bad = 0
for robot in ("ur5", "iiwa14", "panda", "xarm6", "panda7"):
    t = mp.robot_tables(robot)
    gen = _hip.HipModel(t["S_list"], t["Mlist_per_link"], t["Glist"], t["M_ee"], t["joint_limits"])
    spec = _hip.HipModel(t["S_list"], t["Mlist_per_link"], t["Glist"], t["M_ee"], t["joint_limits"])
    ctx.specialize(spec)
    n = gen.n
    lim = np.asarray(t["joint_limits"], dtype=np.float64)
    checks = 0
    for trial in range(12):
        rows = int(rng.choice([1, 2, 3, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 513, int(rng.integers(1, 5000))]))
        q = rng.uniform(lim[:, 0], lim[:, 1], (rows, n)); qd = rng.uniform(-2, 2, (rows, n)); qdd = rng.uniform(-4, 4, (rows, n))
        F = None if trial % 2 else rng.uniform(-3, 3, 6)
        a64 = ctx.id_trajectory_host(gen, q, qd, qdd, None, F, dtype=np.float64)
        b64 = ctx.id_trajectory_host(spec, q, qd, qdd, None, F, dtype=np.float64)
        a32 = ctx.id_trajectory_host(gen, q, qd, qdd, None, F, dtype=np.float32)
        b32 = ctx.id_trajectory_host(spec, q, qd, qdd, None, F, dtype=np.float32)
        sc = max(1.0, float(np.abs(a64).max()))
        ok = (np.abs(a64 - b64).max() <= 1e-9 * sc and np.abs(a32 - a64).max() <= 3e-4 * sc and np.abs(b32 - a64).max() <= 3e-4 * sc)
        Tg, Jg, tg = ctx.fk_jac_id_host(gen, q, qd, qdd, None, F); Ts, Js, ts = ctx.fk_jac_id_host(spec, q, qd, qdd, None, F)
        ok &= np.abs(Tg - Ts).max() <= 1e-10 and np.abs(Jg - Js).max() <= 1e-10 and np.abs(tg - a64).max() <= 1e-9 * sc
        Mg, Ms = ctx.mass_matrix_host(gen, q), ctx.mass_matrix_host(spec, q)
        ok &= np.abs(Mg - Ms).max() <= 1e-10 * max(1.0, float(np.abs(Mg).max())) and np.abs(Mg - np.swapaxes(Mg, 1, 2)).max() <= 1e-12 * max(1.0, float(np.abs(Mg).max()))
        fg, fs = ctx.forward_dynamics_host(gen, q, qd, qdd, None, F), ctx.forward_dynamics_host(spec, q, qd, qdd, None, F)
        ok &= np.abs(fg - fs).max() <= 1e-7 * max(1.0, float(np.abs(fg).max()))
        # FD(ID) round trip in float64
        rt = ctx.forward_dynamics_host(gen, q, qd, a64, None, F)
        ok &= np.abs(rt - qdd).max() <= 1e-6 * max(1.0, float(np.abs(qdd).max()))
        B, Nt = int(rng.choice([1, 2, 63, 64, 65, 130])), int(rng.choice([1, 2, 3, 4, 5, 8, 9, 17]))
        th0 = rng.uniform(0.3 * lim[:, 0], 0.3 * lim[:, 1], (B, n)); d0 = rng.uniform(-0.2, 0.2, (B, n)); tm = rng.uniform(-1, 1, (B, Nt, n))
        Fm = None if trial % 3 else rng.uniform(-1, 1, (B, Nt, 6))
        for dt in (np.float32, np.float64):
            ra = ctx.fd_trajectory_host(gen, th0, d0, tm, None, Fm, 0.004, 1 + trial % 2, dtype=dt)
            rb = ctx.fd_trajectory_host(spec, th0, d0, tm, None, Fm, 0.004, 1 + trial % 2, dtype=dt)
            for x, y in zip(ra, rb):
                ok &= bool(np.isfinite(x).all()) and np.abs(x - y).max() <= (2e-3 if dt == np.float32 else 1e-6) * max(1.0, float(np.abs(x).max()))
        st, en = rng.uniform(lim[:, 0], lim[:, 1], (2, B, n)).astype(np.float32)
        fa, fb = ctx.traj_id_fused_host(gen, st, en, 1.5, max(Nt, 2), 5, None, F), ctx.traj_id_fused_host(spec, st, en, 1.5, max(Nt, 2), 5, None, F)
        p, v, a = ctx.batch_trajectory_host(gen, st, en, 1.5, max(Nt, 2), 5)
        two = ctx.id_trajectory_host(gen, p.reshape(-1, n), v.reshape(-1, n), a.reshape(-1, n), None, F).reshape(fa.shape)
        sc2 = max(1.0, float(np.abs(two).max()))
        ok &= np.abs(fa - fb).max() <= 3e-5 * sc2 and np.abs(fa - two).max() <= 3e-5 * sc2
        checks += 1
        if not ok:
            bad += 1
            print("MISMATCH", robot, trial, rows, B, Nt)
    print(robot, "n =", n, "trials", checks, "bad so far", bad, flush=True)
ctx.destroy()
sys.exit(1 if bad else 0)
