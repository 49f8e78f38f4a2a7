#!/usr/bin/env python3
"""Time the secondary operators of the path on device-resident buffers (HIP events on the library's stream):
mass matrix, forward dynamics, trajectory generation, Cartesian path, potential field.  One JSON line per
operator: rows/s and algorithmic GB/s.  bench.py covers the headline configurations (c2..c5)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import manipulapy_amd as mp  # noqa: E402
from manipulapy_amd import _hip  # noqa: E402


def timed(ctx, fn, steps=10, warmup=2):
    for _ in range(warmup):
        fn()
    ctx.synchronize()
    a, b = ctx.event(), ctx.event()
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    ctx.synchronize()
    ms = b.elapsed_ms_since(a) / steps
    a.destroy(); b.destroy()
    return ms


def main():
    rows = int(os.environ.get("ROWS", 1 << 22))
    ctx = _hip.HipContext(0)
    rng = np.random.default_rng(0)
    out = []
    for robot in ("ur5", "iiwa14"):
        sm, dyn, lim = mp.load_robot(robot)
        model = _hip.HipModel(dyn.S_list, dyn.Mlist_per_link, dyn.Glist, sm.M_list, lim)
        n = model.n
        for dt in (np.float64, np.float32):
            w = np.dtype(dt).itemsize
            q = ctx.to_device(rng.uniform(-1, 1, (rows, n)).astype(dt))
            qd = ctx.to_device(rng.uniform(-1, 1, (rows, n)).astype(dt))
            tau = ctx.to_device(rng.uniform(-1, 1, (rows, n)).astype(dt))
            M = ctx.alloc(rows * n * n * w)
            qdd = ctx.alloc(rows * n * w)
            ms = timed(ctx, lambda: ctx.mass_matrix(model, q, rows, M, dtype=dt))
            by = rows * (n + n * n) * w
            out.append(dict(op="mass_matrix", robot=robot, dtype=np.dtype(dt).name, rows=rows, ms=ms, rows_per_s=rows / ms * 1e3,
                            alg_GBps=by / ms / 1e6))
            ms = timed(ctx, lambda: ctx.forward_dynamics(model, q, qd, tau, rows, qdd, dtype=dt))
            by = rows * 4 * n * w
            out.append(dict(op="forward_dynamics", robot=robot, dtype=np.dtype(dt).name, rows=rows, ms=ms,
                            rows_per_s=rows / ms * 1e3, alg_GBps=by / ms / 1e6))
            spec = _hip.HipModel(dyn.S_list, dyn.Mlist_per_link, dyn.Glist, sm.M_list, lim)
            ctx.specialize(spec)
            ms = timed(ctx, lambda: ctx.forward_dynamics(spec, q, qd, tau, rows, qdd, dtype=dt))
            out.append(dict(op="forward_dynamics (specialised)", robot=robot, dtype=np.dtype(dt).name, rows=rows, ms=ms,
                            rows_per_s=rows / ms * 1e3, alg_GBps=by / ms / 1e6))
            spec.destroy()
            for b in (q, qd, tau, M, qdd):
                b.free()
        # trajectory generation: B x N rows, three float32 outputs
        B, N = 4096, 1000
        st = ctx.to_device(rng.uniform(-1, 1, (B, n)).astype(np.float32))
        en = ctx.to_device(rng.uniform(-1, 1, (B, n)).astype(np.float32))
        o = [ctx.alloc(B * N * n * 4) for _ in range(3)]
        ms = timed(ctx, lambda: ctx.batch_trajectory(model, st, en, B, N, 2.0, 5, *o))
        out.append(dict(op="batch_trajectory", robot=robot, dtype="float32", rows=B * N, ms=ms, rows_per_s=B * N / ms * 1e3,
                        alg_GBps=3 * B * N * n * 4 / ms / 1e6))
        for b in (st, en, *o):
            b.free()
        model.destroy()
    # Cartesian straight-line path
    B, N = 4096, 1000
    X = np.tile(np.eye(4), (B, 1, 1)); X[:, :3, 3] = rng.uniform(-1, 1, (B, 3))
    Y = X.copy(); Y[:, :3, 3] += 0.3
    dX, dY = ctx.to_device(X), ctx.to_device(Y)
    o = [ctx.alloc(B * N * 3 * 4) for _ in range(3)] + [ctx.alloc(B * N * 9 * 4)]
    ms = timed(ctx, lambda: ctx.cartesian_trajectory(dX, dY, B, N, 2.0, 5, *o))
    out.append(dict(op="cartesian_trajectory", dtype="float32", rows=B * N, ms=ms, rows_per_s=B * N / ms * 1e3,
                    alg_GBps=B * N * 18 * 4 / ms / 1e6))
    for b in (dX, dY, *o):
        b.free()
    # potential field: P points x O obstacles
    P, O = 1 << 22, 64
    pos = ctx.to_device(rng.uniform(-2, 2, (P, 3)).astype(np.float32))
    obs = ctx.to_device(rng.uniform(-2, 2, (O, 3)).astype(np.float32))
    pot, grad = ctx.alloc(P * 4), ctx.alloc(P * 12)
    ms = timed(ctx, lambda: ctx.potential_field(pos, [0.5, 0.5, 0.5], obs, P, O, 0.8, pot, grad))
    out.append(dict(op="potential_field", dtype="float32", rows=P, obstacles=O, ms=ms, rows_per_s=P / ms * 1e3,
                    alg_GBps=P * 28 / ms / 1e6, pair_per_s=P * O / ms * 1e3))
    # batched inverse kinematics: targets reachable by construction, guesses 0.3 rad away
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
        ms = timed(ctx, lambda: ctx.inverse_kinematics(model, dT, d0, B, dth, dok, dit, drs, joint_limits=lim, max_iterations=200),
                   steps=3, warmup=1)
        ok = dok.download((B,), np.int32); it = dit.download((B,), np.int32)
        out.append(dict(op="inverse_kinematics", robot=robot, dtype="float64", problems=B, ms=ms, problems_per_s=B / ms * 1e3,
                        success_rate=float(ok.mean()), mean_iterations=float(it.mean()), max_iterations=int(it.max()),
                        iterations_per_s=float(it.sum()) / ms * 1e3))
        ctx.specialize(model)
        ms = timed(ctx, lambda: ctx.inverse_kinematics(model, dT, d0, B, dth, dok, dit, drs, joint_limits=lim, max_iterations=200),
                   steps=3, warmup=1)
        ok = dok.download((B,), np.int32); it = dit.download((B,), np.int32)
        out.append(dict(op="inverse_kinematics (specialised)", robot=robot, dtype="float64", problems=B, ms=ms,
                        problems_per_s=B / ms * 1e3, success_rate=float(ok.mean()), mean_iterations=float(it.mean()),
                        iterations_per_s=float(it.sum()) / ms * 1e3))
        for b in (dq, dT, d0, dth, dok, dit, drs):
            b.free()
        model.destroy()
    # 9 / 10-joint arms (run-time-n kernels): inverse kinematics, and the controller's gain sweep (121 closed-loop runs, one launch)
    import time as _time
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
    zj = np.load(os.path.join(gold, "urdf_suite.npz"))
    proc = mp.URDFToSerialManipulator(os.path.join(gold, "urdf_suite", "jaco_7dof.urdf"), tip_link=str(zj["jaco_7dof__ee"]))
    smj = proc.serial_manipulator
    nj, Bj = len(smj.joint_limits), 1 << 16
    thj = zj["jaco_7dof__theta"]
    goal = mp.ik_helpers.clip_to_limits(thj + rng.uniform(-0.2, 0.2, (Bj, nj)), smj.joint_limits)
    with mp.use_backend("hip"):
        Tg = smj.forward_kinematics(goal)
        smj.batch_inverse_kinematics(Tg[:256], np.tile(thj, (256, 1)), max_iterations=200)
        t0 = _time.perf_counter()
        sol, okj, itj = smj.batch_inverse_kinematics(Tg, np.tile(thj, (Bj, 1)), max_iterations=200)
        ms = (_time.perf_counter() - t0) * 1e3
        out.append(dict(op="inverse_kinematics host call (10 joints, k_dyn_ik)", robot="jaco_7dof", dtype="float64", problems=Bj, ms=ms,
                        problems_per_s=Bj / ms * 1e3, success_rate=float(okj.mean()), mean_iterations=float(itj.mean())))
        smu, dynu, limu = mp.load_robot("ur5")
        ctl = mp.ManipulatorController(dynu)
        ctl.find_ultimate_gain_and_period(np.full(6, 0.1), np.full(6, 0.5), 0.01, 10)
        for steps in (200, 1000):
            t0 = _time.perf_counter()
            Ku, Tu, gh, eh = ctl.find_ultimate_gain_and_period(np.full(6, 0.1), np.full(6, 0.5), 0.01, steps)
            ms = (_time.perf_counter() - t0) * 1e3
            out.append(dict(op="find_ultimate_gain_and_period (121 gains, one k_pd_regulation launch)", robot="ur5", dtype="float64", steps=steps,
                            ms=ms, gains_visited=len(gh), Ku=Ku, closed_loop_steps_per_s=121 * steps / ms * 1e3))
    # host-buffer entry points (PCIe inclusive): what a drop-in caller holding NumPy arrays sees
    sm, dyn, lim = mp.load_robot("ur5")
    model = _hip.HipModel(dyn.S_list, dyn.Mlist_per_link, dyn.Glist, sm.M_list, lim)
    ctx.specialize(model)
    B, N, n = 4096, 1000, 6
    q, qd, qdd = (rng.uniform(-1, 1, (B * N, n)).astype(np.float32) for _ in range(3))
    st, en = (rng.uniform(-1, 1, (B, n)).astype(np.float32) for _ in range(2))
    import time
    pq, pqd, pqdd, ptau = (ctx.pinned_empty((B * N, n), np.float32) for _ in range(4))
    pq[:], pqd[:], pqdd[:] = q, qd, qdd
    ptau3 = ctx.pinned_empty((B, N, n), np.float32)
    reuse = np.empty((B * N, n), np.float32)
    for name, fn, by in (("id_trajectory_host (pageable in, fresh out)", lambda: ctx.id_trajectory_host(model, q, qd, qdd), 16 * B * N * n),
                         ("id_trajectory_host (pageable in, reused out)", lambda: ctx.id_trajectory_host(model, q, qd, qdd, out=reuse), 16 * B * N * n),
                         ("id_trajectory_host (pinned in/out)", lambda: ctx.id_trajectory_host(model, pq, pqd, pqdd, out=ptau), 16 * B * N * n),
                         ("traj_id_fused_host (fresh out)", lambda: ctx.traj_id_fused_host(model, st, en, 2.0, N, 5), 4 * B * N * n),
                         ("traj_id_fused_host (pinned out)", lambda: ctx.traj_id_fused_host(model, st, en, 2.0, N, 5, out=ptau3), 4 * B * N * n)):
        fn()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        out.append(dict(op=name, robot="ur5", dtype="float32", rows=B * N, ms=ms, rows_per_s=B * N / ms * 1e3,
                        jt_per_s=B * N * n / ms * 1e3, pcie_GBps=by / ms / 1e6))
    np.testing.assert_array_equal(ptau, reuse)
    # FK + Jacobian of 2 M configurations back to the host: 36 B up, 464 B down per row
    R = 1 << 21
    q64 = rng.uniform(-1, 1, (R, n))
    pq64 = ctx.pinned_empty((R, n), np.float64); pq64[:] = q64
    oT, oJ = ctx.pinned_empty((R, 4, 4), np.float64), ctx.pinned_empty((R, 6, n), np.float64)
    for name, fn in (("fk_jac_host (pageable in, fresh out)", lambda: ctx.fk_jac_id_host(model, q64)),
                     ("fk_jac_host (pinned in/out)", lambda: ctx.fk_jac_id_host(model, pq64, out_T=oT, out_J=oJ))):
        fn()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        out.append(dict(op=name, robot="ur5", dtype="float64", rows=R, ms=ms, rows_per_s=R / ms * 1e3,
                        pcie_GBps=R * (n * 8 + 128 + 48 * n) / ms / 1e6))
    # the roll-out from host arrays (PCIe inclusive), (B, N, n) in and out: the batch-major kernel directly, against device
    # transposes around the time-major kernel; and the device transpose alone
    sm, dyn, lim = mp.load_robot("xarm6")
    xm = _hip.HipModel(dyn.S_list, dyn.Mlist_per_link, dyn.Glist, sm.M_list, lim)
    ctx.specialize(xm)
    B, N, n = 32768, 100, 6
    th0 = rng.uniform(-0.5, 0.5, (B, n)).astype(np.float32); dth0 = rng.uniform(-0.2, 0.2, (B, n)).astype(np.float32)
    hold = ctx.id_trajectory_host(xm, th0, np.zeros_like(th0), np.zeros_like(th0), [0, 0, -9.81], None, dtype=np.float32)
    tm = (hold[:, None, :] + rng.uniform(-1e-3, 1e-3, (B, N, n))).astype(np.float32)
    Fm = rng.uniform(-0.02, 0.02, (B, N, 6)).astype(np.float32)
    tmT, FmT = np.ascontiguousarray(np.swapaxes(tm, 0, 1)), np.ascontiguousarray(np.swapaxes(Fm, 0, 1))
    pin32 = lambda a: (lambda b: (b.__setitem__(slice(None), a), b)[1])(ctx.pinned_empty(a.shape, np.float32))
    p_th0, p_dth0, p_tm, p_Fm = pin32(th0), pin32(dth0), pin32(tm), pin32(Fm)
    p_out = [ctx.pinned_empty(tm.shape, np.float32) for _ in range(3)]
    for name, fn in (("fd_trajectory_host (B,N,n) arrays, batch-major kernel", lambda: ctx.fd_trajectory_host(xm, th0, dth0, tm, None, Fm, 0.01, 1, dtype=np.float32)),
                     ("fd_trajectory_host (B,N,n) PAGE-LOCKED arrays in / out: chunked upload / roll-out / download pipeline",
                      lambda: ctx.fd_trajectory_host(xm, p_th0, p_dth0, p_tm, None, p_Fm, 0.01, 1, dtype=np.float32, out=p_out)),
                     ("fd_trajectory_host (B,N,n) arrays, device transposes + time-major kernel",
                      lambda: ctx.fd_trajectory_host(xm, th0, dth0, tm, None, Fm, 0.01, 1, dtype=np.float32, device_layout="time_major")),
                     ("fd_trajectory_host (N,B,n) arrays, time-major kernel",
                      lambda: ctx.fd_trajectory_host(xm, th0, dth0, tmT, None, FmT, 0.01, 1, dtype=np.float32, layout="time_major"))):
        fn()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        out.append(dict(op=name, robot="xarm6", dtype="float32", trajectories=B, steps=N, ms=ms, jt_per_s=B * N * n / ms * 1e3,
                        pcie_GBps=B * N * (4 * n + 6) * 4 / ms / 1e6))
    Bt = 131072
    src, dst = ctx.alloc(Bt * N * n * 4), ctx.alloc(Bt * N * n * 4)
    for name, o_, i_ in (("transpose_rows (B,N,24 B) -> (N,B,24 B)", Bt, N), ("transpose_rows (N,B,24 B) -> (B,N,24 B)", N, Bt)):
        ms = timed(ctx, lambda: ctx.transpose_rows(src, o_, i_, n * 4, dst))
        out.append(dict(op=name, dtype="float32", rows=Bt * N, ms=ms, alg_GBps=2 * Bt * N * n * 4 / ms / 1e6))
    for r in out:
        print(json.dumps(r))
    ctx.destroy()


if __name__ == "__main__":
    main()
