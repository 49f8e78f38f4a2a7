#!/usr/bin/env python3
"""would the batch-major roll-out entry point be faster as transpose -> time-major kernel -> transpose back?

c5's arrays on the device, batch-major (B, N, n): (a) mp_fd_trajectory_f32 on them as they are (4-step LDS tiles), against
(b) mp_transpose_rows of taumat and Ftipmat to (N, B, *), mp_fd_trajectory_tm_f32, mp_transpose_rows of pos / vel / acc back -
each leg timed with HIP events, several B.  The transposes move 2 x the roll-out's own bytes (every array once in, once out).

    python3 tools/c5b_transpose_ab.py > gpurun_out/c5b_transpose.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from manipulapy_amd import _hip, robots  # noqa: E402

ctx = _hip.HipContext(0)
t = robots.robot_tables("xarm6")
n = t["S_list"].shape[1]
model = _hip.HipModel(t["S_list"], t["Mlist_per_link"], t["Glist"], t["M_ee"], t["joint_limits"])
ctx.specialize(model)
g = np.array([0.0, 0.0, -9.81])
N = 100
rng = np.random.default_rng(5)


def timed(fn, reps):
    for _ in range(3):
        fn()
    a, b = ctx.event(), ctx.event()
    ctx.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    ctx.synchronize()
    ms = b.elapsed_ms_since(a) / reps
    a.destroy(); b.destroy()
    return ms


for B in (4096, 16384, 65536, 131072, 524288):
    th0 = rng.uniform(-0.5, 0.5, (B, n)).astype(np.float32)
    dth0 = rng.uniform(-0.2, 0.2, (B, n)).astype(np.float32)
    hold = ctx.id_trajectory_host(model, th0, np.zeros_like(th0), np.zeros_like(th0), g, None, dtype=np.float32)
    tau = (hold[:, None, :] + rng.uniform(-1e-3, 1e-3, (B, N, n)).astype(np.float32)).astype(np.float32)
    F = (np.array([1.0, -2.0, 0.5, 3.0, -1.5, 0.75], np.float32) * 0.02 * rng.uniform(0.5, 1.0, (B, N, 1)).astype(np.float32)).astype(np.float32)
    d_th0, d_dth0, d_tau, d_F = ctx.to_device(th0), ctx.to_device(dth0), ctx.to_device(tau), ctx.to_device(F)
    nb = B * N * n * 4
    outs = [ctx.alloc(nb) for _ in range(3)]
    t_tau, t_F = ctx.alloc(nb), ctx.alloc(B * N * 6 * 4)
    t_outs = [ctx.alloc(nb) for _ in range(3)]
    back = [ctx.alloc(nb) for _ in range(3)]
    reps = max(5, min(50, int(2e7 / (B * N))))

    def direct():
        ctx.fd_trajectory(model, d_th0, d_dth0, d_tau, d_F, B, N, g, 0.01, 1, *outs, dtype=np.float32, time_major=False)

    def tin():
        ctx.transpose_rows(d_tau, B, N, n * 4, t_tau)
        ctx.transpose_rows(d_F, B, N, 24, t_F)

    def tm():
        ctx.fd_trajectory(model, d_th0, d_dth0, t_tau, t_F, B, N, g, 0.01, 1, *t_outs, dtype=np.float32, time_major=True)

    def tout():
        for src, dst in zip(t_outs, back):
            ctx.transpose_rows(src, N, B, n * 4, dst)

    def whole():
        tin(); tm(); tout()

    ms = {k: timed(f, reps) for k, f in (("batch-major kernel", direct), ("transposes in", tin), ("time-major kernel", tm), ("transposes out", tout), ("in + kernel + out", whole))}
    got = [(a.download((B, N, n), np.float32), b.download((B, N, n), np.float32)) for a, b in zip(outs, back)]
    same = all(np.array_equal(x, y, equal_nan=True) for x, y in got)                      # (a roll-out that overflows is NaN from there on, in both routes)
    bad = int((~np.isfinite(got[0][0]).all(axis=(1, 2))).sum())
    moved = 2 * (2 * nb + B * N * 24 + 3 * nb - nb)  # bytes the five transposes read + write
    print(f"B = {B:7d}: " + ", ".join(f"{k} {v:.4f} ms" for k, v in ms.items())
          + f"; transposes {moved / 1e9:.2f} GB at {moved / 1e6 / (ms['transposes in'] + ms['transposes out']):.0f} GB/s; the two routes' results bit-equal: {same} ({bad} non-finite trajectories)")
    for b_ in [d_th0, d_dth0, d_tau, d_F, t_tau, t_F] + outs + t_outs + back:
        b_.free()
    ctx.trim_pool()
ctx.destroy()
