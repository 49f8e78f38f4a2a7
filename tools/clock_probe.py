#!/usr/bin/env python3
"""Does the shader-clock sampler (mp_clock_sample_begin / _end) cost the kernels it runs beside, and does a configuration's clock
settle within bench.py's 60 ms ramp?  Alternates K launches of a configuration's step without and with the sampler, for several K.

    python3 tools/clock_probe.py [c2|c4|c4s] > gpurun_out/clock_probe.txt
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from manipulapy_amd import _hip, robots  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
robot, B, N = {"c2": ("ur5", 4096, 1000), "c4": ("panda", 32768, 200), "c4s": ("panda7", 32768, 200)}[name]
ctx = _hip.HipContext(0)
t = robots.robot_tables(robot)
n = t["S_list"].shape[1]
model = _hip.HipModel(t["S_list"], t["Mlist_per_link"], t["Glist"], t["M_ee"], t["joint_limits"])
ctx.specialize(model)
rows = B * N
lo, hi = t["joint_limits"][:, 0], t["joint_limits"][:, 1]
sets = []
for k in range(3):
    rng = np.random.default_rng(100 + k)
    ds, de = ctx.to_device(rng.uniform(lo, hi, (B, n)).astype(np.float32)), ctx.to_device(rng.uniform(lo, hi, (B, n)).astype(np.float32))
    bufs = [ctx.alloc(rows * n * 4) for _ in range(4)]
    ctx.batch_trajectory(model, ds, de, B, N, 2.0, 5, *bufs[:3])
    sets.append(bufs)
ctx.synchronize()
turn = [0]


def step():
    q, qd, qdd, tau = sets[turn[0] % 3]
    turn[0] += 1
    ctx.id_trajectory(model, q, qd, qdd, rows, tau, dtype=np.float32)


def run(K, sample, est_ms):
    a, b = ctx.event(), ctx.event()
    ctx.synchronize()
    if sample:
        ctx.clock_sample_begin(max(0.3, 0.8 * K * est_ms))
    a.record()
    for _ in range(K):
        step()
    b.record()
    ctx.synchronize()
    hz = ctx.clock_sample_end()[0] if sample else 0.0
    ms = b.elapsed_ms_since(a) / K
    a.destroy(); b.destroy()
    return ms, hz


t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.06:   # bench.py's ramp
    for _ in range(8):
        step()
    ctx.synchronize()
est = run(50, False, 0.1)[0]
print(f"{name}: first 50 launches after a 60 ms ramp: {est * 1e3:.2f} us per launch")
for K in (50, 200, 1000, 50):
    for rnd in range(4):
        plain, _ = run(K, False, est)
        with_s, hz = run(K, True, est)
        print(f"K={K:5d} round {rnd}: plain {plain * 1e3:7.2f} us   beside the sampler {with_s * 1e3:7.2f} us ({with_s / plain:5.3f} x)   clock {hz / 1e9:.3f} GHz")
print("after 0.5 s idle:")
for rnd in range(3):
    time.sleep(0.5)
    with_s, hz = run(5, True, est)
    time.sleep(0.5)
    plain, _ = run(5, False, est)
    print(f"K=5 cold round {rnd}: plain {plain * 1e3:7.2f} us   beside the sampler {with_s * 1e3:7.2f} us   clock {hz / 1e9:.3f} GHz")
# How long an idle phase must be before the first launches pay for it, and WHAT they pay: after `idle` seconds without GPU work,
# one launch (t1) and five launches (t5), each behind its own idle phase, each bracketed by one event pair; median of five rounds.
# t5 / 5 is bench.py's kernel_ms_cold; (t5 - t1) / 4 is what the launches cost without the wake-up (event -> first kernel start);
# t1 - the sustained kernel time is that wake-up.
run(2000, False, est)                  # well into the sustained state before the sweep starts
print("after an idle phase of the given length (median of 5 rounds): t5 / 5 | (t5 - t1) / 4 | wake-up = t1 - sustained kernel | clock in the window")
for idle in (0.0, 0.001, 0.005, 0.02, 0.1, 0.5, 2.0):
    t1s, t5s, clocks = [], [], []
    for rnd in range(5):
        run(200, False, est)           # back to the sustained state
        time.sleep(idle)
        t1s.append(run(1, False, est)[0])
        run(200, False, est)
        time.sleep(idle)
        t5s.append(run(5, False, est)[0] * 5)
        run(200, False, est)
        time.sleep(idle)
        clocks.append(run(5, True, est)[1])
    t1s.sort(); t5s.sort(); clocks.sort()
    t1, t5 = t1s[2], t5s[2]
    print(f"idle {idle * 1e3:7.1f} ms: {t5 / 5 * 1e3:7.2f} us ({t5 / 5 / est:5.3f} x) | {(t5 - t1) / 4 * 1e3:7.2f} us ({(t5 - t1) / 4 / est:5.3f} x) | "
          f"{(t1 - est) * 1e3:6.1f} us | {clocks[2] / 1e9:.3f} GHz")
ctx.destroy()
