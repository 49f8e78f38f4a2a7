#!/usr/bin/env python3
"""Dev probe: cost structure of hipGraph launches of K back-to-back c2 kernels vs K stream launches."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import manipulapy_amd as mp
from manipulapy_amd import _hip

ctx = _hip.HipContext(0)
sm, dyn, lim = mp.load_robot("ur5")
model = _hip.HipModel(dyn.S_list, dyn.Mlist_per_link, dyn.Glist, sm.M_list, lim)
ctx.specialize(model)
rows, n = 4096 * 1000, 6
rng = np.random.default_rng(0)
q = ctx.to_device(rng.uniform(-1, 1, (rows, n)).astype(np.float32))
qd = ctx.to_device(rng.uniform(-1, 1, (rows, n)).astype(np.float32))
qdd = ctx.to_device(rng.uniform(-1, 1, (rows, n)).astype(np.float32))
tau = ctx.alloc(rows * n * 4)
step = lambda: ctx.id_trajectory(model, q, qd, qdd, rows, tau)
for _ in range(5): step()
ctx.synchronize()
for K in (1, 10, 50, 200):
    with ctx.capture() as cap:
        for _ in range(K): step()
    g = cap.graph
    g.launch(); ctx.synchronize()
    for rep in range(3):
        a, b = ctx.event(), ctx.event()
        t0 = time.perf_counter(); a.record(); g.launch(); b.record(); t1 = time.perf_counter(); ctx.synchronize(); t2 = time.perf_counter()
        print(f"graph K={K}: cpu launch {1e3*(t1-t0):.3f} ms, wall {1e3*(t2-t0):.3f} ms, events {b.elapsed_ms_since(a):.3f} ms -> {1e3*(t2-t0)/K*1e3:.1f} us/step")
    # two graphs back to back
    t0 = time.perf_counter(); g.launch(); g.launch(); ctx.synchronize(); t2 = time.perf_counter()
    print(f"  2 x graph K={K}: wall {1e3*(t2-t0):.3f} ms -> {1e3*(t2-t0)/(2*K)*1e3:.1f} us/step")
    t0 = time.perf_counter()
    for _ in range(K): step()
    t1 = time.perf_counter(); ctx.synchronize(); t2 = time.perf_counter()
    print(f"stream K={K}: cpu enqueue {1e3*(t1-t0):.3f} ms, wall {1e3*(t2-t0):.3f} ms -> {1e3*(t2-t0)/K*1e3:.1f} us/step")
    g.destroy()
