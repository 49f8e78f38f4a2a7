"""Dev probe: mp_id_trajectory_host_f32 on page-locked arrays (the chunked upload / kernel / download pipeline), c2-sized, ms per call.
    [MANIPULAPY_HIP_LEAD=0 MANIPULAPY_HIP_EXPERIMENT=1 MANIPULAPY_HIP_JIT_DEFINES=MP_ID_LEAD=0] python tools/host_pinned_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import manipulapy_amd as mp  # noqa: E402
from manipulapy_amd import _hip  # noqa: E402

t = mp.robot_tables("ur5")
ctx = _hip.HipContext(0)
model = _hip.HipModel(t["S_list"], t["Mlist_per_link"], t["Glist"], t["M_ee"], t["joint_limits"])
ctx.specialize(model)
rng = np.random.default_rng(0)
B, N, n = 4096, 1000, 6
pq, pqd, pqdd, ptau = (ctx.pinned_empty((B * N, n), np.float32) for _ in range(4))
for a in (pq, pqd, pqdd):
    a[:] = rng.uniform(-1, 1, (B * N, n)).astype(np.float32)
for rep in range(3):
    ctx.id_trajectory_host(model, pq, pqd, pqdd, out=ptau)
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.id_trajectory_host(model, pq, pqd, pqdd, out=ptau)
    print(f"id_trajectory_host pinned in/out: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per call", flush=True)
ctx.destroy()
