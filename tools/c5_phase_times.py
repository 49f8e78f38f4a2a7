#!/usr/bin/env python3
"""Experiment: per-wave cycle totals of the roll-out kernel's tile phases (needs MANIPULAPY_HIP_JIT_DEFINES=MP_FD_EXP_TIMING[=2]).
usage: python tools/c5_phase_times.py [B] [N]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manipulapy_amd import _hip, robots  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t = robots.robot_tables("xarm6")
n = 6
ctx = _hip.HipContext(0)
model = _hip.HipModel(t["S_list"], t["Mlist_per_link"], t["Glist"], t["M_ee"], t["joint_limits"])
ctx.specialize(model)
rng = np.random.default_rng(5)
th0 = rng.uniform(-0.5, 0.5, (B, n)).astype(np.float32)
dth0 = rng.uniform(-0.2, 0.2, (B, n)).astype(np.float32)
g = np.array([0.0, 0.0, -9.81])
hold = ctx.id_trajectory_host(model, th0, np.zeros_like(th0), np.zeros_like(th0), g, None, dtype=np.float32)
tau = (hold[:, None, :] + rng.uniform(-1, 1, (B, N, n)).astype(np.float32) * np.float32(1e-3)).astype(np.float32)
Fm = (np.array([1, 1, 1, 0.25, 0.25, 0.25], np.float32) * 0.02 * rng.uniform(0.5, 1.0, (B, N, 1)).astype(np.float32)).astype(np.float32)
d = [ctx.to_device(x) for x in (th0, dth0, tau, Fm)]
ob = B * N * n * 4
o = [ctx.alloc(ob) for _ in range(3)]
for _ in range(5):
    ctx.fd_trajectory(model, d[0], d[1], d[2], d[3], B, N, g, 0.01, 1, o[0], o[1], o[2], dtype=np.float32)
ctx.synchronize()
acc = o[2].download((B, N, n), np.float32)
w = acc[::64, 1, :6].astype(np.float64)   # one row per wave: in, compute, out, total, start stamp, placement
tiles = (N + 3) // 4
print(f"B {B} N {N} waves {len(w)} defines {os.environ.get('MANIPULAPY_HIP_JIT_DEFINES', '')}")
for k, name in enumerate(("input half", "integration", "flush", "wave total")):
    v = w[:, k]
    per = v / (tiles if k < 3 else 1)
    print(f"  {name:12s} cycles per {'tile' if k < 3 else 'wave'}: mean {per.mean():10.0f}  p10 {np.percentile(per, 10):10.0f}  p50 {np.percentile(per, 50):10.0f}  p90 {np.percentile(per, 90):10.0f}")
print(f"  shares of the wave total: in {w[:,0].sum()/w[:,3].sum():.3f}  compute {w[:,1].sum()/w[:,3].sum():.3f}  flush {w[:,2].sum()/w[:,3].sum():.3f}")
start = (w[:, 4] - w[:, 4].min()) % (1 << 24)
end = start + w[:, 3]
print(f"  wave start (cycles after the first): p50 {np.percentile(start, 50):.0f} p90 {np.percentile(start, 90):.0f} max {start.max():.0f};  last end {end.max():.0f}")
hist, edges = np.histogram(start, bins=12)
print("  start histogram:", [(int(e), int(h)) for e, h in zip(edges[:-1], hist)])
hw = w[:, 5].astype(np.int64)
xcc, se, cu, simd = (hw >> 16) & 15, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3
place = xcc * 10000 + se * 100 + cu
u, cnt = np.unique(place, return_counts=True)
print(f"  distinct CUs {len(u)}; waves per CU: min {cnt.min()} max {cnt.max()} histogram {np.bincount(cnt).tolist()}")
us, cs = np.unique(place * 10 + simd, return_counts=True)
print(f"  distinct SIMDs {len(us)}; waves per SIMD histogram {np.bincount(cs).tolist()}")
late = start > 0.25 * w[:, 3].mean()
print(f"  waves that started late: {late.sum()}; their mean total {w[late, 3].mean() if late.any() else 0:.0f} vs early {w[~late, 3].mean():.0f}")
print(f"  wave total: max {w[:,3].max():.0f}  p99 {np.percentile(w[:,3], 99):.0f}")
for x in np.unique(xcc):
    m = xcc == x
    print(f"  XCC {x}: waves {m.sum():4d} total mean {w[m,3].mean():9.0f} max {w[m,3].max():9.0f} | per tile: in {w[m,0].mean()/tiles:6.0f} compute {w[m,1].mean()/tiles:6.0f} flush {w[m,2].mean()/tiles:6.0f}")
# inside one XCC: by SE and by CU
m = xcc == np.unique(xcc)[0]
for s_ in np.unique(se[m]):
    mm = m & (se == s_)
    print(f"   XCC {np.unique(xcc)[0]} SE {s_}: waves {mm.sum():3d} total mean {w[mm,3].mean():9.0f} max {w[mm,3].max():9.0f} flush/tile {w[mm,2].mean()/tiles:6.0f}; CUs {sorted(set(cu[mm].tolist()))}")
wid = hw & 15
for k in np.unique(wid):
    mm = wid == k
    print(f"   wave slot {k}: waves {mm.sum():4d} total mean {w[mm,3].mean():9.0f} flush/tile {w[mm,2].mean()/tiles:6.0f} compute/tile {w[mm,1].mean()/tiles:6.0f}")
# partner waves of a SIMD: do they take the same time?
key = place * 10 + simd
order = np.argsort(key, kind="stable")
pairs = w[order, 3].reshape(-1, 2)
print(f"  partners on a SIMD: mean |difference| of totals {np.abs(pairs[:,0]-pairs[:,1]).mean():.0f}; mean of slower {pairs.max(1).mean():.0f} faster {pairs.min(1).mean():.0f}")
