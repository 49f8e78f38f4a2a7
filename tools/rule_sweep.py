"""STUDY TOOL: which conditioning test separates the float32 rows that need float64, and at what cost in flagged rows.

    python tools/rule_sweep.py c2 [sets] [--robot NAME]

For bench.py's own seeded rows of a configuration (as tools/f32_margin.py) it evaluates, per row, the kernels' float32 recursion,
the float64 recursion on the same float32 inputs and the largest force component of every joint (tools/rule_sweep.cpp: the
product's templates on the host), the pinned C oracle (cached under /tmp/rule_sweep_cache), and then prints for candidate rules
    flagged  <=>  lscale * max_{i in JOINTS} |f_i|_inf  >  K * max|tau|
the share of rows flagged, the worst  err / (1e-4 |ref| + 5e-6 max|row|)  among the rows that stay float32 and among the flagged
ones.  Imports oracle/: a record tool, not part of the product.
"""
import ctypes
import itertools
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from manipulapy_amd import robots  # noqa: E402
from oracle import c_oracle  # noqa: E402
from oracle import ref_numpy as ref  # noqa: E402

CACHE = "/tmp/rule_sweep_cache"
LIB = "/tmp/librule_sweep.so"


def build():
    src = [os.path.join(ROOT, "tools", "rule_sweep.cpp"), os.path.join(ROOT, "manipulapy_amd", "csrc", "mp_model_compile.cpp")]
    hdr = [os.path.join(ROOT, "manipulapy_amd", "csrc", h) for h in ("mp_core.h", "mp_model.h", "mp_model_compile.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(f) > os.path.getmtime(LIB) for f in src + hdr):
        subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-O2", "-std=c++17", "-ffp-contract=fast", "-fopenmp", "-shared", "-fPIC",
                        "-o", LIB] + src, check=True)
    return ctypes.CDLL(LIB)


def rows_of(cfg, cid, k, t, n):
    B, N = cfg["B"], cfg["N"]
    lo, hi = t["joint_limits"][:, 0], t["joint_limits"][:, 1]
    rng = np.random.default_rng(bench.SEED + cid + 100_000 * k)
    start = rng.uniform(lo, hi, (B, n)).astype(np.float32)
    end = rng.uniform(lo, hi, (B, n)).astype(np.float32)
    chunk = max(1, 1_000_000 // N)
    for b0 in range(0, B, chunk):
        o = ref.batch_joint_trajectory(t["joint_limits"], start[b0:b0 + chunk], end[b0:b0 + chunk], 2.0, N, 5)
        yield b0, tuple(np.ascontiguousarray(o[key].reshape(-1, n), dtype=np.float32) for key in ("positions", "velocities", "accelerations"))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    name = args[0] if args else "c2"
    cfg = dict(bench.CONFIGS[name], name=name)
    robot = cfg["robot"]
    if "--robot" in sys.argv:
        robot = sys.argv[sys.argv.index("--robot") + 1]
    max_b = int(sys.argv[sys.argv.index("--traj") + 1]) if "--traj" in sys.argv else cfg["B"]
    cfg["B"] = min(cfg["B"], max_b)
    t = robots.robot_tables(robot)
    n = t["S_list"].shape[1]
    tab = bench.oracle_tables(ref, robot)
    cid = {"c2": 2, "c2f": 2, "c3": 3, "c4": 4, "c4s": 4}[name]
    nsets = int(args[1]) if len(args) > 1 else 1
    lib = build()
    os.makedirs(CACHE, exist_ok=True)
    dp = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    fp = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    R32, R64, FM, MM, TM, EX = [], [], [], [], [], []
    lever = np.zeros(2 * n + 1)
    t0 = time.time()
    for k in range(nsets):
        for b0, (q, qd, qdd) in rows_of(cfg, cid, k, t, n):
            key = os.path.join(CACHE, f"{name}_{robot}_{cfg['B']}_set{k}_b{b0}.npy")
            if os.path.exists(key):
                want = np.load(key)
            else:
                want = c_oracle.inverse_dynamics_rows(tab, q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64))[0]
                np.save(key, want)
            rows = len(q)
            t32 = np.empty((rows, n), np.float32); t64 = np.empty((rows, n), np.float32); fm = np.empty((rows, n), np.float32); mm = np.empty((rows, n), np.float32); ex = np.empty((rows, 4, n), np.float32)
            err = ctypes.create_string_buffer(256)
            g = np.array([0.0, 0.0, -9.81])
            rc = lib.rule_sweep(ctypes.c_int(n), dp(t["S_list"]), dp(t["Mlist_per_link"]), dp(t["Glist"]), dp(t["M_ee"]), dp(t["joint_limits"]),
                                dp(g), ctypes.c_long(rows), fp(q), fp(qd), fp(qdd), fp(t32), fp(t64), fp(fm), fp(mm), fp(ex),
                                lever.ctypes.data_as(ctypes.c_void_p), err, ctypes.c_long(256))
            assert rc == 0, err.value
            tol = 1e-4 * np.abs(want) + bench.F32_ROW * np.abs(want).max(axis=1, keepdims=True)
            R32.append((np.abs(t32.astype(np.float64) - want) / tol).max(axis=1).astype(np.float32))
            R64.append((np.abs(t64.astype(np.float64) - want) / tol).max(axis=1).astype(np.float32))
            FM.append(fm); MM.append(mm); EX.append(ex); TM.append(np.abs(t32).max(axis=1))
            print(f"# set {k} b0 {b0}: {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    r32 = np.concatenate(R32); r64 = np.concatenate(R64); fm = np.concatenate(FM); mm = np.concatenate(MM); ex = np.concatenate(EX); tm = np.concatenate(TM)
    lscale = np.float32(lever[2 * n])
    print(json.dumps({"config": name, "robot": robot, "rows": int(len(r32)), "lever_a_d": lever[:2 * n].reshape(n, 2).round(4).tolist(),
                      "lscale": float(lscale), "plain_f32_worst": float(r32.max()), "rows_over_bound_plain": int((r32 > 1).sum()),
                      "f64_worst": float(r64.max())}))
    masks = {"all>=1": list(range(1, n))}
    for i in range(1, n):
        masks[f"j{i}"] = [i]
    for i, j in itertools.combinations(range(1, min(n, 5)), 2):
        masks[f"j{i}+j{j}"] = [i, j]
    masks["j1..3"] = [1, 2, 3]
    print(f"{'joints':10s} {'K':>5s} {'flagged %':>10s} {'worst f32':>10s} {'>0.5':>6s} {'>0.25':>7s} {'worst f64':>10s}")
    stats = {k: fm[:, v].max(axis=1) * lscale for k, v in masks.items()}
    if "--quick" in sys.argv:
        stats = {"j1": stats["j1"]}
    for i in range(0, min(n, 4)):
        stats[f"n{i}"] = mm[:, i]
    if n > 2:
        stats["n1|L*f2"] = np.maximum(mm[:, 1], fm[:, 2] * lscale)
        stats["n1+L*f2"] = mm[:, 1] + fm[:, 2] * lscale
        stats["n0|L*f2"] = np.maximum(mm[:, 0], fm[:, 2] * lscale)
    bm, bf, cm, cf = ex[:, 0], ex[:, 1], ex[:, 2], ex[:, 3]   # own body moment / force of link i; joint i's wrench as link i - 1 receives it
    stats["bm1|cm2"] = np.maximum(bm[:, 1], cm[:, 2])
    stats["bm1+cm2"] = bm[:, 1] + cm[:, 2]
    stats["bm0..|cm"] = np.maximum(bm[:, :3].max(axis=1), cm[:, 1:4].max(axis=1))
    stats["bm1|cm2|L*f2"] = np.maximum(np.maximum(bm[:, 1], cm[:, 2]), fm[:, 2] * lscale)
    stats["L*(bf1+f2)"] = (bf[:, 1] + fm[:, 2]) * lscale
    stats["sum bm"] = bm.sum(axis=1)
    stats["S3"] = np.maximum(np.maximum(bm[:, 1], cm[:, 2]), fm[:, 2] * lscale)
    stats["S3+"] = np.maximum(bm[:, 1] + cm[:, 2], fm[:, 2] * lscale)
    stats["bm0+bm1+cm2"] = bm[:, 0] + bm[:, 1] + cm[:, 2]
    stats["bm1+bm2+cm3"] = bm[:, 1] + bm[:, 2] + cm[:, 3]
    if "--only" in sys.argv:
        keep_names = sys.argv[sys.argv.index("--only") + 1].split(",")
        stats = {k: v for k, v in stats.items() if k in keep_names}
    for mname, stat in stats.items():
        for K in (3.0, 4.0, 5.0, 6.0, 8.0, 10.0, 12.0, 14.0, 16.0, 20.0):
            hard = stat * np.float32(1.0 / K) > tm
            keep = ~hard
            print(f"{mname:10s} {K:5.0f} {100 * hard.mean():10.3f} {r32[keep].max():10.3f} {int((r32[keep] > 0.5).sum()):6d} "
                  f"{int((r32[keep] > 0.25).sum()):7d} {(r64[hard].max() if hard.any() else 0):10.3f}")


if __name__ == "__main__":
    main()
