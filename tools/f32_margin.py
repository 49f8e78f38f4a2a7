"""How far the float32 inverse dynamics sits from the suite's bound on the bench's OWN rows, without a GPU.

    python tools/f32_margin.py c2 [sets]      (c2 / c4 / c4s; `sets` input sets, default all the bench rotates over)

Replays bench.py's seeded start / end pairs (SEED + cid + 100000 k) through the reference's trajectory generation (oracle/),
evaluates the product's float32 CPU launcher (the SAME mp_rnea template the kernels instantiate, csrc/mp_cpu.cpp) and the pinned C
oracle on every row, and prints the distribution of  err / (1e-4 |ref| + 5e-6 max|row|)  per row.  Test / record tool: it imports
oracle/, so it is not part of the product.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from manipulapy_amd import _hip, robots  # noqa: E402
from oracle import c_oracle  # noqa: E402
from oracle import ref_numpy as ref  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    cfg = dict(bench.CONFIGS[name], name=name)
    t = robots.robot_tables(cfg["robot"])
    n = t["S_list"].shape[1]
    tab = bench.oracle_tables(ref, cfg["robot"])
    B, N = cfg["B"], cfg["N"]
    cid = {"c2": 2, "c2f": 2, "c3": 3, "c4": 4, "c4s": 4}[name]
    alg = 16 * n * B * N
    nsets = int(sys.argv[2]) if len(sys.argv) > 2 else int(min(4, max(1, -(-1_100_000_000 // alg))))
    lo, hi = t["joint_limits"][:, 0], t["joint_limits"][:, 1]
    m = _hip.HipModel(t["S_list"], t["Mlist_per_link"], t["Glist"], t["M_ee"], t["joint_limits"])
    edges = [0.25, 0.5, 0.75, 1.0]
    counts = np.zeros(len(edges), dtype=np.int64)
    worst, rows_total, max_abs_err, in_f64, waves, waves_f64 = 0.0, 0, 0.0, 0, 0, 0
    t0 = time.time()
    for k in range(nsets):
        rng = np.random.default_rng(bench.SEED + cid + 100_000 * k)
        start = rng.uniform(lo, hi, (B, n)).astype(np.float32)
        end = rng.uniform(lo, hi, (B, n)).astype(np.float32)
        chunk = max(1, 1_000_000 // N)
        for b0 in range(0, B, chunk):
            o = ref.batch_joint_trajectory(t["joint_limits"], start[b0:b0 + chunk], end[b0:b0 + chunk], 2.0, N, 5)
            q, qd, qdd = (np.ascontiguousarray(o[key].reshape(-1, n), dtype=np.float32) for key in ("positions", "velocities", "accelerations"))
            want = c_oracle.inverse_dynamics_rows(tab, q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64))[0]
            got = _hip.cpu_id_trajectory(m, q, qd, qdd, dtype=np.float32)
            hard = _hip.cpu_id_row_precision(m, q, qd, qdd)
            in_f64 += int(hard.sum())
            w = hard[:len(hard) // 64 * 64].reshape(-1, 64).any(axis=1)
            waves += len(w); waves_f64 += int(w.sum())
            err = np.abs(got.astype(np.float64) - want)
            tol = 1e-4 * np.abs(want) + bench.F32_ROW * np.abs(want).max(axis=1, keepdims=True)
            ratio = (err / tol).max(axis=1)
            for i, e in enumerate(edges):
                counts[i] += int((ratio > e).sum())
            worst = max(worst, float(ratio.max()))
            max_abs_err = max(max_abs_err, float(err.max()))
            rows_total += len(ratio)
        print(f"# set {k}: {rows_total} rows so far, worst {worst:.3f}, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    print(json.dumps({"config": name, "robot": cfg["robot"], "rows": rows_total, "input_sets": nsets,
                      "bound": "1e-4 |ref| + 5e-6 max|row|", "worst_over_bound": worst, "max_abs_err": max_abs_err,
                      "rows_over": {str(e): int(c) for e, c in zip(edges, counts)},
                      "rows_evaluated_in_float64": in_f64, "share_of_rows": in_f64 / max(rows_total, 1),
                      "share_of_64_row_waves_with_such_a_row": waves_f64 / max(waves, 1),
                      "evaluator": "mp_id_trajectory_cpu float32 (the kernels' mp_rnea template on the host) vs oracle/oracle.c"}))


if __name__ == "__main__":
    main()
