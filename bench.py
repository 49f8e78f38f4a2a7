#!/usr/bin/env python3
"""Benchmark of the hot path: joint-timesteps/s of inverse_dynamics_trajectory over B x N rows.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W          (one rank per GPU)

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in
HBM: config c2 (default, the one BASELINE.json's metric is quoted on) = UR5 (6 DOF), B = 4096
trajectories x N = 1000 timesteps, float32, materialised q / qd / qdd histories -> tau.  With more than
one rank every GPU gets its own B trajectories (weak scaling) and evaluates its shard with no exchange:
the path has no exchange step, so the timed step has no collective.  The RCCL all-gather that
reassembles the full torque history on every GPU is then measured in a second loop (step + all-gather)
and reported in the "allgather" object next to `value`.

torch is used ONLY for the multi-process rendezvous (gloo barrier / max / 128-byte id broadcast); the
compute path is ctypes -> libmanipula_hip.so.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (robot, B per GPU, N, dtype, what the step runs)
    "c2": dict(robot="ur5", B=4096, N=1000, dtype="f32", op="id",
               desc="UR5 6-DOF, B=4096 x N=1000, inverse_dynamics_trajectory fp32 (BASELINE configs[1])"),
    "c2f": dict(robot="ur5", B=4096, N=1000, dtype="f32", op="fused",
                desc="UR5 6-DOF, B=4096 x N=1000, joint_trajectory fused into inverse_dynamics_trajectory fp32: inputs are the "
                     "(B,n) start/end pairs, only tau is written (4 B per joint-timestep) - reported separately from c2, never mixed"),
    "c3": dict(robot="iiwa14", B=65536, N=500, dtype="f64", op="fk_jac_id",
               desc="KUKA iiwa14 7-DOF, B=65536 x N=500, FK + Jacobian + ID fused fp64 (BASELINE configs[2])"),
    "c4": dict(robot="panda", B=32768, N=200, dtype="f32", op="id",
               desc="Franka Panda (8 DOF as the reference parses it), B=32768/GPU x N=200, ID fp32 (BASELINE configs[3] per-GPU shard)"),
    "c4s": dict(robot="panda7", B=32768, N=200, dtype="f32", op="id",
                desc="Franka Panda, the 7 arm joints only (first-seven-joint truncation of the reference's 8-joint tables), "
                     "B=32768/GPU x N=200, ID fp32 - the 7-DOF reading of BASELINE configs[3]"),
    "c5": dict(robot="xarm6", B=131072, N=100, dtype="f32", op="fd_traj",
               desc="xArm6 (6 DOF; the reference ships no xArm7), gravity + per-step Ftip, B=131072/GPU x N=100, mass matrix + "
                    "forward-dynamics roll-out fp32, dt=0.01 intRes=1 (BASELINE configs[4] per-GPU shard)"),
}
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
SEED = 20260705


def algorithmic_bytes_per_row(cfg, n):
    """SURVEY §8(d).  id: read q, qd, qdd + write tau = 4 values per joint-timestep.
    fk_jac_id: in 3n, out n + 16 + 6n values per timestep."""
    w = 4 if cfg["dtype"] == "f32" else 8
    if cfg["op"] == "fd_traj":  # per timestep: in tau (n) + Ftip (6), out pos/vel/acc (3n float32)
        return (n + 6) * w + 3 * n * 4
    if cfg["op"] == "fused":  # write tau only (the 2*n*4 B per TRAJECTORY of start/end are negligible)
        return n * w
    return (4 * n) * w if cfg["op"] == "id" else (3 * n + n + 16 + 6 * n) * w


def kernel_name(cfg):
    spec = cfg.get("specialized")
    return {"id": ("mp_spec_id_pk_f0" if cfg["dtype"] == "f32" else "mp_spec_id_d_f0") if spec else
                  ("k_id_pk" if cfg["dtype"] == "f32" else "k_id"),
            "fused": "mp_spec_traj_id_pk_f0" if spec else "k_traj_id_pk_tab",
            "fk_jac_id": "mp_spec_fk_jac_id_d_f0" if spec else "k_fk_jac_id",
            "fd_traj": "mp_spec_fd_traj_f1" if spec else "k_fd_traj"}[cfg["op"]]


def oracle_tables(ref, robot):
    """The oracle's view of the tables bench.py runs on (fixtures, or a derived robot such as the 7-joint Panda)."""
    from manipulapy_amd import robots

    t = robots.robot_tables(robot)
    return ref.RobotTables(S=t["S_list"].astype(np.float64), M_ee=t["M_ee"].astype(np.float64), G=t["Glist"].astype(np.float64),
                           Mcom=t["Mlist_per_link"].astype(np.float64), joint_limits=t["joint_limits"].astype(np.float64),
                           B=t["B_list"].astype(np.float64), name=robot)


def cpu_baseline(robot, q, qd, qdd, budget_s=12.0):
    """The CPU oracle — the reference's algorithm (1 + 2n mass matrices per point, finite-difference Coriolis)
    restated in C (oracle/oracle.c, pinned to the reference's golden vectors) — timed on this box's host
    cores (OpenMP over rows) on a bounded sample of the SAME rows.  Reported, never shipped."""
    from oracle import c_oracle
    from oracle import ref_numpy as ref

    tab = oracle_tables(ref, robot)
    n = tab.n
    q, qd, qdd = (np.ascontiguousarray(x, dtype=np.float64) for x in (q, qd, qdd))
    probe = min(2048, q.shape[0])
    t0 = time.perf_counter()
    _, threads = c_oracle.inverse_dynamics_rows(tab, q[:probe], qd[:probe], qdd[:probe])
    rate = probe / max(time.perf_counter() - t0, 1e-6)            # rows/s incl. thread start-up
    rows = int(min(q.shape[0], max(probe, rate * budget_s)))
    t0 = time.perf_counter()
    tau, threads = c_oracle.inverse_dynamics_rows(tab, q[:rows], qd[:rows], qdd[:rows])
    dt = time.perf_counter() - t0
    return {"value": rows * n / dt, "unit": "joint-timesteps/s", "cores": threads, "kind": "port",
            "sample": f"first {rows} rows of the benchmark input, {dt:.1f} s on {threads} OpenMP thread(s); C restatement of the "
                      f"reference algorithm (the reference's own NumPy code runs ~40-80 ms per row, BASELINE.md)"}, tau


def emit(result):
    """The ONE JSON line, as the last line of stdout: native libraries (RCCL prints a version banner) write through C stdio,
    which is block-buffered on a pipe and would otherwise land after an early Python print when the process exits."""
    import ctypes

    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(result), flush=True)


def ramp(ctx, step, ms):
    """Setup, untimed: keep the GPU busy with the step for `ms` milliseconds (clock / power ramp after the idle setup phase;
    measured on c2: 0.080 ms per step right after start-up, 0.0747 ms once warm)."""
    t0 = time.perf_counter()
    n = 0
    while (time.perf_counter() - t0) * 1e3 < ms and n < 100000:
        for _ in range(8):
            step()
        ctx.synchronize()
        n += 8
    return n


FTIP_REF = np.array([1.0, -2.0, 0.5, 3.0, -1.5, 0.75])  # the reference's own non-zero wrench (tests/test_dynamics_golden.py:145)


def bench_fd(args, cfg, info, hg, ctx, model, t, props):
    """Config c5: B independent forward-dynamics roll-outs (mass matrix + bias + solve + integrate per step).
    Sequential in time, so the path is VALU-bound by construction; the HBM roofline line is reported as asked."""
    from oracle import ref_numpy as ref

    n = t["S_list"].shape[1]
    B, N, world = cfg["B"], cfg["N"], info.world
    rng = np.random.default_rng(SEED + 5 + 1000 * info.rank)
    th0 = rng.uniform(-0.5, 0.5, (B, n)).astype(np.float32)
    dth0 = rng.uniform(-0.2, 0.2, (B, n)).astype(np.float32)
    taumat = (rng.uniform(-1, 1, (B, N, n)) * 0.01).astype(np.float32)
    Fm = np.broadcast_to(FTIP_REF.astype(np.float32), (B, N, 6)).copy()
    g = np.array([0.0, 0.0, -9.81])
    d_th0, d_dth0, d_tau, d_F = ctx.to_device(th0), ctx.to_device(dth0), ctx.to_device(taumat), ctx.to_device(Fm)
    ob = B * N * n * 4
    d_pos, d_vel, d_acc = ctx.alloc(ob), ctx.alloc(ob), ctx.alloc(ob)

    def step():
        ctx.fd_trajectory(model, d_th0, d_dth0, d_tau, d_F, B, N, g, 0.01, 1, d_pos, d_vel, d_acc, dtype=np.float32)

    ramp(ctx, step, args.ramp_ms)
    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    a, b = ctx.event(), ctx.event()
    hg.barrier()
    ctx.synchronize()
    t0 = time.perf_counter()
    a.record()
    for k in range(args.steps):
        step()
    b.record()   # HIP events on the launch stream around the K launches of the timed region
    ctx.synchronize()
    dt = time.perf_counter() - t0
    hg.barrier()
    elapsed = hg.max(dt)
    kern_ms = b.elapsed_ms_since(a) / args.steps   # average launch period = kernel duration + dispatch gap
    alg_bytes = algorithmic_bytes_per_row(cfg, n) * B * N
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    result = {
        "metric": "joint-timesteps/sec (NxBxDOF) forward-dynamics trajectory", "value": B * N * n * world * args.steps / elapsed,
        "unit": "joint-timesteps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": cfg["dtype"], "data": "synthetic",
        "config": {"workload": cfg["desc"], "robot": cfg["robot"], "dof": n, "B_per_gpu": B, "N": N, "op": cfg["op"],
                   "kernel_variant": "robot-specialised (hiprtc)" if cfg["specialized"] else "generic",
                   "sharding": f"batch axis over {world} rank(s), no collective in the timed step"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": None, "kernel": kernel_name(cfg), "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes,
                     "kernel_ms_method": "HIP events on the launch stream around the K launches of the timed region, / K",
                     "note": "sequential in time: VALU-bound (mass matrix + solve per step), HBM line shown for reference"},
        "setup": {"ramp_ms": args.ramp_ms, "what": "untimed launches before the W warm-up steps (clock ramp)"},
        "device": props["name"],
    }
    traffic_file = os.path.join(ROOT, "profiles", f"traffic_{args.config}.json")
    if os.path.exists(traffic_file):  # measured HBM bytes per launch from the rocprofv3 --pmc passes
        with open(traffic_file) as f:
            tf = json.load(f)
        if tf.get("kernel") == result["roofline"]["kernel"]:  # counters were collected on this very kernel
            result["roofline"]["traffic"] = tf.get("hbm_bytes_per_launch")
    if info.rank == 0 and world == 1 and not args.no_cpu_baseline:
        tab = oracle_tables(ref, cfg["robot"])
        r0 = ref.forward_dynamics_trajectory(tab, th0[0].astype(np.float64), dth0[0].astype(np.float64), taumat[0, :12].astype(np.float64),
                                             g, Fm[0, :12].astype(np.float64), 0.01, 1)  # first 12 steps of trajectory 0
        tc = time.perf_counter()
        ref.forward_dynamics_trajectory(tab, th0[1].astype(np.float64), dth0[1].astype(np.float64), taumat[1, :12].astype(np.float64), g,
                                        Fm[1, :12].astype(np.float64), 0.01, 1)
        dtc = time.perf_counter() - tc
        pos = d_pos.download((B, N, n), np.float32)
        result["cpu_baseline"] = {"value": 11 * n / dtc, "unit": "joint-timesteps/s", "cores": 1, "kind": "port",
                                  "sample": f"11 integration steps of one trajectory, {dtc:.1f} s, single thread, NumPy oracle"}
        result["parity_sample"] = {"rows": 12, "max_abs_err": float(np.abs(pos[0, :12] - r0["positions"]).max())}
    if info.rank == 0:
        emit(result)
    ctx.destroy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--launch", default="stream", choices=("stream", "graph"),
                    help="stream: one host call per step; graph: the K timed steps captured into one hipGraph launch")
    ap.add_argument("--event-stride", type=int, default=0,
                    help="0: one HIP event pair around the whole timed region (default); S > 0: a pair around every S-th launch")
    ap.add_argument("--ramp-ms", type=float, default=60.0,
                    help="setup: run the step untimed for this long before the W warm-up steps so the GPU is at its sustained clocks")
    ap.add_argument("--no-gather", action="store_true", help="multi-GPU: skip the all-gather (compute-only figure)")
    ap.add_argument("--no-specialize", action="store_true", help="use the generic kernels (no run-time robot specialisation)")
    ap.add_argument("--B", type=int, default=0, help="experiments only: override the config's trajectories per GPU")
    ap.add_argument("--N", type=int, default=0, help="experiments only: override the config's timesteps")
    args = ap.parse_args()

    from manipulapy_amd import _hip, robots, sharding

    info = sharding.dist_env()
    world = info.world
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs one process per GPU: launch with torch.distributed.run "
                             f"--nproc-per-node {args.gpus}")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    hg = sharding.HostGather(info)  # gloo; no-op for a single process

    cfg = dict(CONFIGS[args.config])
    if args.B or args.N:
        cfg["B"], cfg["N"] = args.B or cfg["B"], args.N or cfg["N"]
        cfg["desc"] += f" [OVERRIDDEN: B={cfg['B']} N={cfg['N']}]"
    t = robots.robot_tables(cfg["robot"])
    n = t["S_list"].shape[1]
    B, N = cfg["B"], cfg["N"]
    rows = B * N
    dt_np = np.float32 if cfg["dtype"] == "f32" else np.float64
    wbytes = np.dtype(dt_np).itemsize

    dev = info.local_rank
    if os.environ.get("MANIPULAPY_BENCH_SHARE_DEVICE") == "1":  # development only: several ranks on one GPU
        dev = info.local_rank % max(_hip.device_count(), 1)
    ctx = _hip.HipContext(dev)
    ctx.selftest()
    props = ctx.properties()
    model = _hip.HipModel(t["S_list"], t["Mlist_per_link"], t["Glist"], t["M_ee"], t["joint_limits"])
    if not args.no_specialize:
        ctx.specialize(model)  # setup, untimed: hiprtc build of this robot's kernels (cached on disk)
    cfg["specialized"] = ctx.is_specialized(model)

    if cfg["op"] == "fd_traj":
        return bench_fd(args, cfg, info, hg, ctx, model, t, props)

    # ---- synthetic input, generated ON the device: start / end ~ U(joint limits), quintic, Tf = 2
    cid = {"c2": 2, "c2f": 2, "c3": 3, "c4": 4, "c4s": 4}[args.config]
    rng = np.random.default_rng(SEED + cid + 1000 * info.rank)
    lo, hi = t["joint_limits"][:, 0], t["joint_limits"][:, 1]
    start = rng.uniform(lo, hi, (B, n)).astype(np.float32)
    end = rng.uniform(lo, hi, (B, n)).astype(np.float32)
    d_start, d_end = ctx.to_device(start), ctx.to_device(end)
    nb32 = rows * n * 4
    d_q32, d_qd32, d_qdd32 = ctx.alloc(nb32), ctx.alloc(nb32), ctx.alloc(nb32)
    ctx.batch_trajectory(model, d_start, d_end, B, N, 2.0, 5, d_q32, d_qd32, d_qdd32)
    ctx.synchronize()
    if cfg["dtype"] == "f32":
        d_q, d_qd, d_qdd = d_q32, d_qd32, d_qdd32
    else:  # float64 configs: widen the same histories on the host once (setup, untimed)
        chunk = 1 << 22
        d_q, d_qd, d_qdd = (ctx.alloc(rows * n * 8) for _ in range(3))
        for src, dst in ((d_q32, d_q), (d_qd32, d_qd), (d_qdd32, d_qdd)):
            h = src.download((rows * n,), np.float32).astype(np.float64)
            dst.upload(h)
            del h
        for b in (d_q32, d_qd32, d_qdd32):
            b.free()
    nb = rows * n * wbytes
    d_tau = ctx.alloc(nb)
    d_T = ctx.alloc(rows * 16 * wbytes) if cfg["op"] == "fk_jac_id" else None
    d_J = ctx.alloc(rows * 6 * n * wbytes) if cfg["op"] == "fk_jac_id" else None

    def step():
        if cfg["op"] == "id":
            ctx.id_trajectory(model, d_q, d_qd, d_qdd, rows, d_tau, dtype=dt_np)
        elif cfg["op"] == "fused":
            ctx.traj_id_fused(model, d_start, d_end, B, N, 2.0, 5, d_tau)
        else:
            ctx.fk_jac_id(model, d_q, d_qd, d_qdd, rows, d_T, d_J, d_tau, dtype=dt_np)

    def timed(fn_after_step=None, step_fn=None):
        """W warm-up steps, then exactly K timed steps between barrier + device sync on both sides.
        Returns (max-over-ranks wall seconds, mean kernel ms from HIP events on the launch stream)."""
        do = step_fn or step
        ramp(ctx, step, args.ramp_ms)
        for _ in range(args.warmup):
            do()
            if fn_after_step:
                fn_after_step()
        ctx.synchronize()
        if args.launch == "graph" and fn_after_step is None and step_fn is None:
            # the K launches of the timed region are captured once (hipGraph) and submitted as ONE graph launch: the
            # queue then holds all K dispatch packets up front instead of receiving one per host call
            with ctx.capture() as cap:
                for _ in range(args.steps):
                    step()
            graph = cap.graph
            graph.launch()   # untimed: first launch uploads the graph
            ctx.synchronize()
            a, b = ctx.event(), ctx.event()
            hg.barrier()
            ctx.synchronize()
            t0 = time.perf_counter()
            a.record()
            graph.launch()
            b.record()       # HIP events on the launch stream around the K kernels of the timed region
            ctx.synchronize()
            dt = time.perf_counter() - t0
            hg.barrier()
            wall = hg.max(dt)
            kms = b.elapsed_ms_since(a) / args.steps   # average launch period: kernel + any residual inter-kernel gap
            a.destroy(); b.destroy(); graph.destroy()
            return wall, kms
        # HIP events on the launch stream around the K launches of the timed region: elapsed / K = the kernel's average
        # launch period (duration + the ~2 us dispatch gap), an upper bound on its duration.  --event-stride S instead
        # brackets every S-th launch with its own pair (isolated duration, but each pair costs a queue bubble).
        stride = args.event_stride
        ev = {k: (ctx.event(), ctx.event()) for k in range(0, args.steps, stride)} if stride > 0 else {}
        a, b = ctx.event(), ctx.event()
        hg.barrier()
        ctx.synchronize()
        t0 = time.perf_counter()
        a.record()
        for k in range(args.steps):
            pair = ev.get(k)
            if pair:
                pair[0].record()
            do()
            if pair:
                pair[1].record()
            if fn_after_step:
                fn_after_step()
        b.record()
        ctx.synchronize()
        dt = time.perf_counter() - t0
        hg.barrier()
        wall = hg.max(dt)
        kms = float(np.mean([y.elapsed_ms_since(x) for x, y in ev.values()])) if ev else b.elapsed_ms_since(a) / args.steps
        for x, y in list(ev.values()) + [(a, b)]:
            x.destroy(); y.destroy()
        return wall, kms

    # ---- the timed step: every rank evaluates its own shard; the path has no exchange step, so no collective
    elapsed, kern_ms = timed()
    kern_ms_all = hg.max(kern_ms)

    # ---- multi-GPU only: the RCCL all-gather that reassembles the sharded torque history on every GPU,
    #      measured as a second timed loop (step + all-gather) and reported next to `value`
    allgather = None
    # MANIPULAPY_BENCH_FORCE_GATHER=1 runs the phase on a single GPU too (a one-rank communicator): exercises this code path
    if (world > 1 or os.environ.get("MANIPULAPY_BENCH_FORCE_GATHER") == "1") and not args.no_gather:
        allgather = {"bytes_per_rank": nb, "collective": "ncclAllGather (RCCL), one call per step after the kernel"}

        def verify_gather(d_all):
            """Rank 0 recomputes trajectory 0 of EVERY rank's shard (inputs are seeded per rank) with the same two kernels and
            compares it bit for bit with what the all-gather delivered into that rank's slot."""
            try:
                ctx.synchronize()
                for r in range(world):
                    rr = np.random.default_rng(SEED + cid + 1000 * r)
                    s0 = rr.uniform(lo, hi, (B, n)).astype(np.float32)[:1]
                    e0 = rr.uniform(lo, hi, (B, n)).astype(np.float32)[:1]
                    p, v, a = ctx.batch_trajectory_host(model, s0, e0, 2.0, N, 5)
                    want = ctx.id_trajectory_host(model, p[0], v[0], a[0])
                    got = np.empty((N, n), np.float32)
                    _hip._check(ctx.lib.mp_memcpy_d2h(ctx.handle, got.ctypes.data, d_all.offset(r * nb), got.nbytes))
                    if not np.array_equal(got, want):
                        return f"mismatch in the slot of rank {r}: max abs diff {float(np.abs(got - want).max()):.3e}"
                return True
            except Exception as exc:
                return f"not checked: {str(exc)[:200]}"

        def gather_phase():
            try:
                uid = _hip.HipContext.comm_unique_id() if info.rank == 0 else None
                uid = hg.broadcast_bytes(uid, _hip.UNIQUE_ID_BYTES)
                comm = ctx.comm_create(uid, world, info.rank)
                d_tau_all = ctx.alloc(nb * world)
                wall_g, _ = timed(lambda: comm.allgather(d_tau, d_tau_all, nb))
                ms_g = wall_g / args.steps * 1e3
                allgather.update({"ms_per_step_with_allgather": ms_g,
                                  "value_with_allgather": rows * n * world * args.steps / wall_g,
                                  "busbw_GBps": nb * (world - 1) / max(ms_g - elapsed / args.steps * 1e3, 1e-6) / 1e6})
                if info.rank == 0 and cfg["op"] == "id" and cfg["dtype"] == "f32":
                    allgather["verified"] = verify_gather(d_tau_all)
                if cfg["op"] == "id":
                    # the same reassembly overlapped with compute: the shard is evaluated in 4 chunks straight into this
                    # rank's slot of the gathered buffer; after each chunk's kernel its bytes travel to every peer
                    # (grouped ncclSend / ncclRecv on the communicator's stream) while the next chunk's kernel runs
                    try:
                        ctx.memset(d_tau_all, 0, nb * world)
                        chunks = 4
                        rc = ((rows // chunks) + 1) & ~1  # even, so only the last chunk can end on an odd row
                        row_b = n * wbytes

                        def step_overlapped():
                            for r0 in range(0, rows, rc):
                                nr = min(rc, rows - r0)
                                off = r0 * row_b
                                ctx.id_trajectory(model, d_q.offset(off), d_qd.offset(off), d_qdd.offset(off), nr,
                                                  d_tau_all.offset(info.rank * nb + off), dtype=dt_np)
                                comm.exchange_chunk(d_tau_all, nb, off, nr * row_b)
                            comm.join()

                        wall_o, _ = timed(step_fn=step_overlapped)
                        ms_o = wall_o / args.steps * 1e3
                        allgather["overlapped"] = {"chunks": chunks, "ms_per_step": ms_o,
                                                   "value": rows * n * world * args.steps / wall_o,
                                                   "how": "per chunk: kernel on the compute stream, then grouped ncclSend/ncclRecv "
                                                          "to every peer on the communicator's stream"}
                        if info.rank == 0 and cfg["dtype"] == "f32":
                            allgather["overlapped"]["verified"] = verify_gather(d_tau_all)
                    except Exception as exc:
                        allgather["overlapped"] = {"error": str(exc)[:300]}
                comm.destroy()
            except Exception as exc:  # keep the compute line even if RCCL is unusable on this node
                allgather["error"] = str(exc)[:300]

        # a collective that never returns must not cost the compute line: bounded wait, then report and leave
        import threading

        th = threading.Thread(target=gather_phase, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("MANIPULAPY_BENCH_GATHER_TIMEOUT", "120")))
        if th.is_alive():
            allgather["error"] = "timeout: the RCCL phase did not complete"
            hung = True
        else:
            hung = False
    else:
        hung = False

    jt_per_step = rows * n * world
    value = jt_per_step * args.steps / elapsed
    alg_bytes = algorithmic_bytes_per_row(cfg, n) * rows  # per launch (one rank's kernel)
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9

    result = {
        "metric": "joint-timesteps/sec (NxBxDOF) inverse-dynamics trajectory",
        "value": value, "unit": "joint-timesteps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": cfg["dtype"], "data": "synthetic",
        "config": {"workload": cfg["desc"], "robot": cfg["robot"], "dof": n, "B_per_gpu": B, "N": N,
                   "rows_per_gpu": rows, "op": cfg["op"], "inputs": "q/qd/qdd histories resident in HBM",
                   "kernel_variant": "robot-specialised (hiprtc)" if cfg["specialized"] else "generic",
                   "sharding": f"batch axis over {world} rank(s), no collective in the timed step"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                     "kernel": kernel_name(cfg), "kernel_ms": kern_ms,
                     "kernel_ms_max_over_ranks": kern_ms_all, "algorithmic_bytes_per_launch": alg_bytes,
                     "kernel_ms_method": ("HIP event pair around every %d-th launch of the timed region" % args.event_stride)
                     if args.event_stride > 0 else
                     "HIP events on the launch stream around the K launches of the timed region, / K (launch period: duration + dispatch gap)"},
        "launch": "one hipGraph of K captured launches" if args.launch == "graph" else "one host call per step",
        "setup": {"ramp_ms": args.ramp_ms, "what": "untimed launches before the W warm-up steps (clock ramp)"},
        "device": props["name"],
    }
    traffic_file = os.path.join(ROOT, "profiles", f"traffic_{args.config}.json")
    if os.path.exists(traffic_file):  # measured HBM bytes per launch from the rocprofv3 --pmc passes
        with open(traffic_file) as f:
            tf = json.load(f)
        if tf.get("kernel") == result["roofline"]["kernel"]:  # counters were collected on this very kernel
            result["roofline"]["traffic"] = tf.get("hbm_bytes_per_launch")

    if info.rank == 0 and world == 1 and not args.no_cpu_baseline:
        ns = rows  # the C oracle sizes its own sample from a time budget
        q = d_q.download((rows, n), dt_np)[:ns]
        qd = d_qd.download((rows, n), dt_np)[:ns]
        qdd = d_qdd.download((rows, n), dt_np)[:ns]
        try:
            base, tau_cpu = cpu_baseline(cfg["robot"], q, qd, qdd)
            result["cpu_baseline"] = base
            tau_gpu = d_tau.download((rows, n), dt_np)[: len(tau_cpu)]
            err = np.abs(tau_gpu.astype(np.float64) - tau_cpu)
            result["parity_sample"] = {"rows": int(len(tau_cpu)), "max_abs_err": float(err.max()),
                                       "max_abs_tau": float(np.abs(tau_cpu).max())}
        except Exception as exc:  # the GPU line must not be lost to a host-side problem (no compiler, no OpenMP ...)
            result["cpu_baseline"] = {"value": None, "unit": "joint-timesteps/s", "cores": 0, "kind": "port",
                                      "sample": f"not measured: {str(exc)[:200]}"}
    if allgather is not None:
        result["allgather"] = allgather
    if info.rank == 0:
        emit(result)
    if hung:
        os._exit(0)  # a stuck collective cannot be cancelled from Python
    ctx.destroy()


if __name__ == "__main__":
    main()
