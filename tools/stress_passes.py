#!/usr/bin/env python3
"""Stress run of the parked float64 passes (a bounded run of it is in the GPU suite: tests/test_gpu_parity.py::
test_parked_pass_stress_against_a_context_that_never_parks): random sequences of device-pointer float32 inverse-dynamics launches whose float64
passes the context parks (csrc/mp_capi.cpp, hard_defer / hard_flush), on overlapping sub-ranges of shared device arrays - outputs
landing in other launches' inputs and outputs, three models (two specialised programs and a generic one), interleaved with
uploads, memsets, float64 launches and downloads.  A host mirror of every device array is advanced with what a SECOND context's
host entry point returns for the same rows (it flushes at once); every download must equal its mirror bit for bit.

    SEED=3 OPS=400 [THREADS=3] [FOREIGN=1] python tools/stress_passes.py      (THREADS: the models are dealt out to threads sharing both contexts)

FOREIGN=1 (round 5): the arrays of the first and the second model are the CALLER's (hipMalloc through the HIP runtime, not mp_malloc) and
are read, written and cleared with raw HIP calls on the context's compute stream only - no mp_* entry point between a launch and
the read of its result.  Launches on such arrays get their float64 pass at once (csrc/mp_capi.cpp, hard_park_or_run); and since
round 6 the stream having been handed out (mp_ctx_get_stream) switches parking off for the pool arrays of the third model too.
Without FOREIGN the stream is never asked for: everything stays parked as long as the library allows.

Round 6 added two operations: "recycle" (a launch on freshly mp_malloc'd arrays that are freed while its pass is parked; the same
sizes are allocated again at once - the pool hands the very blocks back - filled with a pattern and used by another launch: the
pattern and that launch's result must come back untouched) and "graph" (two or three launches captured into a launch graph,
replayed once or twice, the mirror advanced per replay).
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import manipulapy_amd as mp  # noqa: E402
from manipulapy_amd import _hip  # noqa: E402

R = 24000  # rows per array


class Foreign:
    """A device array the library does not own, used through raw HIP calls ordered on the context's compute stream."""
    rt = None

    def __init__(self, stream, host=None, nbytes=0, base=None, off=0):
        if Foreign.rt is None:
            rt = ctypes.CDLL("libamdhip64.so")
            rt.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
            rt.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
            rt.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
            rt.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
            rt.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
            rt.hipFree.argtypes = [ctypes.c_void_p]
            Foreign.rt = rt
        self.stream = stream
        if base is not None:
            self.base, self.off, self.ptr = base, off, ctypes.c_void_p(base.ptr.value + off)
            return
        self.base, self.off = None, 0
        nbytes = host.nbytes if host is not None else nbytes
        self.ptr = ctypes.c_void_p()
        assert Foreign.rt.hipMalloc(ctypes.byref(self.ptr), nbytes) == 0
        if host is not None:
            self.upload(host)

    def offset(self, nbytes):
        return Foreign(self.stream, base=self, off=nbytes).ptr

    def upload(self, a):   # the caller orders its own copy: behind everything the stream holds, blocking
        a = np.ascontiguousarray(a)
        assert Foreign.rt.hipStreamSynchronize(ctypes.c_void_p(self.stream)) == 0
        assert Foreign.rt.hipMemcpy(self.ptr, a.ctypes.data_as(ctypes.c_void_p), a.nbytes, 1) == 0

    def download(self, shape, dtype):
        out = np.empty(shape, dtype)
        assert Foreign.rt.hipMemcpyAsync(out.ctypes.data_as(ctypes.c_void_p), self.ptr, out.nbytes, 2, ctypes.c_void_p(self.stream)) == 0
        assert Foreign.rt.hipStreamSynchronize(ctypes.c_void_p(self.stream)) == 0
        return out

    def memset(self, off, nbytes):
        assert Foreign.rt.hipMemsetAsync(ctypes.c_void_p(self.ptr.value + off), 0, nbytes, ctypes.c_void_p(self.stream)) == 0

    def free(self):
        Foreign.rt.hipStreamSynchronize(ctypes.c_void_p(self.stream))
        Foreign.rt.hipFree(self.ptr)


def fast_rows(rng, lim, n, rows):
    """rows of quintic point-to-point trajectories at bench speed: ~1 % of them ill-conditioned in float32"""
    out = []
    N = 600
    for _ in range(-(-rows // N)):
        a, b = rng.uniform(lim[:, 0], lim[:, 1], (2, n))
        t = np.linspace(0.0, 1.0, N)[:, None]
        s, sd, sdd = 10 * t**3 - 15 * t**4 + 6 * t**5, (30 * t**2 - 60 * t**3 + 30 * t**4) / 2.0, (60 * t - 180 * t**2 + 120 * t**3) / 4.0
        out.append(np.stack([a + s * (b - a), sd * (b - a), sdd * (b - a)]))
    o = np.concatenate(out, axis=1)[:, :rows].astype(np.float32)
    return [np.ascontiguousarray(o[k]) for k in range(3)]


def run(seed=0, ops=300, foreign=False, nthreads=1, verbose=True):
    """-> (summary dict, list of mismatches)."""
    rng = np.random.default_rng(seed)
    ctx, ref_ctx = _hip.HipContext(0), _hip.HipContext(0)
    models = []
    stream = ctx.stream() if foreign else 0   # (asking for the stream switches parking off: only the FOREIGN mode does)
    for robot, spec in (("ur5", True), ("xarm6", False), ("panda", True)):
        t = mp.robot_tables(robot)
        m = _hip.HipModel(t["S_list"], t["Mlist_per_link"], t["Glist"], t["M_ee"], t["joint_limits"])
        if spec:
            ctx.specialize(m); ref_ctx.specialize(m)
        lim = np.asarray(t["joint_limits"], dtype=np.float64)
        host = fast_rows(rng, lim, m.n, R) + [np.zeros((R, m.n), np.float32) for _ in range(3)]   # q, qd, qdd, tau0..2
        dev = [Foreign(stream, host=a) for a in host] if foreign and len(models) < 2 else [ctx.to_device(a) for a in host]
        flagged = int(_hip.cpu_id_row_precision(m, *host[:3]).sum())
        models.append({"name": robot, "m": m, "n": m.n, "host": host, "dev": dev, "flagged": flagged})
    launches = checks = recycled = replays = 0
    bad, held, graphs = [], [], []

    def expected(M, rows_in, start, length):
        sl = slice(start, start + length)
        return ref_ctx.id_trajectory_host(M["m"], *(np.ascontiguousarray(M["host"][k][sl]) for k in rows_in), dtype=np.float32)

    def check(M, k, what):
        nonlocal checks
        got = M["dev"][k].download((R, M["n"]), np.float32)
        checks += 1
        if not np.array_equal(got, M["host"][k], equal_nan=True):
            diff = np.flatnonzero((got != M["host"][k]).any(axis=1))
            bad.append((what, M["name"], k, len(diff), int(diff[0]), int(diff[-1])))
            M["host"][k][:] = got   # carry on from what the device holds

    import threading
    lock = threading.Lock()
    nonlocal_counts = {"recycled": 0, "replays": 0}

    def worker(my_models, rng):
      nonlocal launches
      for op in range(ops):
          M = my_models[int(rng.integers(len(my_models)))]
          n, rb = M["n"], M["n"] * 4
          kind = rng.choice(["launch"] * 10 + ["launch_into_input"] * 2 + ["fused"] * 4 + ["download", "memset", "upload", "f64", "sync"]
                            + ["recycle"] * 2 + (["graph"] * 2 if nthreads <= 1 else []))
          if kind in ("launch", "launch_into_input"):
              length = int(rng.choice([1, 63, 64, 65, 200, 1000, int(rng.integers(1, 6000))]))
              s_in, s_out = (2 * int(rng.integers(0, (R - length) // 2 + 1)) for _ in range(2))   # (device pointers: 16-byte aligned)
              dst = int(rng.integers(0, 3)) if kind == "launch_into_input" else int(rng.integers(3, 6))
              rows_in = (0, 1, 2)
              if dst < 3 and not (s_out + length <= s_in or s_in + length <= s_out):
                  continue    # a launch whose output overlaps its OWN input rows is undefined for any kernel
              want = expected(M, rows_in, s_in, length)
              ctx.id_trajectory(M["m"], *(M["dev"][k].offset(s_in * rb) for k in rows_in), length, M["dev"][dst].offset(s_out * rb), dtype=np.float32)
              M["host"][dst][s_out:s_out + length] = want
              with lock:
                launches += 1
          elif kind == "recycle":
              # a launch on arrays of its own from the pool, freed while its float64 pass is (maybe) parked; the same sizes again at
              # once - the pool's exact-size free lists hand those very blocks back - a pattern in each, another launch on them
              nonlocal_counts["recycled"] += 1
              length = int(rng.choice([64, 640, 2000]))
              nb = length * rb
              s1, s2 = (int(rng.integers(0, R - length + 1)) for _ in range(2))
              first = [ctx.to_device(np.ascontiguousarray(M["host"][k][s1:s1 + length])) for k in range(3)] + [ctx.alloc(nb)]
              ctx.id_trajectory(M["m"], *first[:3], length, first[3], dtype=np.float32)
              for b_ in first:
                  b_.free()
              again = [ctx.alloc(nb) for _ in range(4)]
              ins = [np.ascontiguousarray(M["host"][k][s2:s2 + length]) for k in range(3)]
              pattern = rng.uniform(-1, 1, (length, n)).astype(np.float32)
              for b_, a in zip(again, ins + [pattern]):
                  b_.upload(a)
              got_pat = again[3].download((length, n), np.float32)
              ctx.id_trajectory(M["m"], *again[:3], length, again[3], dtype=np.float32)
              got = again[3].download((length, n), np.float32)
              got_in = [b_.download((length, n), np.float32) for b_ in again[:3]]
              want = ref_ctx.id_trajectory_host(M["m"], *ins, dtype=np.float32)
              with lock:
                  launches += 2
              ok3 = (np.array_equal(got_pat, pattern), np.array_equal(got, want, equal_nan=True), all(np.array_equal(x, y, equal_nan=True) for x, y in zip(got_in, ins)))   # (launches that land in input arrays feed torques back: rows do overflow)
              if not all(ok3):
                  rows_bad = np.flatnonzero(((got != want) & ~(np.isnan(got) & np.isnan(want))).any(axis=1))
                  bad.append((f"op {op} recycle", M["name"], -1, len(rows_bad), s1, s2, {"length": length, "pattern_ok": ok3[0], "result_ok": ok3[1], "inputs_ok": ok3[2],
                              "max_abs_diff": float(np.nanmax(np.abs(got - want))) if len(rows_bad) else 0.0, "first_bad_rows": rows_bad[:8].tolist(),
                              "flagged_in_slice": int(_hip.cpu_id_row_precision(M["m"], *ins).sum()),
                              "bad_rows_flagged": int(_hip.cpu_id_row_precision(M["m"], *ins)[rows_bad].sum()) if len(rows_bad) else 0}))
              for b_ in again:
                  b_.free()
          elif kind == "graph":
              # two or three given-rows launches captured into ONE launch graph (their passes are nodes of the graph, lists and counters
              # the graph's own), replayed once or twice; the mirror is advanced launch by launch at every replay
              plan = []
              for _ in range(int(rng.integers(2, 4))):
                  length = int(rng.choice([64, 65, 200, 1000, int(rng.integers(1, 3000))]))
                  s_in, s_out = (2 * int(rng.integers(0, (R - length) // 2 + 1)) for _ in range(2))
                  plan.append((s_in, s_out, int(rng.integers(3, 6)), length))
              with ctx.capture() as cap:
                  for s_in, s_out, dst, length in plan:
                      ctx.id_trajectory(M["m"], *(M["dev"][k].offset(s_in * rb) for k in range(3)), length, M["dev"][dst].offset(s_out * rb), dtype=np.float32)
              graphs.append(cap.graph)
              for _ in range(int(rng.integers(1, 3))):
                  cap.graph.launch()
                  nonlocal_counts["replays"] += 1
                  for s_in, s_out, dst, length in plan:
                      M["host"][dst][s_out:s_out + length] = expected(M, (0, 1, 2), s_in, length)
                  with lock:
                      launches += len(plan)
          elif kind == "download":
              check(M, int(rng.integers(0, 6)), f"op {op}")
          elif kind == "memset":
              k = int(rng.integers(3, 6))
              a, b = sorted(int(x) for x in rng.integers(0, R + 1, 2))
              if b > a:
                  if isinstance(M["dev"][k], Foreign):
                      M["dev"][k].memset(a * rb, (b - a) * rb)
                  else:
                      ctx.memset(M["dev"][k].offset(a * rb), 0, (b - a) * rb)
                  M["host"][k][a:b] = 0
          elif kind == "upload":
              k = int(rng.integers(0, 6))
              fresh = (M["host"][k] * np.float32(0.5)).astype(np.float32) if k < 3 else rng.uniform(-1, 1, (R, n)).astype(np.float32)
              M["dev"][k].upload(fresh)
              M["host"][k][:] = fresh
          elif kind == "f64":    # another entry point altogether: parked passes run first
              rows = 500
              q64 = [ctx.to_device(M["host"][k][:rows].astype(np.float64)) for k in range(3)]
              out = ctx.alloc(rows * n * 8)
              ctx.id_trajectory(M["m"], *q64, rows, out, dtype=np.float64)
              out.download((rows, n), np.float64)
              for b_ in q64 + [out]:
                  b_.free()
          elif kind == "fused":
              B, N = 7, int(rng.choice([300, 300, 300, 257]))     # (a new N or Tf rewrites the time table: parked passes run first)
              Tf = float(rng.choice([2.0, 2.0, 2.0, 1.5]))
              lim = M["m"].joint_limits_f32()
              st, en = rng.uniform(lim[:, 0], lim[:, 1], (2, B, n)).astype(np.float32)
              k = int(rng.integers(3, 6))
              s_out = 2 * int(rng.integers(0, (R - B * N) // 2 + 1))
              want = ref_ctx.traj_id_fused_host(M["m"], st, en, Tf, N, 5).reshape(-1, n)
              ds, de = ctx.to_device(st), ctx.to_device(en)
              ctx.traj_id_fused(M["m"], ds, de, B, N, Tf, 5, M["dev"][k].offset(s_out * rb))
              M["host"][k][s_out:s_out + B * N] = want
              held.append((ds, de))       # the launch's float64 pass is parked and re-reads the end points: freed at the end
          else:
              ctx.synchronize()
    only = os.environ.get("ONLY")
    if nthreads <= 1:
        worker(models, rng)
    elif only is not None:   # one of the threads' workloads alone (same models, same generator): is a mismatch a matter of concurrency?
        worker(models[int(only)::nthreads], np.random.default_rng(seed * 100 + int(only)))
    else:
        ths = [threading.Thread(target=worker, args=(models[i::nthreads], np.random.default_rng(seed * 100 + i))) for i in range(nthreads)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
    for M in models:
        for k in range(6):
            check(M, k, "end")
    summary = {"seed": seed, "ops": ops, "foreign": bool(foreign), "threads": nthreads, "float32_launches": launches, "downloads": checks,
               "recycles": nonlocal_counts["recycled"], "graph_replays": nonlocal_counts["replays"],
               "flagged_rows_per_model": [M["flagged"] for M in models], "mismatches": len(bad)}
    if verbose:
        print(f"seed {seed}{' FOREIGN' if foreign else ''}: {ops} ops, {launches} float32 launches, {checks} downloads, {summary['recycles']} recycles, "
              f"{summary['graph_replays']} graph replays, flagged rows per model {summary['flagged_rows_per_model']}, mismatches {len(bad)}")
        for b_ in bad[:10]:
            print("  MISMATCH", b_)
    for pair in held:
        for b_ in pair:
            b_.free()
    ctx.synchronize()
    for gr in graphs:
        gr.destroy()
    for M in models:
        for d in M["dev"]:
            if isinstance(d, Foreign):
                d.free()
    ctx.destroy(); ref_ctx.destroy()
    return summary, bad


def main():
    _, bad = run(int(os.environ.get("SEED", "0")), int(os.environ.get("OPS", "300")), os.environ.get("FOREIGN", "0") == "1",
                 int(os.environ.get("THREADS", "1")))   # THREADS > 1: the models are dealt out to threads that share BOTH contexts
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
