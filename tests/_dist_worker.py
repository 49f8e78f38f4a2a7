"""Worker for tests/test_host_logic.py::test_multi_process_gloo_shard_and_gather (launched by torch.distributed.run,
world_size 2 or 3, gloo, CPU only).  Every rank evaluates ITS shard of the torque history - what north_star all-gathers -
through the CPU launchers, the shards (uneven when B % world != 0) are gathered in rank order, and rank 0 compares the
result with the single-process evaluation of the whole batch."""
import sys

import numpy as np

import manipulapy_amd as mp
from manipulapy_amd import sharding


def main(out_path):
    info = sharding.dist_env()
    hg = sharding.HostGather(info)
    sm, dyn, lim = mp.load_robot("ur5")
    pl = mp.OptimizedTrajectoryPlanning(sm, None, dyn, lim, use_cuda=False)
    rng = np.random.default_rng(11)
    B, N = 10, 33
    s = rng.uniform(lim[:, 0], lim[:, 1], (B, 6)).astype(np.float32)
    e = rng.uniform(lim[:, 0], lim[:, 1], (B, 6)).astype(np.float32)
    ls, le, (lo, hi) = sharding.shard_batch(s, e, info.world, info.rank)
    local = pl.batch_joint_trajectory(ls, le, 2.0, N, 5)["positions"]
    assert local.shape == (hi - lo, N, 6)
    gathered = hg.allgather(local)
    # the torque history: inverse_dynamics_trajectory of the shard (generation fused into it), gathered
    tau_local = pl.batch_inverse_dynamics_trajectory(ls, le, 2.0, N, 5)
    assert tau_local.shape == (hi - lo, N, 6) and tau_local.dtype == np.float32
    tau = hg.allgather(tau_local)
    counts, offsets = sharding.shard_layout(B, info.world, N * 6 * 4)
    assert counts[info.rank] == tau_local.nbytes and sum(counts) == tau.nbytes and offsets[info.rank] == lo * N * 6 * 4
    payload = hg.broadcast_bytes(bytes(range(128)) if info.rank == 0 else None, 128)
    assert payload == bytes(range(128))
    mx = hg.max(float(info.rank))
    hg.barrier()
    # bench.py's multi-rank agreements (round 6): one stuck rank makes EVERY rank treat the RCCL phase as stuck, and a strong-scaled
    # entry whose set-up failed on one rank is left by all of them together with an "error" entry
    import argparse
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    assert bench.agree_hung(hg, False, {}) is False
    assert bench.agree_hung(hg, info.rank == info.world - 1, {}) is True

    class _Buf:
        def free(self): pass
        def offset(self, nbytes): return None

    class _FakeCtx:                        # a context whose launches do nothing; its set-up FAILS on the last rank only
        def specialize(self, model):
            if info.rank == info.world - 1:
                raise RuntimeError("set-up failed on this rank")
        def is_specialized(self, model): return False
        def to_device(self, a): return _Buf()
        def alloc(self, nbytes): return _Buf()
        def batch_trajectory(self, *a, **k): pass
        def id_trajectory(self, *a, **k): pass
        def synchronize(self): pass

    entry, hung = bench.bench_strong("c4", argparse.Namespace(no_specialize=False), info, hg, _FakeCtx(), {})
    assert hung is False and "error" in entry and entry["scaling"] == "strong", entry
    if info.rank == info.world - 1:
        assert "set-up failed on this rank" in entry["error"]
    else:                                  # their own set-up went through: they leave because a PEER failed, before any barrier
        assert "another rank failed" in entry["error"], entry
    words = np.arange(1000, dtype=np.uint32) * np.uint32(2654435761)
    assert bench.weighted_word_sum(words) == bench.weighted_word_sum(words.copy()) != bench.weighted_word_sum(words[::-1].copy())
    hg.barrier()
    if info.rank == 0:
        single = pl.batch_joint_trajectory(s, e, 2.0, N, 5)["positions"]
        tau_single = pl.batch_inverse_dynamics_trajectory(s, e, 2.0, N, 5)
        np.savez(out_path, gathered=gathered, single=single, tau=tau, tau_single=tau_single, world=info.world, max_val=mx)


if __name__ == "__main__":
    main(sys.argv[1])
