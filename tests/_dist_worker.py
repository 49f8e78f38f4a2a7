"""Worker for tests/test_host_logic.py::test_two_process_gloo_shard_and_gather (launched by
torch.distributed.run, world_size 2, gloo, CPU only)."""
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
    payload = hg.broadcast_bytes(bytes(range(128)) if info.rank == 0 else None, 128)
    assert payload == bytes(range(128))
    mx = hg.max(float(info.rank))
    hg.barrier()
    if info.rank == 0:
        single = pl.batch_joint_trajectory(s, e, 2.0, N, 5)["positions"]
        np.savez(out_path, gathered=gathered, single=single, world=info.world, max_val=mx)


if __name__ == "__main__":
    main(sys.argv[1])
