"""Multi-GPU sharding of trajectory batches: one process per GPU, no exchange during compute.

The reference is single-device (SURVEY §2.1 "Parallelism strategies": none).  Every (trajectory,
timestep) row is independent, so the batch axis B is cut into `world` contiguous shards; each rank
runs the same kernels on its shard, and the only collective is the all-gather that reassembles the
torque history — RCCL over xGMI on the device (`_hip.HipComm`), gloo on the host for CPU tests.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

__all__ = ["ShardInfo", "shard_range", "shard_batch", "shard_layout", "dist_env", "HostGather"]


@dataclass(frozen=True)
class ShardInfo:
    rank: int
    world: int
    local_rank: int


def dist_env() -> ShardInfo:
    """RANK / WORLD_SIZE / LOCAL_RANK as set by torch.distributed.run (defaults: single process)."""
    return ShardInfo(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
                     int(os.environ.get("LOCAL_RANK", "0")))


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of `total` items for `rank`; the first `total % world` ranks get one extra."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} / world {world}")
    if total < 0:
        raise ValueError("negative item count")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_layout(total: int, world: int, bytes_per_item: int):
    """(byte counts, byte offsets) of every rank's shard of `total` items in the gathered buffer, rank order - what
    HipComm.allgatherv / exchange_chunk_v take when total % world != 0."""
    counts = [(hi - lo) * bytes_per_item for lo, hi in (shard_range(total, world, r) for r in range(world))]
    offsets = [sum(counts[:r]) for r in range(world)]
    return counts, offsets


def shard_batch(start_batch: np.ndarray, end_batch: np.ndarray, world: int, rank: int):
    """This rank's rows of the (B, n) start / end arrays."""
    lo, hi = shard_range(start_batch.shape[0], world, rank)
    return start_batch[lo:hi], end_batch[lo:hi], (lo, hi)


class HostGather:
    """All-gather of equally-shaped host arrays over torch.distributed (gloo).  Host-logic twin of the
    RCCL path, used by the CPU tests and by callers that only need the result on the host."""

    def __init__(self, info: Optional[ShardInfo] = None):
        import torch.distributed as dist  # plumbing only; never touches the GPU

        self.dist = dist
        self.info = info or dist_env()
        if self.info.world > 1 and not dist.is_initialized():
            # a peer that died must surface as an error within minutes, not after gloo's default half hour (bench.py still has a
            # line to print); long enough for rank 0's host legs (the other ranks wait in a barrier meanwhile)
            import datetime

            limit = datetime.timedelta(seconds=float(os.environ.get("MANIPULAPY_GLOO_TIMEOUT_S", "600")))
            dist.init_process_group("gloo", rank=self.info.rank, world_size=self.info.world, timeout=limit)

    def allgather(self, local: np.ndarray) -> np.ndarray:
        """(sum of the ranks' rows, ...) in rank order.  The shards may differ in their FIRST dimension (shard_range hands the
        first B % world ranks one trajectory more): the row counts are exchanged first, every shard travels padded to the
        largest and is trimmed on arrival."""
        if self.info.world == 1:
            return local.copy()
        import torch

        local = np.ascontiguousarray(local)
        counts = torch.zeros(self.info.world, dtype=torch.int64)
        mine = torch.tensor([local.shape[0]], dtype=torch.int64)
        parts = [torch.zeros(1, dtype=torch.int64) for _ in range(self.info.world)]
        self.dist.all_gather(parts, mine)
        counts = [int(p.item()) for p in parts]
        top = max(counts)
        padded = np.zeros((top,) + local.shape[1:], dtype=local.dtype)
        padded[: local.shape[0]] = local
        t = torch.from_numpy(padded)
        outs = [torch.empty_like(t) for _ in range(self.info.world)]
        self.dist.all_gather(outs, t)
        return np.concatenate([o.numpy()[:c] for o, c in zip(outs, counts)], axis=0)

    def barrier(self) -> None:
        if self.info.world > 1:
            self.dist.barrier()

    def max(self, value: float) -> float:
        if self.info.world == 1:
            return float(value)
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def broadcast_bytes(self, payload: Optional[bytes], nbytes: int, src: int = 0) -> bytes:
        """Out-of-band broadcast (the RCCL unique id travels this way)."""
        if self.info.world == 1:
            return payload or b""
        import torch

        t = torch.zeros(nbytes, dtype=torch.uint8)
        if self.info.rank == src:
            t = torch.tensor(list(payload), dtype=torch.uint8)
        self.dist.broadcast(t, src=src)
        return bytes(t.tolist())
