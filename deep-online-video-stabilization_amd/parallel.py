"""Data-parallel plumbing (new: the reference is single-device).  Pure host logic on top of
torch.distributed -- backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.

  training : samples (siamese pairs) are independent given the weights -> shard the global batch by sample, local BN
             statistics, ONE exchange per step: sum of the flat gradient buffer, in a few large buckets (xGMI is
             point-to-point, rings are per-link bound: few big messages, not many small ones).
  inference: replicas only -- frames of a clip are serially dependent through the fed-back history
             (deploy_bundle.py:322-323); a clip/stream lives on one GPU and there is no collective."""
from __future__ import annotations

import os

import torch


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend: str = None, device: torch.device = None, single_rank_group: bool = False):
    """single_rank_group: build the group even for world_size 1 (a one-rank RCCL communicator: how the collective code
    path is run on a one-GPU box)."""
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    if world == 1 and not single_rank_group:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist.group.WORLD


def shard_range(n_global: int, rank: int, world: int):
    """Contiguous sample range of `rank`: the first n_global % world ranks get one extra sample."""
    base, rem = divmod(n_global, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(batch: dict, rank: int, world: int) -> dict:
    """Slice every per-sample array of a global batch dict along dim 0."""
    n = next(iter(batch.values())).shape[0]
    lo, hi = shard_range(n, rank, world)
    return {k: v[lo:hi] for k, v in batch.items()}


def bucket_bounds(n: int, n_buckets: int):
    per = (n + n_buckets - 1) // n_buckets
    return [(b * per, min(n, (b + 1) * per)) for b in range(n_buckets) if b * per < n]


def allreduce_sum_(flat: torch.Tensor, group=None, n_buckets: int = 4):
    """In-place sum of a flat buffer over the group, bucketed; returns the async work handles' completion."""
    import torch.distributed as dist
    works = [dist.all_reduce(flat[lo:hi], group=group, async_op=True) for lo, hi in bucket_bounds(flat.numel(), n_buckets)]
    for w in works:
        w.wait()
    return flat
