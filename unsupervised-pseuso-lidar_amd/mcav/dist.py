"""Data parallelism: one process per GPU, one RCCL all-reduce of the flat gradient arena per step.

The reference has no distributed code; this is the single exchange step the sharded path needs.  Each rank takes a
disjoint slice of the batch stream, gradients are SUMMED over ranks in one collective on the contiguous arena
(63.7 MB fp32 for ResNet-18 + PoseNet) and scaled by 1/world_size inside the fused Adam kernel, every rank applies
the same update.  BatchNorm uses per-rank batch statistics (as stock DDP without SyncBN).
"""
import os

import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def init_from_env(backend=None):
    """torchrun-style env (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR/PORT).  backend 'nccl' is RCCL on ROCm."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1 or dist.is_initialized():
        return rank(), world()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, device_id=torch.device("cuda", local))      # bind the communicator to this rank's GPU
    else:
        dist.init_process_group(backend=backend)
    return rank(), world()


def broadcast_parameters(arena, src=0):
    if world() > 1:
        dist.broadcast(arena.flat, src)
        arena.bump()


def allreduce_gradients(arena):
    """Sum the gradient arena over ranks (one collective).  Returns the scale Adam must apply (1/world)."""
    w = world()
    if w > 1:
        dist.all_reduce(arena.gflat, op=dist.ReduceOp.SUM)
    return 1.0 / w


def shard_indices(indices, r=None, w=None):
    """Disjoint, equal-sized slices of a (seeded-shuffled) index list; the tail that does not divide is dropped."""
    r = rank() if r is None else r
    w = world() if w is None else w
    per = len(indices) // w
    return indices[r * per:(r + 1) * per]
