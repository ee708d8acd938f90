"""Multi-GPU sharding: one process per GPU, games sharded, ONE exchange of finished tuples.

Self-play games never interact and the weights are read-only during a round, so rank r of R
simply owns the global game indices {r, r+R, r+2R, ...} (its own evaluator handle, its own RNG
streams base_seed + index, a full 12.7 MB weight replica).  The only collective is the
all-gather of finished (state, pi, z) tuples at the end of a round (the reference has no
collectives at all -- SURVEY.md F8 / section 8e): an all_gather of per-rank tuple counts, then
one all_gather_into_tensor of the padded, bit-packed buffers (states travel as the 1-byte-per-
cell position codes, ~30x smaller than float32 planes, and are expanded by the consumer).

Backend "nccl" IS RCCL on ROCm (xGMI inside a node); "gloo" is used by the CPU tests.
"""
import os

import numpy as np


def init(backend=None, device_index=None, force=False):
    """Initialise torch.distributed from the torchrun environment.  -> (rank, world, local_rank)
    force=True creates the process group even for a single rank (exercises RCCL on a 1-GPU box)."""
    global _ACTIVE
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world <= 1 and not force:
        return rank, world, local            # single process: torch is never imported
    import torch
    import torch.distributed as dist
    _ACTIVE = True
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local if device_index is None else device_index)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


_ACTIVE = False        # a process group was requested: only then do the helpers below touch torch


def shard_indices(total_games, rank, world):
    """Global game indices owned by `rank` (round-robin, so every prefix of the global order is
    balanced)."""
    return list(range(rank, total_games, world))


def _device():
    import torch
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def barrier():
    if not _ACTIVE:
        return
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()


def shutdown():
    """Leave the process group in an orderly way (barrier, then destroy): a rank that simply returns while its
    peers are still inside the last collective can take the backend's worker threads down mid-flight
    ("terminate called without an active exception" from gloo on a loaded host)."""
    global _ACTIVE
    if not _ACTIVE:
        return
    import torch.distributed as dist
    if dist.is_initialized():
        try:
            dist.barrier()
        finally:
            dist.destroy_process_group()
    _ACTIVE = False


def all_reduce_max(x):
    if not _ACTIVE:
        return float(x)
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_reduce_sum(x):
    if not _ACTIVE:
        return float(x)
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=_device())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def all_gather_floats(x):
    """One float per rank -> the list over ranks, in rank order (per-rank rates of a measurement)."""
    if not _ACTIVE:
        return [float(x)]
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return [float(x)]
    dev = _device()
    out = torch.zeros(dist.get_world_size(), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, torch.tensor([float(x)], dtype=torch.float64, device=dev))
    return [float(v) for v in out.cpu()]


def all_gather_tuples(codes, pis, zs):
    """codes uint8 [T, S], pis float32 [T, HW], zs float32 [T] of this rank ->
    the concatenation over ranks in rank order (every rank gets everything).
    Variable T per rank: counts are gathered first, payloads are padded to max T."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    pis = np.ascontiguousarray(pis, dtype=np.float32)
    zs = np.ascontiguousarray(zs, dtype=np.float32).reshape(-1)
    if not _ACTIVE:
        return codes, pis, zs
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return codes, pis, zs
    world = dist.get_world_size()
    dev = _device()
    T, S, HW = codes.shape[0], codes.shape[1], pis.shape[1]
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([T], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, mine)
    counts = counts.cpu().numpy()
    tmax = int(counts.max())
    if tmax == 0:
        return codes[:0], pis[:0], zs[:0]
    # one byte buffer per tuple: [codes | pi (f32) | z (f32)]
    row = S + 4 * HW + 4
    buf = np.zeros((tmax, row), dtype=np.uint8)
    buf[:T, :S] = codes
    buf[:T, S:S + 4 * HW] = pis.view(np.uint8).reshape(T, 4 * HW)
    buf[:T, S + 4 * HW:] = zs.view(np.uint8).reshape(T, 4)
    send = torch.from_numpy(buf).to(dev)
    recv = torch.empty((world * tmax, row), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send)
    out = recv.cpu().numpy().reshape(world, tmax, row)
    keep = np.concatenate([out[r, :counts[r]] for r in range(world)])
    g_codes = np.ascontiguousarray(keep[:, :S])
    g_pis = np.ascontiguousarray(keep[:, S:S + 4 * HW]).view(np.float32).reshape(-1, HW)
    g_zs = np.ascontiguousarray(keep[:, S + 4 * HW:]).view(np.float32).reshape(-1)
    return g_codes, g_pis, g_zs
