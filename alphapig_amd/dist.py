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
        # Ranks legitimately sit in a collective while rank 0 runs an arena evaluation (lock-step schedule) or drains its
        # trainer thread at the end of a run (asynchronous schedule: pipeline.py); the backend default (10 minutes on
        # NCCL / RCCL) can be too short for that, two hours left the GPUs of a job whose rank 0 had crashed hanging for
        # two hours.  30 minutes; a trainer that dies is announced in the round header, not by a timeout.
        import datetime
        timeout = datetime.timedelta(seconds=float(os.environ.get("APZ_DIST_TIMEOUT_S", "1800")))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
    return rank, world, local


_ACTIVE = False        # a process group was requested: only then do the helpers below touch torch
last_gather_total = 0  # all_gather_tuples: number of tuples all ranks contributed to the last exchange


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


def group_info():
    """What the live process group says about itself (bench.py prints it): backend (nccl == RCCL on ROCm), world size
    as the BACKEND sees it, this rank.  None without a process group."""
    if not _ACTIVE:
        return None
    import torch.distributed as dist
    if not dist.is_initialized():
        return None
    return {"backend": str(dist.get_backend()), "world_size": int(dist.get_world_size()), "rank": int(dist.get_rank())}


def rank_world():
    """(rank, world) of this process: from the live process group, else from the launcher's environment, else (0, 1)."""
    if _ACTIVE:
        import torch.distributed as dist
        if dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def is_active():
    """True once init() has created (or found) a process group with more than one rank, or was forced."""
    return _ACTIVE


def broadcast_params(params, src=0):
    """The weight exchange after a training step (the reference re-syncs its predict modules from the train module,
    policy_value_net_mxnet.py:295-297; across ranks that is one 12.7 MB broadcast): `params` is an ordered
    {name: tensor or ndarray}; rank `src` sends its values, every rank returns {name: tensor} with the SAME names, shapes
    and order holding src's values -- device tensors under RCCL (ready for PolicyValueNet.load_device_params), CPU tensors
    under gloo.  All tensors travel as ONE flat float32 buffer (one collective, not one per tensor)."""
    import torch
    import torch.distributed as dist
    names = list(params.keys())
    if not (_ACTIVE and dist.is_initialized()):
        return {k: (v if torch.is_tensor(v) else torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))) for k, v in params.items()}
    dev = _device()
    parts = []
    for k in names:
        v = params[k]
        t = v if torch.is_tensor(v) else torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
        parts.append(t.detach().to(device=dev, dtype=torch.float32).reshape(-1))
    flat = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.float32, device=dev)
    dist.broadcast(flat, src=src)
    out, off = {}, 0
    for k, t in zip(names, parts):
        n = t.numel()
        v = params[k]
        out[k] = flat[off:off + n].reshape(tuple(v.shape)).contiguous()
        off += n
    return out


def broadcast_floats(values, src=0):
    """A few Python floats from rank `src` to everybody (loss / KL / lr multiplier of a policy update, for the logs)."""
    if not _ACTIVE:
        return [float(v) for v in values]
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=_device())
    dist.broadcast(t, src=src)
    return [float(v) for v in t.cpu()]


last_gather_stats = None   # all_gather_tuples(timing=True): where the time of the last exchange went (bench.py's `exchange`)


def all_gather_tuples(codes, pis, zs, consumer=None, timing=False):
    """codes uint8 [T, S], pis float32 [T, HW], zs float32 [T] of this rank ->
    the concatenation over ranks in rank order (every rank gets everything).
    Variable T per rank: counts are gathered first, payloads are padded to max T.
    consumer = r: only rank r needs the result (the training pipeline's replay buffer lives on rank 0): the other
    ranks take part in the collective but skip the device -> host copy and the unpacking and get empty arrays.
    timing = True: the phases are bracketed by device synchronisations and their wall times land in
    `last_gather_stats` (a measurement mode: the product path never waits for the collective on non-consumers)."""
    import time
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

    def mark():
        if timing and dev.type == "cuda":
            torch.cuda.synchronize(dev)
        return time.perf_counter()
    t_start = mark()
    T, S, HW = codes.shape[0], codes.shape[1], pis.shape[1]
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([T], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, mine)
    counts = counts.cpu().numpy()
    global last_gather_total, last_gather_stats
    last_gather_total = int(counts.sum())            # tuples in the exchange, on every rank (also on non-consumers)
    tmax = int(counts.max())
    if tmax == 0:
        last_gather_stats = None
        return codes[:0], pis[:0], zs[:0]
    t_counts = mark()
    # the arrays the consumer keeps: allocated now, and their pages touched by a helper thread while this thread packs and
    # the device copies / the collective run (first touch of 80 MB of fresh pages is 5 - 12 ms: most of what unpacking cost)
    is_consumer = consumer is None or dist.get_rank() == int(consumer)
    g_codes = g_pis = g_zs = touched = None
    if is_consumer:
        total = int(counts.sum())
        g_pis = np.empty((total, HW), np.float32)
        g_zs = np.empty(total, np.float32)
        g_codes = np.empty((total, S), np.uint8)
        touched = _copy_pool().submit(_touch_pages, (g_pis, g_codes, g_zs))
    # one byte buffer per rank, three contiguous sections padded to tmax rows each: [pi (f32) | z (f32) | codes] -- the
    # float sections first (4-byte aligned whatever S is); packing and unpacking are plain block copies (a row-interleaved
    # layout cost 0.86 s to unpack per 80 MB on the consumer).  The padding is never read (every rank unpacks counts[r]
    # rows), so nothing is zero-filled; on a GPU the rows are packed straight into a reused PINNED staging buffer (round 5:
    # a fresh np.zeros of the whole buffer + a copy from pageable memory were 16 of the call's 36 ms per 80 MB).
    row = S + 4 * HW + 4
    o_z, o_c = tmax * 4 * HW, tmax * (4 * HW + 4)
    nbytes = tmax * row
    cuda = dev.type == "cuda"
    if cuda:
        send_host = _staging("send", nbytes, torch)
        buf = send_host.numpy()
    else:
        buf = np.empty(nbytes, dtype=np.uint8)
    _block_copy(buf[:T * 4 * HW], pis.reshape(-1).view(np.uint8))
    buf[o_z:o_z + 4 * T] = zs.view(np.uint8)
    _block_copy(buf[o_c:o_c + T * S], codes.reshape(-1))
    t_pack = mark()
    if cuda:
        send = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        send.copy_(send_host[:nbytes], non_blocking=True)
    else:
        send = torch.from_numpy(buf)
    recv = torch.empty(world * nbytes, dtype=torch.uint8, device=dev)
    t_h2d = mark()
    dist.all_gather_into_tensor(recv, send)
    t_coll = mark()
    if timing:
        last_gather_stats = {"rows_per_rank": [int(c) for c in counts], "row_bytes": int(row),
                             "bytes_sent_per_rank": int(tmax) * int(row), "bytes_gathered": int(world) * int(tmax) * int(row),
                             "counts_ms": 1e3 * (t_counts - t_start), "pack_ms": 1e3 * (t_pack - t_counts),
                             "h2d_ms": 1e3 * (t_h2d - t_pack), "collective_ms": 1e3 * (t_coll - t_h2d), "d2h_ms": 0.0,
                             "unpack_ms": 0.0, "backend": str(dist.get_backend()), "world_size": int(world),
                             "consumer": None if consumer is None else int(consumer)}
    if not is_consumer:
        if cuda:
            torch.cuda.current_stream(dev).synchronize()      # the staging buffer is reused by the next call
        return codes[:0], pis[:0], zs[:0]
    if cuda:
        recv_host = _staging("recv", world * nbytes, torch)
        recv_host[:world * nbytes].copy_(recv, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        out = recv_host.numpy()[:world * nbytes].reshape(world, nbytes)
    else:
        out = recv.numpy().reshape(world, nbytes)
    t_d2h = time.perf_counter()
    # one copy out of the (reused) staging buffer into the arrays the caller keeps
    touched.result()
    at = 0
    for r in range(world):
        c = int(counts[r])
        if c:
            _block_copy(g_pis[at:at + c].reshape(-1).view(np.uint8), out[r, :c * 4 * HW])
            g_zs[at:at + c].view(np.uint8)[:] = out[r, o_z:o_z + 4 * c]
            _block_copy(g_codes[at:at + c].reshape(-1), out[r, o_c:o_c + c * S])
            at += c
    if timing:
        last_gather_stats["d2h_ms"] = 1e3 * (t_d2h - t_coll)
        last_gather_stats["unpack_ms"] = 1e3 * (time.perf_counter() - t_d2h)
    return g_codes, g_pis, g_zs


_STAGING = {}
_COPY_POOL = None
_COPY_THREADS = int(os.environ.get("APZ_COPY_THREADS", "4"))      # 1: large blocks are copied by the calling thread alone


def _block_copy(dst, src):
    """dst[:] = src for two flat uint8 views, in parallel slices when the block is large: a single thread copies an 80 MB
    round at 7 - 30 GB/s depending on the box (first touch of the fresh destination pages included: 5 - 12 ms of the
    exchange's 14 - 23 ms per 80 MB in round 6); numpy releases the GIL inside the copy, so four threads share the page
    faults and the memcpy."""
    n = dst.shape[0]
    if n < (8 << 20) or _COPY_THREADS <= 1:
        dst[:] = src
        return
    pool = _copy_pool()
    parts = max(1, min(_COPY_THREADS, pool._max_workers))
    step = -(-n // parts)
    step += -step % 4096

    def one(a):
        dst[a:a + step] = src[a:a + step]
    list(pool.map(one, range(0, n, step)))


def _copy_pool():
    global _COPY_POOL
    if _COPY_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        try:
            share = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            share = os.cpu_count() or 1
        _COPY_POOL = ThreadPoolExecutor(max_workers=max(2, min(4, share)), thread_name_prefix="apz-copy")
    return _COPY_POOL


def _touch_pages(arrays):
    """First touch of freshly allocated arrays (one byte per 4 KB page; the arrays' contents are undefined anyway)."""
    for a in arrays:
        flat = a.reshape(-1).view(np.uint8)
        if flat.shape[0]:
            flat[::4096] = 0
            flat[-1] = 0


def _staging(name, nbytes, torch):
    """A pinned host buffer of at least `nbytes` bytes, kept between calls (grown by half when too small)."""
    t = _STAGING.get(name)
    if t is None or t.numel() < nbytes:
        _STAGING[name] = t = torch.empty(int(nbytes * 1.5) + 4096, dtype=torch.uint8).pin_memory()
    return t


def synthetic_round_payload(rows, code_stride=240, hw=225, seed=0):
    """A full round's (codes, pi, z) rows of the shape self-play produces -- position codes 0 ... 8 + the colour byte,
    a normalised pi, z in {-1, 0, 1} -- for measuring the exchange without playing the round (bench.py's `exchange`)."""
    rs = np.random.RandomState(seed)
    codes = rs.randint(0, 9, size=(rows, code_stride)).astype(np.uint8)
    codes[:, hw:] = 0
    codes[:, hw] = rs.randint(0, 2, size=rows)
    pis = rs.rand(rows, hw).astype(np.float32)
    pis /= pis.sum(axis=1, keepdims=True)
    zs = rs.randint(-1, 2, size=rows).astype(np.float32)
    return codes, pis, zs


def measure_exchange(rows, code_stride=240, hw=225, repeats=3, seed=0, consumer=None, both=True):
    """Time `all_gather_tuples` on a synthetic full-round payload of `rows` rows per rank (this rank's rows are seeded
    by its rank): one untimed call, then `repeats` timed ones between barriers; -> dict for the bench line (whole-call
    wall time as the MAX over ranks, the collective alone, bytes, GB/s, every phase of rank 0's call), or None without a
    process group.  The gathered rows are checked against what every rank must have sent (regenerated from the seeds).
    both = True: measured twice -- every rank unpacks everything (consumer=None: the top-level numbers) and the training
    pipeline's mode, only rank 0 unpacks (`pipeline_mode`, consumer=0: pipeline.TrainPipeline._run_async)."""
    import time
    if not _ACTIVE:
        return None
    import torch.distributed as dist
    if not dist.is_initialized():
        return None
    rank, world = dist.get_rank(), dist.get_world_size()
    if rows <= 0:
        return {"what": "no rows to exchange", "rows_per_rank": 0, "ranks_seen": int(round(all_reduce_sum(1))), "payload_verified": True}
    codes, pis, zs = synthetic_round_payload(rows, code_stride, hw, seed + rank)

    def one_mode(cons):
        wall, coll, stats, g = [], [], None, None
        for it in range(repeats + 1):
            barrier()
            t0 = time.perf_counter()
            g = all_gather_tuples(codes, pis, zs, consumer=cons, timing=True)
            dt_local = time.perf_counter() - t0
            dt = all_reduce_max(dt_local)
            if it:
                wall.append(dt)
                coll.append(all_reduce_max(last_gather_stats["collective_ms"]))
                stats = dict(last_gather_stats, whole_call_ms_this_rank=1e3 * dt_local)
        ok = True
        if cons is None or rank == cons:
            ok = g[0].shape[0] == rows * world
            for r in range(world) if ok else ():
                c, p, z = (codes, pis, zs) if r == rank else synthetic_round_payload(rows, code_stride, hw, seed + r)
                sl = slice(r * rows, (r + 1) * rows)
                ok = ok and np.array_equal(g[0][sl], c) and np.array_equal(g[1][sl], p) and np.array_equal(g[2][sl], z)
        ok = all_reduce_sum(0.0 if ok else 1.0) == 0.0
        ms, cms = 1e3 * float(np.median(wall)), float(np.median(coll))
        phases = {k: stats[k] for k in ("counts_ms", "pack_ms", "h2d_ms", "collective_ms", "d2h_ms", "unpack_ms")}
        phases["unattributed_ms"] = stats["whole_call_ms_this_rank"] - sum(phases.values())
        return {"ms": ms, "collective_ms": cms,
                "GB_per_s_collective": stats["bytes_gathered"] / (cms * 1e-3) / 1e9 if cms > 0 else None,
                "GB_per_s_whole_call": stats["bytes_gathered"] / (ms * 1e-3) / 1e9 if ms > 0 else None,
                "phases_ms_rank0": phases, "payload_verified": bool(ok)}, stats
    top, stats = one_mode(consumer)
    res = {"what": "one dist.all_gather_tuples of a full round's (codes | pi | z) rows per rank, synthetic rows, outside the "
                   "timed region; every rank's rows checked on arrival",
           "rows_per_rank": int(rows), "row_bytes": stats["row_bytes"], "bytes_sent_per_rank": stats["bytes_sent_per_rank"],
           "bytes_gathered_per_rank": stats["bytes_gathered"], "consumer": consumer}
    res.update(top)
    if both and consumer is None:
        res["pipeline_mode"] = dict(one_mode(0)[0], consumer=0,
                                    what="the training pipeline's call: only rank 0 copies the gathered rows to the host and unpacks them")
        res["payload_verified"] = bool(res["payload_verified"] and res["pipeline_mode"]["payload_verified"])
    res.update(backend=stats["backend"], ranks_seen=int(round(all_reduce_sum(1))), repeats=int(repeats))
    return res
