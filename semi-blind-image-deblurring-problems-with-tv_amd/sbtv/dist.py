"""Multi-GPU layer: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" on CPU-only hosts for tests).

The hot path shards by independent units:
  * images of a batch        -> image i runs on rank i mod world, no data-path collective;
                                one gather of per-image scalars at the end;
  * MYULA chains on ONE image -> chains split over ranks; per SAPG iteration ONE all-reduce(sum)
                                of 5 doubles [sum G_theta, sum G_p0, sum G_p1, sum G_sigma, n_chains]
                                (the reference's `mean(g_*)`, SAPG_algorithm_moffat.m:158-173);
  * a single image is never split across GPUs (global FFT / TV halo exchange): replicas only.
"""
from __future__ import annotations

import ctypes as C
import os


def init(backend=None):
    """Initialise the default process group from the torchrun environment (idempotent)."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        lr = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(lr)
        kw["device_id"] = torch.device(f"cuda:{lr}")
    dist.init_process_group(backend, **kw)
    return dist.get_rank(), dist.get_world_size()


def rank_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard(n_items, rank=None, world=None):
    """Indices of the items this rank owns: item i -> rank i mod world."""
    if rank is None or world is None:
        rank, world = rank_world()
    return list(range(rank, n_items, world))


def split_chains(n_chains, rank=None, world=None):
    """Number of MYULA chains this rank runs and the index of its first chain: pass them to the SAPG call as
    op["chains"] and op["chain_offset"] so that chain b draws the Philox stream chain_offset + b.

    Every rank must own at least one chain: a rank without chains could not make the SAPG call and the others
    would wait for it in the per-iteration all-reduce.  n_chains and world are the same on all ranks, so this check
    raises on all of them together."""
    if rank is None or world is None:
        rank, world = rank_world()
    if n_chains < world:
        raise ValueError(f"shared-gradient SAPG needs at least one chain per rank: {n_chains} chains on {world} ranks "
                         "(use fewer ranks or more chains)")
    base, extra = divmod(n_chains, world)
    mine = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return mine, first


def gather_objects(obj):
    """All ranks' per-item results (small python objects: scalars per image) on every rank."""
    import torch.distributed as dist
    r, w = rank_world()
    if w == 1:
        return [obj]
    out = [None] * w
    dist.all_gather_object(out, obj)
    return out


def merge_sharded(n_items, per_rank_lists):
    """Inverse of `shard`: per_rank_lists[r][k] is the result of item r + k*world."""
    world = len(per_rank_lists)
    out = [None] * n_items
    for r, lst in enumerate(per_rank_lists):
        for k, v in enumerate(lst):
            out[r + k * world] = v
    return out


def make_allreduce_fn():
    """A `reduce_fn` for sbtv_SAPG_algorithm(share_gradients=1): sums `buf[0:n]` (n = 6: four gradient sums, the
    chain count and the failed-rank flag) over all ranks in place.  RCCL needs a device tensor: one persistent
    8-double device buffer and one pinned host buffer are kept for the life of the callback, so an iteration costs
    two 48-byte copies and the collective, no allocation."""
    import torch
    import torch.distributed as dist
    r, w = rank_world()
    if w == 1:
        return None
    use_cuda = dist.get_backend() == "nccl"
    host = torch.zeros(8, dtype=torch.float64)
    if use_cuda:
        host = host.pin_memory()
        dev = torch.zeros(8, dtype=torch.float64, device="cuda")

    def fn(user, buf, n):
        try:
            for i in range(n):
                host[i] = buf[i]
            if use_cuda:
                dev.copy_(host, non_blocking=True)
                dist.all_reduce(dev, op=dist.ReduceOp.SUM)
                host.copy_(dev)                       # synchronising copy: the sums are needed on the host now
            else:
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
            for i in range(n):
                buf[i] = float(host[i])
            return 0
        except Exception:      # never raise through the C boundary
            return 1
    return fn


class _DeviceView:
    """Zero-copy view of `n` doubles at a raw device address (CUDA array interface, read by torch.as_tensor)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def make_device_allreduce_fn(device=None):
    """A `reduce_dev_fn` for sbtv_SAPG_algorithm(share_gradients=1, flags SBTV_REDUCE_DEVICE): sums the library's own
    6-double DEVICE buffer over all ranks in place, in stream order on the library's stream, so the SAPG loop never
    waits for the host: the buffer is wrapped as a tensor once (CUDA array interface), the library's hipStream_t as a
    torch ExternalStream, and `all_reduce` is enqueued under that stream (RCCL orders its own stream after the work
    already enqueued there and makes the stream wait for the collective; no host synchronisation).  With the gloo
    backend (CPU rehearsals) torch stages the tensor through the host, which synchronises - correct, only slower."""
    import torch
    import torch.distributed as dist
    r, w = rank_world()
    if w == 1:
        return None
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
    views, streams = {}, {}

    def fn(user, ptr, n, stream):
        try:
            key = (int(ptr), int(n))
            t = views.get(key)
            if t is None:
                t = views[key] = torch.as_tensor(_DeviceView(ptr, n), device=dev)
            st = streams.get(stream)
            if st is None:
                st = streams[stream] = torch.cuda.ExternalStream(int(stream or 0), device=dev)
            with torch.cuda.stream(st):
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return 0
        except Exception:      # never raise through the C boundary
            return 1
    return fn


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
