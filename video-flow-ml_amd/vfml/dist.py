"""Multi-GPU sharding of flow-field work items (one process per GPU, torch.distributed).

Every flow field depends only on its own T-frame window (no warm start: the reference never passes
flow_init, processing/videoflow_core.py:188), and in --tile mode every tile only on its own crop
(processing/videoflow_processor.py:258-277).  So the job is a list of independent work items
(frame i) or (frame i, tile j); ranks take contiguous blocks of that list, compute them with no
data-path communication, and one gather collects the finished fields on rank 0 — RCCL over xGMI on
GPUs (backend "nccl" is RCCL on ROCm), gloo on CPU for the tests.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when not launched by it."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def init_distributed(backend=None):
    """Initialise the default process group if WORLD_SIZE > 1. Returns (rank, local_rank, world)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        use_gpu = torch.cuda.is_available()
        if backend is None:
            backend = "nccl" if use_gpu else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def host_cpu_share(cap=None):
    """CPUs this process may actually use: scheduler affinity and the cgroup-v2 quota (os.cpu_count() reports the whole
    host, and a 16-CPU share oversubscribed with 100+ threads is far slower than 16 threads)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap) if cap else n)


def shard_bounds(num_items, rank, world):
    """Contiguous, balanced block [lo, hi) of `num_items` for `rank` (first num_items % world ranks
    get one extra item). Contiguity keeps a rank's sliding frame windows overlapping."""
    base, extra = divmod(num_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def work_items(frame_indices, tiles_per_frame=1, tile_major=False):
    """Flattened (frame, tile) list.  Frame-major is the order the reference's serial loops visit them
    (flow_processor.py:959 over frames, videoflow_processor.py:258 over tiles); tile-major (all frames
    of tile 0, then tile 1, ...) computes the same items but lets consecutive items share T-1 frames
    of the same crop, which is what the engine's sliding-window caches key on."""
    frame_indices = list(frame_indices)
    if tile_major:
        return [(f, t) for t in range(tiles_per_frame) for f in frame_indices]
    return [(f, t) for f in frame_indices for t in range(tiles_per_frame)]


def gather_to_rank0(local_flat, sizes, group=None):
    """Gather one flat float32 buffer per rank onto rank 0.

    local_flat : 1-D float32 tensor holding this rank's finished fields back to back
    sizes      : list of per-rank element counts (every rank can compute all of them: the item
                 list and field shapes are deterministic), len == world
    Returns on rank 0 a list of `world` 1-D tensors (views trimmed to `sizes`); None elsewhere.
    One collective: ranks pad to the largest shard so that dist.gather applies."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if len(sizes) != world:
        raise ValueError(f"sizes has {len(sizes)} entries for world size {world}")
    if local_flat.dim() != 1 or local_flat.numel() != sizes[rank]:
        raise ValueError(f"rank {rank}: buffer has {local_flat.numel()} elements, expected {sizes[rank]}")
    if world == 1:
        return [local_flat]
    cap = max(sizes)
    send = local_flat
    if send.numel() < cap:
        send = torch.zeros(cap, dtype=local_flat.dtype, device=local_flat.device)
        send[:local_flat.numel()] = local_flat
    recv = [torch.empty(cap, dtype=local_flat.dtype, device=local_flat.device) for _ in range(world)] \
        if rank == 0 else None
    dist.gather(send, recv, dst=0, group=group)
    if rank != 0:
        return None
    return [r[:n] for r, n in zip(recv, sizes)]


def max_over_ranks(value, device):
    """MAX-reduce a Python float over all ranks (timing)."""
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(device=None):
    if dist.is_initialized():
        if device is not None and device.type == "cuda":
            dist.barrier(device_ids=[device.index if device.index is not None else torch.cuda.current_device()])
        else:
            dist.barrier()
