"""Sharded flow-field job: the loop of reference flow_processor.py:959-976 / :1460-1470
(`for i in frames: compute_optical_flow[_tiled]`) spread over the GPUs of one node.

Work items are (frame) or (frame, tile); ranks take contiguous blocks (vfml.dist.shard_bounds), keep
the clip resident in their own HBM, compute with no data-path communication, and one gather brings
the finished fields to rank 0, which pastes tiles with the reference's hard seams
(processing/videoflow_processor.py:277) and owns the cache files."""
import numpy as np
import torch

from . import dist as vdist


def _tiles(proc, width, height, tile_mode):
    return proc.calculate_tile_grid(width, height)[4] if tile_mode else [None]


def item_numel(height, width, tile):
    h, w = (height, width) if tile is None else (tile['height'], tile['width'])
    return h * w * 2


def _run_streaming(proc, clip, frame_indices, on_field):
    """One rank, whole frames: every field goes to host memory as soon as it is finished (pinned staging
    ring, the copy of field i overlaps the computation of field i+1) and is handed to `on_field(k, array)`
    - the cache writer of flow_processor.py - while the GPU keeps computing."""
    H, W = clip.shape[1], clip.shape[2]
    out = np.zeros((len(frame_indices), H, W, 2), dtype=np.float32)
    on_gpu = clip.is_cuda
    ring = [torch.empty((H, W, 2), dtype=torch.float32).pin_memory() for _ in range(3)] if on_gpu else None
    events = [None] * 3
    pending = []                                       # (slot in out, ring index)

    def drain(limit):
        while len(pending) > limit:
            k, r = pending.pop(0)
            events[r].synchronize()
            out[k] = ring[r].numpy()
            on_field(k, out[k])

    for k, flow in _fields_in_order(proc, clip, frame_indices):
        if not on_gpu:
            out[k] = flow.numpy()
            on_field(k, out[k])
            continue
        drain(2)                                       # the ring slot about to be reused is free
        r = k % 3
        ring[r].copy_(flow, non_blocking=True)
        events[r] = torch.cuda.Event()
        events[r].record()
        pending.append((k, r))
        drain(1)                                       # hand over field k-1 while field k is in flight
    drain(0)
    return out


def _fields_in_order(proc, clip, frame_indices):
    """(position, field) for every frame, whole frames: a few fields per pass where the processor can batch
    them (the tri-frame network, compute_optical_flow_resident_batch), else one call per field."""
    batch = getattr(proc, "compute_optical_flow_resident_batch", None)
    step = (getattr(proc, "TRI_BATCH", None) or getattr(proc, "PAIR_BATCH", 1)) if batch is not None else 1
    for k0 in range(0, len(frame_indices), step):
        chunk = frame_indices[k0:k0 + step]
        flows = batch(clip, chunk) if batch is not None else [proc.compute_optical_flow_resident(clip, f) for f in chunk]
        for j, flow in enumerate(flows):
            yield k0 + j, flow


def run_sharded(proc, clip, frame_indices, tile_mode=False, rank=0, world=1, group=None, on_field=None):
    """Compute the flow field of every frame in `frame_indices` of the device-resident uint8 clip
    [F,H,W,3].  Returns on rank 0 a float32 numpy array [len(frame_indices), H, W, 2]; None elsewhere.
    `on_field(k, field)` (optional) is called on rank 0 for every finished field, k = position in
    `frame_indices`: as the fields finish when one rank computes whole frames, after the gather otherwise."""
    frame_indices = list(frame_indices)
    F, H, W = clip.shape[0], clip.shape[1], clip.shape[2]
    if on_field is not None and world == 1 and not tile_mode:
        return _run_streaming(proc, clip, frame_indices, on_field)
    tiles = _tiles(proc, W, H, tile_mode)
    items = vdist.work_items(frame_indices, len(tiles), tile_major=True)   # keeps the per-crop caches hot
    bounds = [vdist.shard_bounds(len(items), r, world) for r in range(world)]
    sizes = [sum(item_numel(H, W, tiles[t]) for _, t in items[lo:hi]) for lo, hi in bounds]
    lo, hi = bounds[rank]
    local = torch.empty(sizes[rank], dtype=torch.float32, device=clip.device)
    off = 0
    if len(tiles) == 1:                                # whole frames: batched where the processor can
        for _, flow in _fields_in_order(proc, clip, [f for f, _ in items[lo:hi]]):
            n = flow.numel()
            local[off:off + n].copy_(flow.reshape(-1))
            off += n
    else:
        for f, t in items[lo:hi]:
            flow = proc.compute_optical_flow_resident(clip, f, tile=tiles[t])
            n = flow.numel()
            local[off:off + n].copy_(flow.reshape(-1))
            off += n
    parts = vdist.gather_to_rank0(local, sizes, group=group)
    if rank != 0:
        return None
    out = np.zeros((len(frame_indices), H, W, 2), dtype=np.float32)
    slot = {f: k for k, f in enumerate(frame_indices)}
    for (blo, bhi), part in zip(bounds, parts):
        host = part.cpu().numpy()
        off = 0
        for f, t in items[blo:bhi]:
            tile = tiles[t]
            n = item_numel(H, W, tile)
            if tile is None:
                out[slot[f]] = host[off:off + n].reshape(H, W, 2)
            else:
                y, x, th, tw = tile['y'], tile['x'], tile['height'], tile['width']
                out[slot[f], y:y + th, x:x + tw] = host[off:off + n].reshape(th, tw, 2)
            off += n
    if on_field is not None:
        for k in range(len(frame_indices)):
            on_field(k, out[k])
    return out
