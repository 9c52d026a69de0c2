"""Sharded flow-field job: the loop of reference flow_processor.py:959-976 / :1460-1470
(`for i in frames: compute_optical_flow[_tiled]`) spread over the GPUs of one node, host memory to host memory.

Work items are (frame) or (frame, tile); ranks take contiguous blocks (vfml.dist.shard_bounds), keep
the clip resident in their own HBM and compute with no data-path communication.  Finished fields STREAM to
rank 0 while the job runs: every `chunk` items each rank contributes its newest fields to one gather (RCCL over
xGMI; gloo in the CPU tests) that runs beside the computation of the next chunk, rank 0 copies the received
chunk to pinned host memory on a side stream, pastes tiles with the reference's hard seams
(processing/videoflow_processor.py:277) and hands every finished frame to `on_field` (the cache writer).
Rank 0 therefore holds two chunks of receive buffers, not the whole job, and the tail of a job is one chunk
long.  With one rank the gather is the identity and the same pipeline is the D2H ring of a single GPU.

Input side: `ClipFeeder` uploads the uint8 frames from (pageable) host memory through a pinned ring on a side
stream, a few frames ahead of the window being computed, so the clip never has to be stacked, pinned and
uploaded as a whole before the first field starts."""
import numpy as np
import torch

from . import dist as vdist


def _tiles(proc, width, height, tile_mode):
    return proc.calculate_tile_grid(width, height)[4] if tile_mode else [None]


def item_numel(height, width, tile):
    h, w = (height, width) if tile is None else (tile['height'], tile['width'])
    return h * w * 2


def lod_shapes(height, width, num_lods):
    """[(h, w)] of LOD levels 1..num_lods-1 of a field (each level halves both sides, rounding up:
    reference storage/cache_manager.py:77-161)."""
    out, h, w = [], height, width
    for _ in range(1, max(1, num_lods)):
        h, w = (h + 1) // 2, (w + 1) // 2
        out.append((h, w))
    return out


class ClipFeeder:
    """Device-resident uint8 clip [F,H,W,3] filled frame by frame from host arrays, ahead of use.

    `ensure(upto)` makes frames 0..upto usable by work queued on the current stream afterwards: frames not yet
    uploaded are copied into a pinned ring (host memcpy, 1 ms per 1080p frame) and from there to the device on a
    side stream; the current stream waits for those copies only.  The reference uploads T float32 frames per
    field from pageable memory (processing/videoflow_processor.py:161); here every frame crosses PCIe once, as
    uint8, while the GPU computes earlier fields."""

    RING = 4

    def __init__(self, frames, device):
        f0 = frames[0]
        self.device = torch.device(device)
        self.clip = torch.empty((len(frames),) + tuple(f0.shape), dtype=torch.uint8, device=self.device)
        self.on_gpu = self.device.type == "cuda"
        if self.on_gpu:
            self.ring = [torch.empty(tuple(f0.shape), dtype=torch.uint8).pin_memory() for _ in range(self.RING)]
            self.ring_np = [r.numpy() for r in self.ring]
            self.events = [None] * self.RING
            self.stream = torch.cuda.Stream(device=self.device)
        self.reset(frames)

    def reset(self, frames):
        """Start over with another list of frames of the same shape and count (the buffers are kept; the clip gets
        a new identity, so nothing cached for the old frames is found under the new ones)."""
        from .clip_id import new_id
        f0 = frames[0]
        if (any(f.dtype != np.uint8 or f.shape != f0.shape for f in frames) or len(frames) != self.clip.shape[0]
                or tuple(f0.shape) != tuple(self.clip.shape[1:])):
            raise ValueError("ClipFeeder expects uint8 frames of one shape (and, on reset, the shape it was built for)")
        self.frames = frames
        self.next = 0
        # a frame never changes once it is uploaded: the token is fixed although uploads move clip._version
        self.clip._vfml_clip_token = (new_id(), "fed")
        # per-frame maxima, known at upload: MemFlow's value-range heuristic (memflow_inference_isolated.py:81-85)
        self.clip._vfml_frame_maxima = [None] * len(frames)
        self.clip._vfml_frames_ready = 0
        if self.on_gpu:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))   # earlier readers of the old frames

    def skip_to(self, frame):
        """Frames before `frame` will not be needed (a rank whose shard starts later in the clip): they are never
        uploaded.  Only moves forward."""
        self.next = max(self.next, min(frame, len(self.frames)))
        self.clip._vfml_frames_ready = self.next

    def ensure(self, upto):
        upto = min(upto, len(self.frames) - 1)
        if upto < self.next:
            return
        maxima = self.clip._vfml_frame_maxima
        if not self.on_gpu:
            for f in range(self.next, upto + 1):
                self.clip[f] = torch.from_numpy(np.ascontiguousarray(self.frames[f]))
                maxima[f] = float(self.frames[f].max())
            self.next = upto + 1
            self.clip._vfml_frames_ready = self.next
            return
        last = None
        for f in range(self.next, upto + 1):
            r = f % self.RING
            maxima[f] = float(self.frames[f].max())
            if self.events[r] is not None:
                self.events[r].synchronize()          # the slot's previous upload has left the pinned buffer
            np.copyto(self.ring_np[r], self.frames[f])
            with torch.cuda.stream(self.stream):
                self.clip[f].copy_(self.ring[r], non_blocking=True)
                last = torch.cuda.Event()
                last.record(self.stream)
            self.events[r] = last
        self.next = upto + 1
        self.clip._vfml_frames_ready = self.next      # frames [0, next) are (stream-ordered) in the clip: what a prefetch may read
        torch.cuda.current_stream(self.device).wait_event(last)    # (copies on one stream complete in order)


def _fields_in_order(proc, clip, frame_indices, before=None):
    """(position, field) for every frame, whole frames: a few fields per pass where the processor can batch
    them (the tri-frame network, compute_optical_flow_resident_batch), else one call per field."""
    batch = getattr(proc, "compute_optical_flow_resident_batch", None)
    step = (getattr(proc, "TRI_BATCH", None) or getattr(proc, "PAIR_BATCH", 1)) if batch is not None else 1
    for k0 in range(0, len(frame_indices), step):
        chunk = frame_indices[k0:k0 + step]
        if before is not None:
            before(max(chunk))
        flows = batch(clip, chunk) if batch is not None else [proc.compute_optical_flow_resident(clip, f) for f in chunk]
        for j, flow in enumerate(flows):
            yield k0 + j, flow


def default_chunk(slot_floats, n_items, world, batch=1):
    """Items per chunk: with a single rank one field (its D2H then hides under the next field), with several ranks two
    (one collective per two fields of every rank; one when an item exceeds 64 MB) - and never fewer than the processor
    computes per pass of the engine (`batch`: eight triples of the tri-frame network, three MemFlow pairs), or the
    chunking would undo the batching."""
    k = 1 if world == 1 else max(1, min(2, (128 << 20) // max(1, 4 * slot_floats)))
    return int(max(1, min(max(k, batch), n_items)))


_BUFFERS = {}


def _buffer(kind, shape, device, pinned=False):
    """Staging buffers are kept across jobs of one geometry (pinned host allocations cost ~0.2 ms per MB)."""
    key = (kind, tuple(shape), str(device), pinned)
    t = _BUFFERS.get(key)
    if t is None:
        t = torch.empty(shape, dtype=torch.float32, device="cpu" if pinned else device)
        if pinned:
            t = t.pin_memory()
        _BUFFERS[key] = t
    return t


def release_buffers():
    _BUFFERS.clear()


def run_sharded(proc, clip, frame_indices, tile_mode=False, rank=0, world=1, group=None, on_field=None,
                collect=True, num_lods=0, chunk=None, feeder=None, prepare_only=False):
    """Compute the flow field of every frame in `frame_indices` of the device-resident uint8 clip [F,H,W,3]
    (`clip` may be None when a ClipFeeder is given: its clip is used and fed as the job advances).
    Returns on rank 0 a float32 numpy array [len(frame_indices), H, W, 2] (None with collect=False, when the
    fields only go to `on_field`); None on the other ranks.
    `on_field(k, field, lods)` is called on rank 0 for every finished frame as soon as it (in tile mode: its last
    tile) has reached host memory, k = position in `frame_indices`; `field` is a host array the callee may keep;
    `lods` is None or, with num_lods > 1 (whole frames only), the reference's LOD pyramid [field, lod1, ...] reduced
    on the GPU that computed the field (vfml_flow_lod, bit-identical to the reference's loop).
    prepare_only: allocate the job's staging buffers (kept for later jobs of the same geometry) and return."""
    frame_indices = list(frame_indices)
    if feeder is not None:
        clip = feeder.clip
    F, H, W = clip.shape[0], clip.shape[1], clip.shape[2]
    tiles = _tiles(proc, W, H, tile_mode)
    whole = len(tiles) == 1
    on_gpu = clip.is_cuda
    lods_on = num_lods > 1 and whole and on_gpu
    lshapes = lod_shapes(H, W, num_lods) if lods_on else []
    items = vdist.work_items(frame_indices, len(tiles), tile_major=True)   # keeps the per-crop caches hot
    bounds = [vdist.shard_bounds(len(items), r, world) for r in range(world)]
    counts = [hi - lo for lo, hi in bounds]
    lo, hi = bounds[rank]
    mine = items[lo:hi]
    slot = max([item_numel(H, W, t) for t in tiles]) + sum(h * w * 2 for h, w in lshapes)
    batch = 1
    if whole and getattr(proc, "compute_optical_flow_resident_batch", None) is not None:
        batch = getattr(proc, "TRI_BATCH", None) or getattr(proc, "PAIR_BATCH", 1)
        if getattr(proc, "TRI_BATCH", None) and not getattr(getattr(getattr(proc, "core", None), "model", None), "tri_frame", False):
            batch = 1                                       # (the multi-frame network takes one window per pass)
    K = chunk or default_chunk(slot, max(counts) if counts else 1, world, batch)
    n_chunks = -(-max(counts) // K) if counts and max(counts) > 0 else 0
    dev = clip.device
    nb = min(2, n_chunks)
    send = [_buffer(f"send{i}", (K, slot), dev) for i in range(nb)]
    recv = host = None
    if rank == 0 and n_chunks:
        if world > 1:
            recv = [[_buffer(f"recv{i}.{r}", (K, slot), dev) for r in range(world)] for i in range(nb)]
        host = [_buffer(f"host{i}", (world, K, slot), dev, pinned=on_gpu) for i in range(nb)]
    if prepare_only:
        if world > 1 and n_chunks:
            # one collective of the job's shape now: the backend's peer-to-peer connections (RCCL sets them up on first
            # use) and its staging are then in place before a timed job starts
            torch.distributed.gather(send[0], recv[0] if rank == 0 else None, dst=0, group=group)
            if on_gpu:
                torch.cuda.synchronize(dev)
        return None
    side = torch.cuda.Stream(device=dev) if on_gpu and rank == 0 else None
    seq = getattr(proc, "sequence_length", 1)

    out = np.zeros((len(frame_indices), H, W, 2), dtype=np.float32) if (rank == 0 and collect) else None
    slot_of = {f: k for k, f in enumerate(frame_indices)}
    partial, left = {}, {}                                  # tile mode without `collect`: frames being assembled

    if feeder is not None and mine:
        feeder.skip_to(min(f for f, _ in mine) - seq)       # (a window reaches at most seq - 1 frames back)

    def feed(frame):
        if feeder is not None:
            feeder.ensure(frame + seq)                      # the window's last frame and a few ahead

    def compute_chunk(c, sbuf):
        part = mine[c * K:(c + 1) * K]
        if not part:
            return
        if whole:
            for j, flow in _fields_in_order(proc, clip, [f for f, _ in part], before=feed):
                row = sbuf[j]
                n = H * W * 2
                row[:n].copy_(flow.reshape(-1))
                if lods_on:
                    from . import hip
                    off = n
                    for lvl in hip.flow_lods(row[:n].view(H, W, 2), num_lods)[1:]:
                        row[off:off + lvl.numel()].copy_(lvl.reshape(-1))
                        off += lvl.numel()
        else:
            for j, (f, t) in enumerate(part):
                feed(f)
                # (items are tile-major: the call after this one is the NEXT FRAME of the same tile - what the processor's
                # encoder prefetch assumes for whole frames holds for tiles too)
                proc.tiles_in_frame_order = not whole
                flow = proc.compute_optical_flow_resident(clip, f, tile=tiles[t])
                sbuf[j, :flow.numel()].copy_(flow.reshape(-1))

    def deliver(k, field, lods):
        if on_field is not None:
            on_field(k, field, lods)

    def unpack(c, hbuf):
        """rank 0: chunk c of every rank, now in host memory -> frames / callbacks."""
        harr = hbuf.numpy()
        for r in range(world):
            blo = bounds[r][0]
            for j in range(min(K, counts[r] - c * K)):
                f, t = items[blo + c * K + j]
                k = slot_of[f]
                row = harr[r, j]
                tile = tiles[t]
                n = item_numel(H, W, tile)
                if tile is None:
                    if out is not None:
                        out[k] = row[:n].reshape(H, W, 2)
                        field = out[k]
                    else:
                        field = row[:n].reshape(H, W, 2).copy()       # the pinned chunk buffer is reused
                    lods = None
                    if lods_on:
                        lods, off = [field], n
                        for h, w in lshapes:
                            lods.append(row[off:off + h * w * 2].reshape(h, w, 2).copy())
                            off += h * w * 2
                    deliver(k, field, lods)
                else:
                    y, x, th, tw = tile['y'], tile['x'], tile['height'], tile['width']
                    if out is not None:
                        frame = out[k]
                    else:
                        frame = partial.get(k)
                        if frame is None:
                            frame = partial[k] = np.zeros((H, W, 2), dtype=np.float32)
                    frame[y:y + th, x:x + tw] = row[:n].reshape(th, tw, 2)
                    left[k] = left.get(k, len(tiles)) - 1
                    if left[k] == 0:
                        partial.pop(k, None)
                        deliver(k, frame, None)

    def finish(c, work, ev_done):
        """Chunk c has been queued (and its gather started): bring it to the host and unpack it.  The GPU already
        has the next chunk's kernels queued, so the host waits here while the device stays busy."""
        b = c % 2
        if world > 1 and rank != 0:
            work.wait()                                     # before this send buffer is written again
            return
        if not on_gpu:
            if world > 1:
                work.wait()
                host[b].copy_(torch.stack(recv[b]))
            else:
                host[b][0].copy_(send[b])
            unpack(c, host[b])
            return
        with torch.cuda.stream(side):
            if world > 1:
                work.wait()                                 # the side stream waits for the collective
                for r in range(world):
                    host[b][r].copy_(recv[b][r], non_blocking=True)
            else:
                side.wait_event(ev_done)
                host[b][0].copy_(send[b], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        ev.synchronize()
        unpack(c, host[b])

    import os
    import time
    trace = [] if os.environ.get("VFML_RUNNER_TIMING") else None     # host-side seconds per chunk: (enqueue, finish)
    prev = None
    for c in range(n_chunks):
        b = c % 2
        t_a = time.perf_counter()
        compute_chunk(c, send[b])
        t_b = time.perf_counter()
        work = ev_done = None
        if world > 1:
            work = torch.distributed.gather(send[b], recv[b] if rank == 0 else None, dst=0, group=group, async_op=True)
        elif on_gpu:
            ev_done = torch.cuda.Event()
            ev_done.record()
        if prev is not None:
            finish(*prev)
        if trace is not None:
            trace.append((t_b - t_a, time.perf_counter() - t_b))
        prev = (c, work, ev_done)
    if prev is not None:
        finish(*prev)
    if trace is not None and rank == 0:
        print("[runner] host ms per chunk (enqueue, finish previous): " +
              " ".join(f"({1e3 * a:.1f},{1e3 * f:.1f})" for a, f in trace), flush=True)
    return out
