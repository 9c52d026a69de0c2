"""Sharded flow-field job: the loop of reference flow_processor.py:959-976 / :1460-1470
(`for i in frames: compute_optical_flow[_tiled]`) spread over the GPUs of one node, host memory to host memory.

Work items are (frame) or (frame, tile); ranks take contiguous blocks (vfml.dist.shard_bounds), keep
the clip resident in their own HBM and compute with no data-path communication.  Finished fields leave the GPU
while the job runs, in chunks of a few items, by one of two routes:

* gather (the API route: the caller wants the fields on rank 0): every `chunk` items each rank contributes its newest
  fields to one gather (RCCL over xGMI; gloo in the CPU tests) that runs beside the computation of the next chunk;
  rank 0 copies the received chunk to pinned host memory on a side stream, and a WORKER THREAD waits for that copy,
  pastes tiles with the reference's hard seams (processing/videoflow_processor.py:277), fills the result array and
  calls `on_field`.  The thread that drives the GPU never waits for a copy or touches a field's bytes: at 8 ranks rank 0
  takes in 8 x 16.6 MB per 1080p field time, which one thread that also launches kernels cannot move.  Rank 0 holds a
  few chunks of receive buffers, not the job, and the tail of a job is one chunk long.
* local sinks (`local_sink`: whole-frame jobs whose output is files on a filesystem every rank sees - the CLI's flow
  cache): NO collective at all.  Every rank copies its own fields to its own pinned buffers and hands them to its own
  sink (its own cache writer); writer throughput then scales with the ranks instead of funnelling into rank 0.

With one rank the gather is the identity and the same pipeline is the D2H ring of a single GPU.

Input side: `ClipFeeder` uploads the uint8 frames from (pageable) host memory through a pinned ring on a side
stream, a few frames ahead of the window being computed, so the clip never has to be stacked, pinned and
uploaded as a whole before the first field starts."""
import os
import queue
import threading
import time

import numpy as np
import torch

from . import dist as vdist

TILE_BLOCK_FRAMES = int(os.environ.get("VFML_TILE_BLOCK", "32"))     # frames per tile-major block of a tiled job


def _tiles(proc, width, height, tile_mode):
    return proc.calculate_tile_grid(width, height)[4] if tile_mode else [None]


def item_numel(height, width, tile):
    h, w = (height, width) if tile is None else (tile['height'], tile['width'])
    return h * w * 2


def lod_shapes(height, width, num_lods):
    """[(h, w)] of LOD levels 1..num_lods-1 of a field (each level halves both sides, rounding up:
    reference storage/cache_manager.py:77-161)."""
    shapes, h, w = [], height, width
    for _ in range(1, max(1, num_lods)):
        h, w = (h + 1) // 2, (w + 1) // 2
        shapes.append((h, w))
    return shapes


class _LazyMaxima:
    """Per-frame maxima of the host frames, computed when asked for (MemFlow's value-range heuristic,
    memflow_inference_isolated.py:81-85, is the only reader: a VideoFlow job never pays for them)."""

    def __init__(self, frames):
        self._frames, self._vals = frames, {}

    def __len__(self):
        return len(self._frames)

    def __getitem__(self, i):
        v = self._vals.get(i)
        if v is None:
            v = self._vals[i] = float(self._frames[i].max())
        return v


class ClipFeeder:
    """Device-resident uint8 clip [F,H,W,3] filled frame by frame from host arrays, ahead of use.

    `ensure(upto)` makes frames lo..upto usable by work queued on the current stream afterwards (lo = 0, or what
    `skip_to` left out): frames not yet uploaded are copied into a pinned ring (host memcpy, 1 ms per 1080p frame) and
    from there to the device on a side stream; the current stream waits for those copies only.  The reference uploads T
    float32 frames per field from pageable memory (processing/videoflow_processor.py:161); here every frame crosses PCIe
    once, as uint8, while the GPU computes earlier fields."""

    RING = 4

    def __init__(self, frames, device):
        """frames: list of uint8 [H,W,3] arrays; entries may be None where this process will never need the frame (another
        rank's stretch of a sharded clip): those are never uploaded, and asking for one raises."""
        f0 = next(f for f in frames if f is not None)
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
        f0 = next(f for f in frames if f is not None)
        if (any(f is not None and (f.dtype != np.uint8 or f.shape != f0.shape) for f in frames) or len(frames) != self.clip.shape[0]
                or tuple(f0.shape) != tuple(self.clip.shape[1:])):
            raise ValueError("ClipFeeder expects uint8 frames of one shape (and, on reset, the shape it was built for)")
        self.frames = frames
        self.lo = 0            # frames [lo, next) are in the clip; frames below lo were skipped, never uploaded
        self.next = 0
        # a frame never changes once it is uploaded: the token is fixed although uploads move clip._version
        self.clip._vfml_clip_token = (new_id(), "fed")
        self.clip._vfml_frame_maxima = _LazyMaxima(frames)
        self.clip._vfml_frames_ready = 0
        if self.on_gpu:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))   # earlier readers of the old frames

    def skip_to(self, frame):
        """Frames before `frame` will not be needed (a rank whose shard starts later in the clip): they are never
        uploaded.  Only moves forward, and only while nothing has been uploaded yet (the resident range stays one
        contiguous run [lo, next): with an uploaded prefix the frames in between simply go up as `ensure` reaches them)."""
        frame = max(0, min(frame, len(self.frames)))
        if frame > self.next and self.next == self.lo:
            self.lo = self.next = frame
            self.clip._vfml_frames_ready = self.next

    def require(self, first):
        """A window is about to read frames from `first` on: they must be resident (a second job on the same feeder that
        reaches below what an earlier job's skip_to left out would read device memory nobody wrote)."""
        if first < self.lo:
            raise RuntimeError(f"ClipFeeder: frame {first} was skipped by an earlier shard of this feeder (frames below "
                               f"{self.lo} were never uploaded); reset() the feeder or build a new one for this job")

    def ensure(self, upto, need=None):
        """Upload ahead to frame `upto` (as far as this process holds the frames); `need`: the last frame the caller is
        about to read - not having that one is an error."""
        upto = min(upto, len(self.frames) - 1)
        while self.next == self.lo and self.next <= upto and self.frames[self.next] is None:
            self.lo = self.next = self.next + 1         # (leading frames this process does not hold: skipped like skip_to's)
        for f in range(self.next, upto + 1):
            if self.frames[f] is None:          # another rank's stretch: the read-ahead stops here
                upto = f - 1
                break
        if need is not None and min(need, len(self.frames) - 1) > max(upto, self.next - 1):
            raise RuntimeError(f"ClipFeeder: frame {max(upto, self.next - 1) + 1} is not held by this process (None in its "
                               f"frame list) but a window reaches frame {need}")
        if upto < self.next:
            return
        if not self.on_gpu:
            for f in range(self.next, upto + 1):
                self.clip[f] = torch.from_numpy(np.ascontiguousarray(self.frames[f]))
            self.next = upto + 1
            self.clip._vfml_frames_ready = self.next
            return
        last = None
        for f in range(self.next, upto + 1):
            r = f % self.RING
            if self.events[r] is not None:
                self.events[r].synchronize()          # the slot's previous upload has left the pinned buffer
            np.copyto(self.ring_np[r], self.frames[f])
            with torch.cuda.stream(self.stream):
                self.clip[f].copy_(self.ring[r], non_blocking=True)
                last = torch.cuda.Event()
                last.record(self.stream)
            self.events[r] = last
        self.next = upto + 1
        self.clip._vfml_frames_ready = self.next      # frames [lo, next) are (stream-ordered) in the clip: what a prefetch may read
        torch.cuda.current_stream(self.device).wait_event(last)    # (copies on one stream complete in order)


def _fields_in_order(proc, clip, frame_indices, before=None):
    """(position, field) for every frame, whole frames: a few fields per pass where the processor can batch
    them (the tri-frame network, compute_optical_flow_resident_batch), else one call per field."""
    batch = getattr(proc, "compute_optical_flow_resident_batch", None)
    step = (getattr(proc, "TRI_BATCH", None) or getattr(proc, "PAIR_BATCH", 1)) if batch is not None else 1
    for k0 in range(0, len(frame_indices), step):
        chunk = frame_indices[k0:k0 + step]
        if before is not None:
            before(min(chunk), max(chunk))
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
NBUF = 3       # chunk buffers in flight: one being computed, one being copied out, one being unpacked


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


def tile_items(frame_indices, n_tiles, block=None):
    """Work items of a tiled job: frame-major BLOCKS of `block` frames, tile-major inside a block.  Tile-major order keeps a
    crop's sliding-window caches hot and lets the encoder prefetch see the next frame of the same tile; blocks bound what
    rank 0 holds: a frame is complete - and handed on - once its block's last tile has passed, so partial frames never
    exceed a block or two per job (66 MB per 4K frame) and the cache writer works beside the job instead of after it.  Each
    block boundary restarts a tile's sliding window (T - 1 frames re-encoded per tile: ~1.5 % of a 32-frame block)."""
    frame_indices = list(frame_indices)
    block = block or TILE_BLOCK_FRAMES
    if n_tiles == 1:
        return [(f, 0) for f in frame_indices]
    items = []
    for b0 in range(0, len(frame_indices), block):
        items += vdist.work_items(frame_indices[b0:b0 + block], n_tiles, tile_major=True)
    return items


class _Unpacker(threading.Thread):
    """The thread that turns chunks that reached host memory into frames / callbacks.  `put((c, b, event))`: chunk c sits
    (after `event`) in chunk buffer b; when it is unpacked `free[b]` is set and the buffer may be overwritten."""

    def __init__(self, fn, nbuf):
        super().__init__(name="vfml-unpack", daemon=True)
        self.fn = fn
        self.q = queue.Queue(maxsize=nbuf)
        self.free = [threading.Event() for _ in range(nbuf)]
        for e in self.free:
            e.set()
        self.error = None
        self.busy = []                   # seconds of unpack work per chunk (VFML_RUNNER_TIMING)
        self.start()

    def run(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            c, b, ev = item
            try:
                if self.error is None:
                    if ev is not None:
                        ev.synchronize()
                    t0 = time.perf_counter()
                    self.fn(c, b)
                    self.busy.append(time.perf_counter() - t0)
            except BaseException as e:          # re-raised on the job's thread
                self.error = e
            finally:
                self.free[b].set()

    def put(self, item):
        self.check()
        self.q.put(item)

    def check(self):
        if self.error is not None:
            err, self.error = self.error, None
            raise err

    def close(self):
        self.q.put(None)
        self.join()
        self.check()


def run_sharded(proc, clip, frame_indices, tile_mode=False, rank=0, world=1, group=None, on_field=None,
                collect=True, num_lods=0, chunk=None, feeder=None, prepare_only=False, local_sink=None, out=None):
    """Compute the flow field of every frame in `frame_indices` of the device-resident uint8 clip [F,H,W,3]
    (`clip` may be None when a ClipFeeder is given: its clip is used and fed as the job advances).
    Returns on rank 0 a float32 numpy array [len(frame_indices), H, W, 2] (None with collect=False, when the
    fields only go to `on_field`); None on the other ranks.
    `on_field(k, field, lods)` is called on rank 0 for every finished frame as soon as it (in tile mode: its last tile)
    has reached host memory - from the unpack thread(s), in completion order, so it must be thread-safe - k = position in
    `frame_indices`; `field` is a host array the callee may keep; `lods` is None or, with num_lods > 1 (whole frames only),
    the reference's LOD pyramid [field, lod1, ...] reduced on the GPU that computed the field (vfml_flow_lod, bit-identical
    to the reference's loop).
    `local_sink(k, field, lods)` (whole-frame jobs only; excludes collect / on_field): called on EVERY rank for that
    rank's own fields - no collective, nothing funnels into rank 0 (the CLI's per-rank cache writers).
    prepare_only: allocate the job's staging buffers (kept for later jobs of the same geometry) and return.
    out: optional result array (rank 0, with collect): float32 [len(frame_indices), H, W, 2], filled and returned instead of a
    fresh one - a caller that times the job hands in memory it has already touched (at 8 ranks x 41 fields/s the result
    grows by 5 GB/s; first-touch page faults of a fresh array would be part of what is measured)."""
    frame_indices = list(frame_indices)
    if feeder is not None:
        clip = feeder.clip
    F, H, W = clip.shape[0], clip.shape[1], clip.shape[2]
    tiles = _tiles(proc, W, H, tile_mode)
    whole = len(tiles) == 1
    on_gpu = clip.is_cuda
    local = local_sink is not None
    if local and (not whole or collect or on_field is not None):
        raise ValueError("local_sink applies to whole-frame jobs and replaces collect / on_field")
    lods_on = num_lods > 1 and whole and on_gpu
    lshapes = lod_shapes(H, W, num_lods) if lods_on else []
    items = tile_items(frame_indices, len(tiles))
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
    K = chunk or default_chunk(slot, max(counts) if counts else 1, 1 if local else world, batch)
    n_chunks = -(-max(counts) // K) if counts and max(counts) > 0 else 0
    if local:
        n_chunks = -(-counts[rank] // K) if counts[rank] > 0 else 0
    dev = clip.device
    nb = min(NBUF, n_chunks)
    gather = world > 1 and not local
    sink_here = local or rank == 0                          # this rank brings chunks to its host memory
    hworld = world if gather else 1                         # ranks per host chunk buffer
    send = [_buffer(f"send{i}", (K, slot), dev) for i in range(nb)]
    recv = host = None
    if sink_here and n_chunks:
        if gather:
            recv = [[_buffer(f"recv{i}.{r}", (K, slot), dev) for r in range(world)] for i in range(nb)]
        if on_gpu:
            host = [_buffer(f"host{i}", (hworld, K, slot), dev, pinned=True) for i in range(nb)]
    if prepare_only:
        if gather and n_chunks:
            # one collective of the job's shape now: the backend's peer-to-peer connections (RCCL sets them up on first
            # use) and its staging are then in place before a timed job starts
            torch.distributed.gather(send[0], recv[0] if rank == 0 else None, dst=0, group=group)
            if on_gpu:
                torch.cuda.synchronize(dev)
        return None
    side = torch.cuda.Stream(device=dev) if on_gpu and sink_here else None
    seq = getattr(proc, "sequence_length", 1)

    if rank == 0 and collect and not local:
        if out is None:
            out = np.zeros((len(frame_indices), H, W, 2), dtype=np.float32)
        elif out.shape != (len(frame_indices), H, W, 2) or out.dtype != np.float32 or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous float32 array of shape {(len(frame_indices), H, W, 2)}")
    else:
        out = None
    slot_of = {f: k for k, f in enumerate(frame_indices)}
    partial, left = {}, {}                                  # tile mode without `collect`: frames being assembled
    tile_lock = threading.Lock()

    if feeder is not None and mine:
        feeder.skip_to(min(f for f, _ in mine) - seq)       # (a window reaches at most seq - 1 frames back)

    def window(f):
        """Frames the processor reads for field f (VideoFlow: centred, seq // 2 either side; MemFlow: the seq - 1 before it)."""
        wi = getattr(proc, "window_indices", None)
        if wi is None:
            return [max(0, f - (seq - 1)), min(F - 1, f + seq // 2)]
        try:
            return wi(F, f)
        except TypeError:
            return wi(f)

    def feed(first, last):
        if feeder is not None:
            feeder.require(min(window(first)))
            feeder.ensure(last + seq, need=max(window(last)))   # the window's last frame, and a few more ahead of use

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
                feed(f, f)
                # (inside a block items are tile-major: the call after this one is the NEXT FRAME of the same tile - what
                # the processor's encoder prefetch assumes for whole frames holds for tiles too)
                proc.tiles_in_frame_order = not whole
                flow = proc.compute_optical_flow_resident(clip, f, tile=tiles[t])
                sbuf[j, :flow.numel()].copy_(flow.reshape(-1))

    def deliver(k, field, lods):
        if local:
            local_sink(k, field, lods)
        elif on_field is not None:
            on_field(k, field, lods)

    def unpack_rank(c, rows, r):
        """chunk c of rank r (rows[r]: its K slots in host memory; this rank's own without a gather) -> frames / callbacks."""
        src = r if gather else rank
        blo = bounds[src][0]
        for j in range(min(K, counts[src] - c * K)):
            f, t = items[blo + c * K + j]
            k = slot_of[f]
            row = rows[r][j]
            tile = tiles[t]
            n = item_numel(H, W, tile)
            if tile is None:
                if out is not None:
                    out[k] = row[:n].reshape(H, W, 2)
                    field = out[k]
                else:
                    field = row[:n].reshape(H, W, 2).copy()       # the chunk buffer is reused
                lods = None
                if lods_on:
                    lods, off = [field], n
                    for h, w in lshapes:
                        lods.append(row[off:off + h * w * 2].reshape(h, w, 2).copy())
                        off += h * w * 2
                deliver(k, field, lods)
            else:
                y, x, th, tw = tile['y'], tile['x'], tile['height'], tile['width']
                with tile_lock:
                    if out is not None:
                        frame = out[k]
                    else:
                        frame = partial.get(k)
                        if frame is None:
                            frame = partial[k] = np.zeros((H, W, 2), dtype=np.float32)
                frame[y:y + th, x:x + tw] = row[:n].reshape(th, tw, 2)
                with tile_lock:
                    left[k] = left.get(k, len(tiles)) - 1
                    done = left[k] == 0
                    if done:
                        partial.pop(k, None)
                if done:
                    deliver(k, frame, None)

    # with several ranks' fields per chunk the copies out of the chunk buffer run on a few threads (numpy releases the
    # GIL for them): 8 x 2 x 16.6 MB per chunk at 1080p is more than one thread moves in a chunk's compute time
    pool = None
    if gather and rank == 0 and world > 2:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=min(4, world), thread_name_prefix="vfml-unpack")

    def unpack(c, b):
        if on_gpu:
            rows = host[b].numpy()
        else:       # a CPU "device" (the gloo tests): the chunk is unpacked where the collective left it, no staging copy
            rows = [t.numpy() for t in (recv[b] if gather else [send[b]])]
        if pool is None:
            for r in range(hworld):
                unpack_rank(c, rows, r)
        else:
            for fut in [pool.submit(unpack_rank, c, rows, r) for r in range(hworld)]:
                fut.result()

    unpacker = _Unpacker(unpack, nb) if (sink_here and n_chunks) else None
    d2h_done = [None] * max(nb, 1)      # per buffer: event of the last copy out of send[b] / recv[b]
    waited = [0.0]                      # seconds this thread was blocked (host-blocking collective, unpack thread behind)

    def wait_free(b):
        t_w = time.perf_counter()
        unpacker.free[b].wait()
        waited[0] += time.perf_counter() - t_w

    def finish(c, work, ev_done):
        """Chunk c has been queued (and its gather started): start bringing it to the host and hand it to the unpack
        thread.  The GPU already has the next chunk's kernels queued; this thread waits only when the unpack thread is a
        whole ring of chunks behind."""
        b = c % nb
        if gather and rank != 0:
            work.wait()                                     # before this send buffer is written again
            return
        if not sink_here:
            return
        if not on_gpu:
            if gather:
                t_w = time.perf_counter()
                work.wait()                                 # (gloo: blocks this thread until the chunk has arrived)
                waited[0] += time.perf_counter() - t_w
            unpacker.free[b].clear()
            unpacker.put((c, b, None))
            return
        wait_free(b)                                        # host[b]'s previous chunk has been unpacked
        unpacker.free[b].clear()
        with torch.cuda.stream(side):
            if gather:
                work.wait()                                 # the side stream waits for the collective
                for r in range(world):
                    nv = min(K, counts[r] - c * K)          # (slots past the end of a rank's shard are not copied)
                    if nv > 0:
                        host[b][r, :nv].copy_(recv[b][r][:nv], non_blocking=True)
            else:
                side.wait_event(ev_done)
                nv = min(K, counts[rank] - c * K)
                host[b][0, :nv].copy_(send[b][:nv], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        d2h_done[b] = ev
        unpacker.put((c, b, ev))

    # host-side seconds per chunk (VFML_RUNNER_TIMING): (enqueue the chunk's launches, hand the previous chunk on, of which
    # blocked on a host-blocking collective - gloo - or on the unpack thread being a ring of chunks behind)
    trace = [] if os.environ.get("VFML_RUNNER_TIMING") else None
    prev = None
    try:
        for c in range(n_chunks):
            b = c % nb
            t_a = time.perf_counter()
            if on_gpu and d2h_done[b] is not None:
                # send[b] / recv[b] are about to be rewritten: behind the copy that last read them (device-side order only)
                torch.cuda.current_stream(dev).wait_event(d2h_done[b])
            elif not on_gpu and unpacker is not None:
                wait_free(b)                                # (CPU "device": the unpack thread reads send[b] / recv[b] themselves)
            t_a2 = time.perf_counter()
            compute_chunk(c, send[b])
            t_b = time.perf_counter()
            work = ev_done = None
            if gather:
                work = torch.distributed.gather(send[b], recv[b] if rank == 0 else None, dst=0, group=group, async_op=True)
            elif on_gpu:
                ev_done = torch.cuda.Event()
                ev_done.record()
            if prev is not None:
                finish(*prev)
            if trace is not None:
                trace.append((t_b - t_a2, (time.perf_counter() - t_b) + (t_a2 - t_a), waited[0]))
                waited[0] = 0.0
            prev = (c, work, ev_done)
        if prev is not None:
            finish(*prev)
    finally:
        if unpacker is not None:
            unpacker.close()
        if pool is not None:
            pool.shutdown(wait=True)
    if trace is not None and sink_here:
        busy = unpacker.busy if unpacker is not None else []
        print(f"[runner] rank {rank} host ms per chunk (enqueue, hand-off [of which blocked] | unpack thread): " +
              " ".join(f"({1e3 * a:.1f},{1e3 * f:.1f}[{1e3 * w:.1f}]|{1e3 * (busy[i] if i < len(busy) else 0):.1f})"
                       for i, (a, f, w) in enumerate(trace)), flush=True)
        run_sharded.last_trace = {"enqueue": [a for a, _, _ in trace], "handoff": [f - w for _, f, w in trace],
                                  "blocked": [w for _, _, w in trace], "unpack": list(busy)}
    return out
