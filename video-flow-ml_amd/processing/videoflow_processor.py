"""VideoFlowProcessor — frame windows, tiles and numpy<->tensor plumbing around VideoFlowCore.

API mirror of reference processing/videoflow_processor.py:20-361.  Behaviour kept bit-for-bit:
window selection and padding asymmetry (:133-147), u8 -> float32 /255 and HWC->CHW (:150-161),
fixed 1280x1280 row-major tiles with ragged edge tiles and hard seams (:73-110, :231-283),
[H,W,2] float32 numpy output (:185).
"""
import collections
import hashlib
import os

import numpy as np
import torch

from .videoflow_core import VideoFlowCore

try:                                    # content keys of host frames: xxh3 runs at memory speed
    import xxhash

    def _digest(buf):
        return xxhash.xxh3_128_digest(buf)
except ImportError:                     # pragma: no cover - xxhash is part of the image
    def _digest(buf):
        return hashlib.blake2b(buf, digest_size=16).digest()

HOST_FRAME_CACHE = 24                   # uint8 frames kept in HBM for the host-array API (LRU)


class VideoFlowProcessor:
    def __init__(self, device, fast_mode=False, tile_mode=False, sequence_length=5,
                 dataset='sintel', architecture='mof', variant='standard'):
        self.device = device
        self.fast_mode = fast_mode
        self.tile_mode = tile_mode
        self.sequence_length = sequence_length
        self.dataset = dataset
        self.architecture = architecture
        self.variant = variant
        self.core = VideoFlowCore(device, fast_mode, dataset, architecture, variant)
        self._dev_frames = collections.OrderedDict()      # content key -> uint8 [H,W,3] device tensor
        print("VideoFlow Processor initialized:")
        for label, value in (("Device", device), ("Fast mode", fast_mode), ("Tile mode", tile_mode),
                             ("Sequence length", sequence_length), ("Dataset", dataset),
                             ("Architecture", architecture.upper()), ("Variant", variant)):
            print(f"  {label}: {value}")

    def load_model(self):
        print(f"VideoFlow model loaded successfully from: {self.core.load_model()}")

    # -- tiles ------------------------------------------------------------------------------
    def calculate_tile_grid(self, width, height, tile_size=1280):
        """-> (tile_width, tile_height, cols, rows, tiles_info); tiles row-major, edge tiles clipped."""
        cols = -(-width // tile_size)
        rows = -(-height // tile_size)
        tiles = [{'x': c * tile_size, 'y': r * tile_size,
                  'width': min(tile_size, width - c * tile_size),
                  'height': min(tile_size, height - r * tile_size),
                  'col': c, 'row': r}
                 for r in range(rows) for c in range(cols)]
        return tile_size, tile_size, cols, rows, tiles

    def extract_tile(self, frame, tile_info):
        x, y = tile_info['x'], tile_info['y']
        return frame[y:y + tile_info['height'], x:x + tile_info['width']]

    # -- windows ----------------------------------------------------------------------------
    def window_indices(self, num_frames, frame_idx):
        """Frame indices of the T-frame window the reference builds for `frame_idx` (:133-147):
        centred, clipped to the clip, then padded to T by repeating the first frame at the front when
        the window starts at frame 0, else the last frame at the back."""
        T = self.sequence_length
        half = T // 2
        start = max(0, frame_idx - half)
        idx = list(range(start, min(num_frames, frame_idx + half + 1)))
        while len(idx) < T:
            if start == 0:
                idx.insert(0, idx[0])
            else:
                idx.append(idx[-1])
        return idx[:T]

    def prepare_frame_sequence(self, frames, frame_idx):
        """list of [H,W,3] arrays -> float32 tensor [1,T,3,H,W] on the device; u8 is scaled by 1/255,
        float input passes through unscaled (:152-157)."""
        planes = []
        for i in self.window_indices(len(frames), frame_idx):
            f = frames[i]
            a = f.astype(np.float32) / 255.0 if f.dtype == np.uint8 else f.astype(np.float32)
            planes.append(torch.from_numpy(a).permute(2, 0, 1))
        return torch.stack(planes).unsqueeze(0).to(self.device)

    # -- flow -------------------------------------------------------------------------------
    def _require_model(self):
        if not self.core.is_model_loaded():
            raise RuntimeError("Model not loaded. Call load_model() first.")

    def _flow_from_host_u8(self, frames, frame_idx):
        """The reference's host-array call (list of numpy frames in, numpy field out) without its per-call
        cost: the reference converts, stacks and uploads T float32 frames (124 MB at 1080p) for every
        field and T-1 of them are the frames of the previous call (:150-161).  Here every uint8 frame of
        the window is identified by a 128-bit hash of its CONTENT (3 ms for five 1080p frames; an in-place
        edit of a frame changes its key), uploaded once as uint8, and the engine's per-frame / per-pair
        caches are keyed on the same hash - so a sliding loop over host frames runs at the resident
        rate.  Same pixels, same /255, same network: bit-identical to the float path (tests).  Returns
        None when the fast path does not apply (float frames, sizes that need padding, CPU)."""
        model = self.core.model
        if not (str(self.device).startswith('cuda') and hasattr(model, "forward_u8")):
            return None
        ids = self.window_indices(len(frames), frame_idx)
        f0 = frames[ids[0]]
        if not all(isinstance(frames[i], np.ndarray) and frames[i].dtype == np.uint8 and frames[i].ndim == 3 and
                   frames[i].shape == f0.shape for i in set(ids)):
            return None
        H, W, C = f0.shape
        if C != 3 or H % 8 or W % 8:
            return None
        keys = {}
        for i in set(ids):
            a = np.ascontiguousarray(frames[i])
            k = ("host-u8", _digest(memoryview(a).cast("B")), a.shape)
            keys[i] = k
            if k in self._dev_frames:
                self._dev_frames.move_to_end(k)
            else:
                self._dev_frames[k] = torch.from_numpy(a).to(self.device)
                while len(self._dev_frames) > HOST_FRAME_CACHE:
                    self._dev_frames.popitem(last=False)
        win = torch.stack([self._dev_frames[keys[i]] for i in ids])
        flows, _ = model.forward_u8(win, return_lowres=False, frame_keys=[keys[i] for i in ids], pick_only=self.PICK_ONLY)
        return flows[0, flows.shape[1] // 2].permute(1, 2, 0).cpu().numpy()

    def compute_optical_flow(self, frames, frame_idx):
        """-> numpy [H,W,2] float32, pixels."""
        self._require_model()
        fast = self._flow_from_host_u8(frames, frame_idx)
        if fast is not None:
            return fast
        flow = self.core.compute_flow_from_tensor(self.prepare_frame_sequence(frames, frame_idx))
        return flow.permute(1, 2, 0).cpu().numpy()

    def compute_optical_flow_with_progress(self, frames, frame_idx, tile_pbar=None):
        self._require_model()

        def step(desc, n=0, reset=False):
            if tile_pbar is not None:
                tile_pbar.set_description(desc)
                if reset:
                    tile_pbar.reset()
                if n:
                    tile_pbar.update(n)

        step("Preparing frames", reset=True)
        fast = self._flow_from_host_u8(frames, frame_idx)
        if fast is not None:
            step("Completed", 4)
            return fast
        batch = self.prepare_frame_sequence(frames, frame_idx)
        step("Running VideoFlow", 2)
        flow = self.core.compute_flow_from_tensor(batch)
        step("Processing output", 1)
        out = flow.permute(1, 2, 0).cpu().numpy()
        step("Completed", 1)
        return out

    def compute_optical_flow_tiled(self, frames, frame_idx, tile_pbar=None, overall_pbar=None):
        if not self.tile_mode:
            return self.compute_optical_flow(frames, frame_idx)
        height, width = frames[frame_idx].shape[:2]
        tiles = self.calculate_tile_grid(width, height)[4]
        full = np.zeros((height, width, 2), dtype=np.float32)
        for k, t in enumerate(tiles):
            if overall_pbar is not None:
                overall_pbar.set_description(f"Tile {k + 1}/{len(tiles)} ({t['width']}x{t['height']})")
            crops = [self.extract_tile(f, t) for f in frames]
            full[t['y']:t['y'] + t['height'], t['x']:t['x'] + t['width']] = \
                self.compute_optical_flow_with_progress(crops, frame_idx, tile_pbar)
            if overall_pbar is not None:
                overall_pbar.update(1)
        return full

    # -- HBM-resident clips (extension; results identical to compute_optical_flow) ------------
    def upload_clip(self, frames):
        """list of uint8 [H,W,3] frames -> one uint8 [F,H,W,3] tensor on the device (uploaded once;
        the reference re-uploads T float32 frames per field, 4x the bytes and T-1 of them repeats)."""
        if any(f.dtype != np.uint8 for f in frames):
            raise ValueError("upload_clip expects uint8 frames")
        from vfml.clip_id import clip_token
        host = torch.from_numpy(np.stack(frames))
        if str(self.device).startswith('cuda'):
            host = host.pin_memory()
        clip = host.to(self.device, non_blocking=True)
        clip_token(clip)          # a fresh identity per upload: cache keys never follow a recycled address
        return clip

    def compute_optical_flow_resident(self, clip, frame_idx, tile=None):
        """Flow field of `frame_idx` from a clip already in HBM -> device tensor [H,W,2] float32.
        Same window, same /255, same network and index pick as compute_optical_flow; the u8->float
        conversion runs inside the engine's first kernel."""
        self._require_model()
        from vfml.clip_id import clip_token
        from vfml.network import take_frames
        frame_ids = self.window_indices(clip.shape[0], frame_idx)
        win = take_frames(clip, frame_ids)        # a view in the steady state: no index upload, no sync
        rect = None
        if tile is not None:
            rect = (tile['x'], tile['y'], tile['width'], tile['height'])
            win = win[:, tile['y']:tile['y'] + tile['height'], tile['x']:tile['x'] + tile['width']]
        H, W = win.shape[1:3]
        model = self.core.model
        if H % 8 == 0 and W % 8 == 0 and hasattr(model, "forward_u8"):
            # a frame of this clip (and tile) is the same pixels in every window that contains it:
            # let the engine reuse its per-frame encoder outputs across the sliding windows
            tok = clip_token(clip)
            keys = [(tok, i, rect) for i in frame_ids]
            flows, _ = model.forward_u8(win, return_lowres=False, frame_keys=keys, pick_only=self.PICK_ONLY)
            # the frame after this one is what a job asks for next: its window's new frame goes through the encoders
            # now, on a side stream beside this field's update iterations (vfml/network.py prefetch_frames)
            # (whole frames, or a tile when the caller walks a tile's frames before the next tile - vfml/runner.py sets
            # tiles_in_frame_order; the reference's own per-frame tile loop visits the other tiles of THIS frame next, and by
            # the time it is back at this tile the engine's per-frame cache has turned over)
            if ((tile is None or getattr(self, "tiles_in_frame_order", False)) and frame_idx + 1 < clip.shape[0]
                    and hasattr(model, "prefetch_frames")):
                nxt = self.window_indices(clip.shape[0], frame_idx + 1)
                # (consecutive frames only: such a window is a VIEW of the clip - a gathered one would be produced on this
                # stream, behind the iterations the prefetch is meant to run beside)
                if (nxt != frame_ids and nxt == list(range(nxt[0], nxt[0] + len(nxt)))
                        and max(nxt) <= getattr(clip, "_vfml_frames_ready", clip.shape[0]) - 1):
                    nwin = take_frames(clip, nxt)
                    if tile is not None:
                        nwin = nwin[:, tile['y']:tile['y'] + tile['height'], tile['x']:tile['x'] + tile['width']]
                    model.prefetch_frames(nwin, [(tok, i, rect) for i in nxt])
            return flows[0, flows.shape[1] // 2].permute(1, 2, 0)
        batch = (win.float() / 255.0).permute(0, 3, 1, 2).unsqueeze(0)
        return self.core.compute_flow_from_tensor(batch).permute(1, 2, 0)

    def compute_optical_flow_resident_batch(self, clip, frame_idxs):
        """Fields of several frames of a clip already in HBM -> list of device tensors [H,W,2].
        For the tri-frame network (`--vf-architecture bof`) runs of consecutive frames whose centre triples are
        (j-1, j, j+1) go through the engine in one pass (forward_u8(..., tri_batch=True)): at 720p a single
        triple keeps a third of the chip busy.  Everything else falls back to one call per frame.  Same
        fields as compute_optical_flow_resident, bit for bit."""
        self._require_model()
        from vfml.clip_id import clip_token
        from vfml.network import take_frames
        model = self.core.model
        frame_idxs = list(frame_idxs)
        F, H, W = clip.shape[0], clip.shape[1], clip.shape[2]
        can_batch = getattr(model, "tri_frame", False) and hasattr(model, "forward_u8") and H % 8 == 0 and W % 8 == 0
        lo = self.sequence_length // 2 - 1
        triples = [self.window_indices(F, i)[lo:lo + 3] if self.sequence_length >= 3 else None for i in frame_idxs]
        out, k = [], 0
        while k < len(frame_idxs):
            e = k
            if can_batch and triples[k] is not None and triples[k][0] + 1 == triples[k][1] == triples[k][2] - 1:
                while (e + 1 < len(frame_idxs) and e + 1 - k < self.TRI_BATCH and
                       triples[e + 1] == [t + 1 for t in triples[e]]):
                    e += 1
            if e == k:
                out.append(self.compute_optical_flow_resident(clip, frame_idxs[k]))
            else:
                B = e - k + 1
                ids = list(range(triples[k][0], triples[e][2] + 1))              # B + 2 consecutive frames
                tok = clip_token(clip)
                keys = [(tok, i, None) for i in ids]
                flows, _ = model.forward_u8(take_frames(clip, ids), return_lowres=False, frame_keys=keys, tri_batch=True)
                out.extend(flows[0, B + j].permute(1, 2, 0) for j in range(B))
            k = e + 1
        return out

    TRI_BATCH = int(os.environ.get("VFML_TRI_BATCH", "8"))        # fields per pass of the tri-frame network
    # The reference keeps ONE of the model's 2(T-2) flows per call (`flow_predictions[0, shape[1]//2]`,
    # processing/videoflow_core.py:194-195).  By default the engine is asked for that flow only and drops, in
    # the last iterations, the centre frames outside its dependency cone (bit-identical for that flow);
    # VFML_FULL_OUTPUT=1 (bench.py --full-output) computes all of them as `model(x, {})` does.
    PICK_ONLY = not os.environ.get("VFML_FULL_OUTPUT")

    # -- misc -------------------------------------------------------------------------------
    def is_model_loaded(self):
        return self.core.is_model_loaded()

    def get_model_info(self):
        info = self.core.get_model_info()
        if info["status"] == "loaded":
            info.update(tile_mode=self.tile_mode, sequence_length=self.sequence_length,
                        processor_type="VideoFlowProcessor")
        return info

    def get_memory_usage(self):
        return self.core.get_memory_usage()

    def validate_frames(self, frames, frame_idx):
        """ValueError on anything the reference rejects (:307-351)."""
        if not isinstance(frames, list):
            raise ValueError("Frames must be a list of numpy arrays")
        if not frames:
            raise ValueError("Frames list cannot be empty")
        if not 0 <= frame_idx < len(frames):
            raise ValueError(f"Frame index {frame_idx} out of range [0, {len(frames)-1}]")
        f = frames[0]
        if not isinstance(f, np.ndarray):
            raise ValueError("Frames must be numpy arrays")
        if f.ndim != 3:
            raise ValueError(f"Frames must be 3D arrays [H,W,C], got shape {f.shape}")
        if f.shape[2] != 3:
            raise ValueError(f"Frames must have 3 color channels, got {f.shape[2]}")
        if f.dtype not in (np.uint8, np.float32, np.float64):
            raise ValueError(f"Unsupported frame dtype: {f.dtype}")
        if f.dtype != np.uint8:
            lo, hi = float(f.min()), float(f.max())
            if (hi > 1.0 or lo < 0.0) and not (hi <= 255.0 and lo >= 0.0):
                raise ValueError("Float frames must be in range [0.0, 1.0] or [0.0, 255.0]")

    def set_tile_mode(self, enabled):
        self.tile_mode = enabled

    def set_sequence_length(self, length):
        if length < 1 or length > 10:
            raise ValueError("Sequence length must be between 1 and 10")
        self.sequence_length = length
