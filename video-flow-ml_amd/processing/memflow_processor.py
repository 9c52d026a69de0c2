"""MemFlowProcessor — frame windows and numpy<->tensor plumbing around MemFlowCore.

API mirror of reference processing/memflow_processor.py:20-256: the window is the `sequence_length`
frames ENDING at frame_idx, front-padded by repeating the first (:113-122); frames stay 0..255 floats on
the CPU (:124-139); MemFlow never tiles — the tile methods return one full-frame "tile" (:190-247)."""
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .memflow_core import MemFlowCore


class MemFlowProcessor:
    def __init__(self, device='cuda', model_path='MemFlow_ckpt/MemFlowNet_sintel.pth', stage='sintel',
                 sequence_length=3):
        self.device = device
        self.model_path = model_path
        self.stage = stage
        self.sequence_length = max(2, sequence_length)
        self.core_engine = MemFlowCore(device=device, model_path=model_path, stage=stage)
        print("MemFlow Processor initialized:")
        print(f"  Sequence length: {self.sequence_length} frames")
        print("  Note: MemFlow processes frame pairs (last 2 frames of sequence)")

    def load_model(self):
        self.core_engine.load_model()

    def validate_frame_sequence(self, frames: List[np.ndarray]) -> bool:
        if not frames:
            raise ValueError("Frame sequence is empty")
        if len(frames) < 2:
            raise ValueError(f"Need at least 2 frames for optical flow, got {len(frames)}")
        first = frames[0]
        if first.ndim != 3:
            raise ValueError(f"Frames must be 3D (H, W, C), got shape: {first.shape}")
        height, width, channels = first.shape
        if channels != 3:
            raise ValueError(f"Frames must have 3 channels (RGB), got: {channels}")
        if height < 64 or width < 64:
            raise ValueError(f"Frame dimensions must be at least 64x64, got: {height}x{width}")
        for i, f in enumerate(frames[1:], 1):
            if f.shape != first.shape:
                raise ValueError(f"Frame {i} shape {f.shape} differs from first frame {first.shape}")
        return True

    def window_indices(self, frame_idx):
        end = frame_idx + 1
        idx = list(range(max(0, end - self.sequence_length), end))
        while len(idx) < self.sequence_length:
            idx.insert(0, idx[0])
        return idx

    def prepare_frame_sequence(self, frames: List[np.ndarray], frame_idx: int) -> torch.Tensor:
        self.validate_frame_sequence(frames)
        planes = []
        for i in self.window_indices(frame_idx):
            f = frames[i]
            if f.dtype != np.uint8:
                f = np.clip(f, 0, 255).astype(np.uint8)
            planes.append(torch.from_numpy(f).permute(2, 0, 1).float())
        return torch.stack(planes, dim=0).unsqueeze(0)

    def compute_optical_flow(self, frames: List[np.ndarray], frame_idx: int) -> np.ndarray:
        flow = self.core_engine.compute_flow_from_tensor(self.prepare_frame_sequence(frames, frame_idx))
        return flow.permute(1, 2, 0).numpy()

    # -- HBM-resident clips (extension; same fields as compute_optical_flow) -------------------------
    def upload_clip(self, frames):
        from vfml.clip_id import clip_token
        host = torch.from_numpy(np.stack(frames))
        dev = self.core_engine.device
        if dev.startswith('cuda'):
            host = host.pin_memory()
        clip = host.to(dev, non_blocking=True)
        clip_token(clip)          # a fresh identity per upload (vfml/clip_id.py)
        return clip

    def compute_optical_flow_resident(self, clip, frame_idx, tile=None):
        """uint8 clip [F,H,W,3] on the device -> flow [H,W,2] on the device (tile is ignored: MemFlow
        works on full frames, reference :190-247)."""
        from vfml.clip_id import clip_token
        from vfml.network import take_frames
        ids = self.window_indices(frame_idx)
        x = take_frames(clip, ids).permute(0, 3, 1, 2).float().unsqueeze(0)
        tok = clip_token(clip)
        keys = [(tok, i) for i in ids]      # same clip, same frame = same pixels
        return self.core_engine.compute_flow_from_tensor(x, keep_on_device=True, frame_keys=keys).permute(1, 2, 0)

    PAIR_BATCH = int(os.environ.get("VFML_PAIR_BATCH", "3"))   # fields per pass of the engine in job loops

    def _frame_maxima(self, clip):
        """Largest value of every frame of a resident clip (one pass + one sync per clip): the range heuristic
        of the reference's script looks at the window's maximum."""
        from vfml.clip_id import clip_token
        fed = getattr(clip, "_vfml_frame_maxima", None)      # a clip fed frame by frame knows them from the host side
        if fed is not None:
            return fed
        key = (clip_token(clip), tuple(clip.shape))
        if getattr(self, "_maxima_key", None) != key:
            self._maxima = clip.reshape(clip.shape[0], -1).amax(dim=1).tolist()
            self._maxima_key = key
        return self._maxima

    def compute_optical_flow_resident_batch(self, clip, frame_idxs):
        """Fields of several frames of a resident clip -> list of device tensors [H,W,2].  Runs of consecutive
        frames (pairs (i-1, i), same branch of the range heuristic) go through the engine PAIR_BATCH at a time;
        frame 0 (pair (0, 0)) and anything irregular one by one.  Same fields as compute_optical_flow_resident."""
        from vfml.clip_id import clip_token
        from vfml.network import take_frames
        frame_idxs = list(frame_idxs)
        model = self.core_engine.model
        if not hasattr(model, "forward_pairs") or not clip.is_cuda:
            return [self.compute_optical_flow_resident(clip, i) for i in frame_idxs]
        mx = self._frame_maxima(clip)

        def mode_of(i):
            m = max(mx[j] for j in self.window_indices(i))
            return 0 if m > 2.0 else (1 if m > 1.0 else 2)

        out, k = [], 0
        while k < len(frame_idxs):
            i = frame_idxs[k]
            e = k
            if i >= 1:
                while (e + 1 < len(frame_idxs) and e + 1 - k < self.PAIR_BATCH and frame_idxs[e + 1] == frame_idxs[e] + 1
                       and mode_of(frame_idxs[e + 1]) == mode_of(i)):
                    e += 1
            if e == k:
                out.append(self.compute_optical_flow_resident(clip, i))
            else:
                ids = list(range(i - 1, frame_idxs[e] + 1))                      # B + 1 consecutive frames
                x = take_frames(clip, ids).permute(0, 3, 1, 2).float().unsqueeze(0)
                tok = clip_token(clip)
                keys = [(tok, j) for j in ids]
                flows = self.core_engine.compute_pair_flows(x, mode_of(i), frame_keys=keys)
                out.extend(flows[j].permute(1, 2, 0) for j in range(flows.shape[0]))
            k = e + 1
        return out

    def compute_optical_flow_with_progress(self, frames, frame_idx, tile_pbar=None) -> np.ndarray:
        if tile_pbar is not None:
            tile_pbar.set_description("MemFlow processing")
            tile_pbar.update(1)
        return self.compute_optical_flow(frames, frame_idx)

    def calculate_tile_grid(self, width: int, height: int, tile_size: int = 1280) -> Tuple:
        return width, height, 1, 1, [{'x': 0, 'y': 0, 'width': width, 'height': height, 'tile_idx': 0}]

    def extract_tile(self, frame: np.ndarray, tile_info: Dict[str, int]) -> np.ndarray:
        return frame

    def compute_optical_flow_tiled(self, frames, frame_idx, tile_pbar=None, overall_pbar=None) -> np.ndarray:
        for bar, text in ((tile_pbar, "MemFlow full-frame"), (overall_pbar, "MemFlow processing")):
            if bar is not None:
                bar.set_description(text)
                bar.reset(total=1)
                bar.update(1)
        return self.compute_optical_flow(frames, frame_idx)

    def get_core_engine(self) -> MemFlowCore:
        return self.core_engine

    def get_memory_usage(self) -> Dict[str, float]:
        return self.core_engine.get_memory_usage()

    def cleanup(self):
        self.core_engine.cleanup()
