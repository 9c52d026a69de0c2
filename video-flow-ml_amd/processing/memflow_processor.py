"""MemFlowProcessor — frame windows and numpy<->tensor plumbing around MemFlowCore.

API mirror of reference processing/memflow_processor.py:20-256: the window is the `sequence_length`
frames ENDING at frame_idx, front-padded by repeating the first (:113-122); frames stay 0..255 floats on
the CPU (:124-139); MemFlow never tiles — the tile methods return one full-frame "tile" (:190-247)."""
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
        host = torch.from_numpy(np.stack(frames))
        dev = self.core_engine.device
        if dev.startswith('cuda'):
            host = host.pin_memory()
        return host.to(dev, non_blocking=True)

    def compute_optical_flow_resident(self, clip, frame_idx, tile=None):
        """uint8 clip [F,H,W,3] on the device -> flow [H,W,2] on the device (tile is ignored: MemFlow
        works on full frames, reference :190-247)."""
        from vfml.network import take_frames
        ids = self.window_indices(frame_idx)
        x = take_frames(clip, ids).permute(0, 3, 1, 2).float().unsqueeze(0)
        keys = [(clip.data_ptr(), clip._version, i) for i in ids]      # same clip, same frame = same pixels
        return self.core_engine.compute_flow_from_tensor(x, keep_on_device=True, frame_keys=keys).permute(1, 2, 0)

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
