"""MemFlowCore — tensor-in / tensor-out MemFlow engine.

API mirror of reference processing/memflow_core.py:28-206 (constructor :40, _validate_device :62-78,
load_model :80-131, validate_input_tensor :133-168, compute_flow_from_tensor :170-196).  The reference
spawns a Python process per field that rebuilds the network from the (absent) MemFlow submodule and
exchanges tensors through files (memflow_inference_isolated.py); that isolation only existed to keep
two submodules' `core`/`utils` packages apart and has no equivalent here: the network is built once,
in process, on the HIP engine.  What the isolated script does to the data is kept (its lines :80-112):
value-range normalisation, InputPadder, LAST TWO frames, one step with an empty memory, unpad, CPU.
"""
import os
from typing import Dict

import torch

from vfml import InputPadder
from vfml.memflow_net import build_memflow_network, memflow_cfg
from vfml.weights import load_checked


class MemFlowCore:
    def __init__(self, device='cuda', model_path='MemFlow_ckpt/MemFlowNet_sintel.pth', stage='sintel'):
        self.device = self._validate_device(device)
        self.model_path = model_path
        self.stage = stage
        self.model = None
        self.cfg = None
        self.input_padder = None
        print("[MemFlow] Core engine initialized:")
        print(f"  Device: {self.device}\n  Model: {model_path}\n  Stage: {stage}")

    def _validate_device(self, device):
        if device == 'auto':
            device = 'cuda' if torch.cuda.is_available() else 'cpu'
        if device.startswith('cuda'):
            if not torch.cuda.is_available():
                print("Warning: CUDA requested but not available, falling back to CPU")
                return 'cpu'
            return 'cuda:0' if device == 'cuda' else device
        return device

    def load_model(self):
        if self.model is not None:
            print("MemFlow model already loaded")
            return
        if not os.path.exists(self.model_path):
            raise FileNotFoundError(f"MemFlow model not found: {self.model_path}")
        print(f"[Model] Loading MemFlow model from: {self.model_path}")
        try:
            cfg = memflow_cfg()
            cfg.restore_ckpt = self.model_path
            model = build_memflow_network(cfg)
            state = torch.load(self.model_path, map_location=self.device, weights_only=True)
            if any(k.startswith('module.') for k in state):
                state = {k.replace('module.', ''): v for k, v in state.items()}
            load_checked(model, state, self.model_path)
            self.model = model.to(self.device).eval()
            self.cfg = cfg
        except Exception as e:
            raise RuntimeError(f"Failed to load MemFlow model: {e}")
        print(f"[Model] MemFlow model loaded successfully:\n  Path: {self.model_path}\n  Stage: {self.stage}\n"
              f"  Device: {self.device}\n  Inference: in process (HIP engine)")

    def validate_input_tensor(self, tensor):
        if not isinstance(tensor, torch.Tensor):
            raise TypeError("Input must be a torch.Tensor")
        if tensor.dim() != 5:
            raise ValueError(f"Input tensor must be 5D [B, T, C, H, W], got shape: {tensor.shape}")
        B, T, C, H, W = tensor.shape
        if B != 1:
            raise ValueError(f"Batch size must be 1, got: {B}")
        if T < 2:
            raise ValueError(f"Temporal dimension must be at least 2, got: {T}")
        if C != 3:
            raise ValueError(f"Channel dimension must be 3 (RGB), got: {C}")
        if H < 64 or W < 64:
            raise ValueError(f"Spatial dimensions must be at least 64x64, got: {H}x{W}")
        have, want = str(tensor.device), self.device
        same = have == want or {have, want} == {'cuda', 'cuda:0'}
        if not same:
            print(f"Warning: Tensor device ({have}) differs from model device ({want})")

    @staticmethod
    def normalise(frames, return_mode=False):
        """Value-range heuristic of the reference's inference script (:81-85); with return_mode also which of
        its three branches was taken (part of a frame's cache identity: the same pixels normalised
        differently are different network inputs)."""
        mx = frames.max().item()
        if mx > 2.0:
            out, mode = 2 * (frames / 255.0) - 1.0, 0
        elif mx > 1.0:
            out, mode = 2 * frames - 1.0, 1
        else:
            out, mode = frames, 2
        return (out, mode) if return_mode else out

    @staticmethod
    def normalise_as(frames, mode):
        """The branch `mode` of normalise() (0: 0..255 values, 1: 0..2, 2: already normalised), no host sync."""
        if mode == 0:
            return 2 * (frames / 255.0) - 1.0
        if mode == 1:
            return 2 * frames - 1.0
        return frames

    def compute_pair_flows(self, frames_tensor, mode, frame_keys=None):
        """[1,B+1,3,H,W] raw frame values on the device, B consecutive pairs (k, k+1), normalised by branch `mode`
        of the range heuristic (the caller knows each field's window maximum) -> flows [B,2,H,W] on the device.
        An extension for job loops: B fields per pass of the engine, same values as B single calls."""
        if self.model is None:
            raise RuntimeError("Model not loaded. Call load_model() first.")
        with torch.no_grad():
            x = self.normalise_as(frames_tensor.to(self.device).float(), mode)
            padder = InputPadder(x.shape)
            x = padder.pad(x)
            keys = None
            if frame_keys is not None:
                keys = [(k, mode, tuple(x.shape[-2:])) for k in frame_keys]
            _, flows = self.model.forward_pairs(x, None, frame_keys=keys)
            return padder.unpad(flows)

    def compute_flow_from_tensor(self, frames_tensor: torch.Tensor, keep_on_device: bool = False,
                                 frame_keys=None) -> torch.Tensor:
        """[1,T,3,H,W] (0..255, 0..1 or -1..1 floats, any device) -> flow [2,H,W] on the CPU
        (keep_on_device=True, an extension, skips the download for callers that gather on the GPU;
        frame_keys, an extension: one hashable id per frame of the window - the engine then keeps each
        frame's feature encoder output for the call in which that frame is the previous one)."""
        if self.model is None:
            raise RuntimeError("Model not loaded. Call load_model() first.")
        self.validate_input_tensor(frames_tensor)
        with torch.no_grad():
            x, mode = self.normalise(frames_tensor.to(self.device).float(), return_mode=True)
            padder = InputPadder(x.shape)
            x = padder.pad(x)
            if frame_keys is not None and hasattr(self.model, "_frame_features"):
                keys = [(k, mode, tuple(x.shape[-2:])) for k in list(frame_keys)[-2:]]
                _, flow = self.model(x[:, -2:], None, frame_keys=keys)
            else:
                _, flow = self.model(x[:, -2:])
            flow = padder.unpad(flow[0])
            return flow if keep_on_device else flow.cpu()

    def get_memory_usage(self) -> Dict[str, float]:
        if self.device.startswith('cuda'):
            gb = 1024 ** 3
            return {'allocated_gb': torch.cuda.memory_allocated(self.device) / gb,
                    'reserved_gb': torch.cuda.memory_reserved(self.device) / gb, 'device': self.device}
        return {'device': 'cpu', 'note': 'CPU memory tracking not available'}

    def cleanup(self):
        if self.device.startswith('cuda'):
            torch.cuda.empty_cache()
