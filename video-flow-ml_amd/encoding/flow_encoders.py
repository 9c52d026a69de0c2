"""Flow fields -> 8-bit images: the gamedev and motion-vector codes of the reference.

API mirror of reference encoding/flow_encoders.py for the three formats that are pure arithmetic on the
field: `gamedev` (:70-117), `motion-vectors-rg8` (:120-190) and `motion-vectors-rgb8` in its 'rgb+' variant
(the reference's module-level `method`, :10, :242-293, decode :336-343).  Outputs are bytes, so they are
reproduced bit for bit (tests/golden/flow_encoders.npz, cut from the reference).  A field that is still a
device tensor is encoded there by `vfml_flow_encode` (16.6 MB read, 6.2 MB written per 1080p field) and
returned as a device uint8 tensor; numpy input takes the same float32 steps on the host.
`hsv` and `torchvision` need OpenCV / torchvision colour wheels and are not part of this build.
"""
from abc import ABC, abstractmethod

import numpy as np

try:
    import torch
except ImportError:          # pragma: no cover
    torch = None


def _on_gpu(flow):
    return torch is not None and torch.is_tensor(flow) and flow.is_cuda


def _to_u8(img):
    """rgb in [0,1] (NaN / inf allowed) -> uint8, as the reference finishes every encoder."""
    x = img * 255
    return np.nan_to_num(x, nan=0.0, posinf=255.0, neginf=0.0).astype(np.uint8)


def _clamp_to_unit(v, clamp_range):
    """clip to +-clamp_range, map to [0,1], clip again (the shared tail of gamedev and rg8)."""
    e = (np.clip(v, -clamp_range, clamp_range) + clamp_range) / (2 * clamp_range)
    return np.clip(e, 0, 1)


def _rg_image(e):
    rgb = np.zeros(e.shape[:2] + (3,), dtype=np.float32)
    rgb[:, :, :2] = e
    return _to_u8(rgb)


class FlowEncoder(ABC):
    @abstractmethod
    def encode(self, flow, width: int, height: int):
        """flow [H,W,2] float32 (numpy, or a device tensor) -> RGB [H,W,3] uint8 (same kind)."""


class GamedevFlowEncoder(FlowEncoder):
    """R, G = flow / (width, height) * scale_factor, clamped to +-clamp_range and mapped to [0, 255]; B = 0."""

    def __init__(self, scale_factor: float = 200.0, clamp_range: float = 20.0):
        self.scale_factor = scale_factor
        self.clamp_range = clamp_range

    def encode(self, flow, width: int, height: int):
        if _on_gpu(flow):
            from vfml import hip
            return hip.flow_encode(flow, hip.ENCODE_GAMEDEV, self.clamp_range, width, height, self.scale_factor)
        n = np.array(flow, dtype=np.float32, copy=True)
        n[:, :, 0] /= width
        n[:, :, 1] /= height
        n *= self.scale_factor
        return _rg_image(_clamp_to_unit(n, self.clamp_range))


class MotionVectorsRG8FlowEncoder(FlowEncoder):
    """R, G = flow clamped to +-clamp_range pixels, UNORM 8; B = 0."""

    def __init__(self, clamp_range: float = 64.0):
        self.clamp_range = clamp_range

    def encode(self, flow, width: int, height: int):
        if _on_gpu(flow):
            from vfml import hip
            return hip.flow_encode(flow, hip.ENCODE_RG8, self.clamp_range)
        return _rg_image(_clamp_to_unit(np.asarray(flow, dtype=np.float32), self.clamp_range))

    def decode(self, encoded_flow: np.ndarray) -> np.ndarray:
        rg = encoded_flow.astype(np.float32)[:, :, :2] / 255.0
        return (rg * 2 * self.clamp_range) - self.clamp_range


class MotionVectorsRGB8FlowEncoder(FlowEncoder):
    """'rgb+' code: d = flow / clamp_range shortened to the unit disc; R, G = (d + 1) / 2; B = sqrt(1 - |d|^2)."""

    def __init__(self, clamp_range: float = 32.0):
        self.clamp_range = clamp_range

    def encode(self, flow, width: int, height: int):
        if _on_gpu(flow):
            from vfml import hip
            return hip.flow_encode(flow, hip.ENCODE_RGB8, self.clamp_range)
        f = np.asarray(flow, dtype=np.float32)
        with np.errstate(all="ignore"):
            d = f / self.clamp_range                       # (a fresh array: the steps below write into it)
            dx, dy = d[:, :, 0], d[:, :, 1]
            length = np.sqrt(dx ** 2 + dy ** 2)
            far = length > 1
            dx[far] = dx[far] / length[far]
            dy[far] = dy[far] / length[far]
            rgb = np.zeros(f.shape[:2] + (3,), dtype=np.float32)
            rgb[:, :, 2] = np.sqrt(1 - dx ** 2 - dy ** 2)
            rgb[:, :, 0] = (np.clip(dx, -1, 1) + 1) / 2
            rgb[:, :, 1] = (np.clip(dy, -1, 1) + 1) / 2
            return _to_u8(rgb)

    def decode(self, encoded_flow: np.ndarray) -> np.ndarray:
        n = encoded_flow.astype(np.float32) / 255.0
        with np.errstate(all="ignore"):
            dx, dy = n[:, :, 0] * 2 - 1, n[:, :, 1] * 2 - 1
            magnitude = 1 / np.sqrt(dx ** 2 + dy ** 2 + n[:, :, 2] ** 2) * self.clamp_range
            out = np.zeros(encoded_flow.shape[:2] + (2,), dtype=np.float32)
            out[:, :, 0] = dx * magnitude
            out[:, :, 1] = dy * magnitude
        return out


class FlowEncoderFactory:
    _encoders = {'gamedev': GamedevFlowEncoder, 'motion-vectors-rg8': MotionVectorsRG8FlowEncoder,
                 'motion-vectors-rgb8': MotionVectorsRGB8FlowEncoder}
    _not_built = ('hsv', 'torchvision')

    @classmethod
    def create_encoder(cls, format_name: str, **kwargs) -> FlowEncoder:
        format_name = format_name.lower()
        if format_name in cls._not_built:
            raise ValueError(f"Format '{format_name}' needs OpenCV / torchvision and is not part of this build. "
                             f"Available formats: {', '.join(cls._encoders)}")
        if format_name not in cls._encoders:
            raise ValueError(f"Unsupported format '{format_name}'. Available formats: {', '.join(cls._encoders)}")
        return cls._encoders[format_name](**kwargs)

    @classmethod
    def get_available_formats(cls):
        return list(cls._encoders)

    @classmethod
    def register_encoder(cls, format_name: str, encoder_class: type):
        if not issubclass(encoder_class, FlowEncoder):
            raise ValueError("Encoder class must inherit from FlowEncoder")
        cls._encoders[format_name.lower()] = encoder_class


def encode_flow(flow, width: int, height: int, format_name: str = 'gamedev'):
    return FlowEncoderFactory.create_encoder(format_name).encode(flow, width, height)


def encode_motion_vectors(flow, clamp_range: float = 64.0, format_variant: str = 'rgb8'):
    enc = (MotionVectorsRG8FlowEncoder if format_variant.lower() == 'rg8' else MotionVectorsRGB8FlowEncoder)(clamp_range=clamp_range)
    h, w = flow.shape[:2]
    return enc.encode(flow, w, h)


def decode_motion_vectors(encoded_flow: np.ndarray, clamp_range: float = 64.0, format_variant: str = 'rgb8') -> np.ndarray:
    enc = (MotionVectorsRG8FlowEncoder if format_variant.lower() == 'rg8' else MotionVectorsRGB8FlowEncoder)(clamp_range=clamp_range)
    return enc.decode(encoded_flow)
