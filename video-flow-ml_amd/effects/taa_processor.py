"""Temporal anti-aliasing with flow reprojection: the consumer of the flow fields in `--taa` jobs.

API mirror of reference effects/taa_processor.py (TAAProcessor :20-324, TAAComparisonProcessor :326-383,
apply_taa_effect :386-411): same methods, arguments, per-sequence history and result dtypes.  A frame that is a
device tensor is blended there by `vfml_taa_blend` (history stays in HBM; ~60 B of traffic per pixel) and the
result is a device tensor; numpy frames take the same steps on the host, as the reference does.  The two agree
with the reference's outputs (tests/golden/taa.npz) to ~1e-12 relative on the host and to the last ulps of exp()
on the device.
"""
from typing import Optional, Tuple

import numpy as np

try:
    import torch
except ImportError:          # pragma: no cover
    torch = None


def _on_gpu(x):
    return torch is not None and torch.is_tensor(x) and x.is_cuda


def _gather4(image, px, py, clamp_low):
    """Integer corners and fractional offsets of the sample points (px, py).  clamp_low: corners are pulled back so
    that the +1 neighbour exists (bilateral sampler); otherwise the +1 neighbour is clamped (bilinear sampler)."""
    h, w = image.shape[:2]
    x0 = np.floor(px).astype(int)
    y0 = np.floor(py).astype(int)
    if clamp_low:
        x0 = np.clip(x0, 0, w - 2)
        y0 = np.clip(y0, 0, h - 2)
        x1, y1 = x0 + 1, y0 + 1
    else:
        x1 = np.clip(np.minimum(x0 + 1, w - 1), 0, w - 1)     # the +1 neighbour of the UNclamped corner
        y1 = np.clip(np.minimum(y0 + 1, h - 1), 0, h - 1)
        x0 = np.clip(x0, 0, w - 1)
        y0 = np.clip(y0, 0, h - 1)
    return (x0, x1, y0, y1), px - x0, py - y0


class TAAProcessor:
    """Exponential moving average of a frame sequence, the history optionally reprojected along the flow."""

    def __init__(self, alpha: float = 0.1, bilateral_sigma_color: float = 25.0):
        self.alpha = alpha
        self.bilateral_sigma_color = bilateral_sigma_color
        self.history = {}

    # -- public steps ---------------------------------------------------------------------------------
    def apply_taa(self, current_frame, flow_pixels=None, previous_taa_frame=None, alpha: Optional[float] = None,
                  use_flow: bool = True, use_bilateral: bool = True, sequence_id: str = 'default'):
        """current_frame RGB 0..255 [H,W,3]; flow_pixels [H,W,2] pointing at the pixel's place in the history;
        history from `previous_taa_frame` or the sequence's stored one.  Returns (and stores) the new history."""
        alpha = self.alpha if alpha is None else alpha
        if previous_taa_frame is None:
            previous_taa_frame = self.history.get(sequence_id)
        gpu = _on_gpu(current_frame)
        if previous_taa_frame is None:
            result = current_frame.float() if gpu else current_frame.astype(np.float32)
        elif gpu:
            result = self._device_step(current_frame, flow_pixels, previous_taa_frame, alpha, use_flow, use_bilateral)
        else:
            current = current_frame.astype(np.float32)
            if not use_flow or flow_pixels is None:
                result = alpha * current + (1 - alpha) * previous_taa_frame
            else:
                result = self._apply_flow_based_taa(current, flow_pixels, previous_taa_frame, alpha, use_bilateral)
        self.history[sequence_id] = result
        return result

    def apply_simple_taa(self, current_frame, previous_taa_frame=None, alpha: Optional[float] = None,
                         sequence_id: str = 'simple'):
        return self.apply_taa(current_frame=current_frame, flow_pixels=None, previous_taa_frame=previous_taa_frame,
                              alpha=alpha, use_flow=False, use_bilateral=False, sequence_id=sequence_id)

    def reset_history(self, sequence_id: Optional[str] = None):
        if sequence_id is None:
            self.history.clear()
        else:
            self.history.pop(sequence_id, None)

    def get_history(self, sequence_id: str = 'default'):
        return self.history.get(sequence_id)

    def set_alpha(self, alpha: float):
        if not 0.0 <= alpha <= 1.0:
            raise ValueError("Alpha must be between 0.0 and 1.0")
        self.alpha = alpha

    # -- device ---------------------------------------------------------------------------------------
    def _device_step(self, current, flow, history, alpha, use_flow, use_bilateral):
        from vfml import hip
        if not _on_gpu(history):
            history = torch.as_tensor(history).to(current.device)
        if current.dtype not in (torch.uint8, torch.float32):
            current = current.float()
        if not use_flow or flow is None:
            return hip.taa_blend(current, None, history, hip.TAA_SIMPLE, alpha)
        if not _on_gpu(flow):
            flow = torch.as_tensor(np.asarray(flow, dtype=np.float32)).to(current.device)
        mode = hip.TAA_BILATERAL if use_bilateral else hip.TAA_BILINEAR
        return hip.taa_blend(current, flow.float(), history, mode, alpha, self.bilateral_sigma_color)

    # -- host -----------------------------------------------------------------------------------------
    def _apply_flow_based_taa(self, current_frame, flow_pixels, previous_taa_frame, alpha, use_bilateral):
        h, w = current_frame.shape[:2]
        gy, gx = np.mgrid[0:h, 0:w]
        px = np.clip(np.nan_to_num(gx + flow_pixels[:, :, 0], nan=0.0, posinf=w - 1, neginf=0.0), 0, w - 1)
        py = np.clip(np.nan_to_num(gy + flow_pixels[:, :, 1], nan=0.0, posinf=h - 1, neginf=0.0), 0, h - 1)
        if use_bilateral:
            reprojected = self._bilateral_reprojection_sample(previous_taa_frame, px, py, current_frame)
        else:
            reprojected = self._bilinear_sample(previous_taa_frame, px, py)
        return alpha * current_frame + (1 - alpha) * reprojected

    def _bilateral_reprojection_sample(self, image, x_coords, y_coords, current_frame):
        """Four-neighbour average of `image` at the sample points: bilinear weights times a Gaussian of the luminance
        difference to the current frame, renormalised."""
        (x0, x1, y0, y1), fx, fy = _gather4(image, x_coords, y_coords, clamp_low=True)
        fx, fy = fx[..., None], fy[..., None]
        lum = np.mean(current_frame, axis=2)
        spread = 2 * (self.bilateral_sigma_color ** 2 * 0.1) + 1e-6
        acc, total = None, None
        for (yy, xx, sw) in ((y0, x0, (1 - fx) * (1 - fy)), (y0, x1, fx * (1 - fy)),
                             (y1, x0, (1 - fx) * fy), (y1, x1, fx * fy)):
            tap = image[yy, xx]
            diff = lum - np.mean(tap, axis=2)
            wgt = sw * np.exp(-diff ** 2 / spread)[..., None]
            acc = tap * wgt if acc is None else acc + tap * wgt
            total = wgt if total is None else total + wgt
        return acc / np.where(total == 0, 1e-6, total)

    def _bilinear_sample(self, image, x_coords, y_coords):
        (x0, x1, y0, y1), fx, fy = _gather4(image, x_coords, y_coords, clamp_low=False)
        out = np.zeros_like(image, dtype=np.float32)
        for c in range(image.shape[2]):
            out[:, :, c] = (image[y0, x0, c] * (1 - fx) * (1 - fy) + image[y0, x1, c] * fx * (1 - fy) +
                            image[y1, x0, c] * (1 - fx) * fy + image[y1, x1, c] * fx * fy)
        return out


class TAAComparisonProcessor:
    """Flow-reprojected and plain temporal blending of the same sequence, side by side."""

    def __init__(self, alpha: float = 0.1):
        self.flow_taa = TAAProcessor(alpha)
        self.simple_taa = TAAProcessor(alpha)

    def apply_comparison(self, current_frame, flow_pixels=None, alpha: Optional[float] = None) -> Tuple:
        with_flow = self.flow_taa.apply_taa(current_frame=current_frame, flow_pixels=flow_pixels, alpha=alpha,
                                            use_flow=True, use_bilateral=True, sequence_id='flow')
        plain = self.simple_taa.apply_simple_taa(current_frame=current_frame, alpha=alpha, sequence_id='simple')
        return with_flow, plain

    def reset_history(self):
        self.flow_taa.reset_history()
        self.simple_taa.reset_history()

    def set_alpha(self, alpha: float):
        self.flow_taa.set_alpha(alpha)
        self.simple_taa.set_alpha(alpha)


def apply_taa_effect(current_frame, flow_pixels=None, previous_taa_frame=None, alpha: float = 0.1,
                     use_flow: bool = True):
    return TAAProcessor(alpha).apply_taa(current_frame=current_frame, flow_pixels=flow_pixels,
                                         previous_taa_frame=previous_taa_frame, alpha=alpha, use_flow=use_flow)
