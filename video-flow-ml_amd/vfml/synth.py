"""Deterministic synthetic clips: a blurred-noise background translating by
(+1.5, -0.75) px/frame with a textured disc moving (-3, +2) px/frame on top; uint8 RGB [H,W,3].
There is no network / dataset access, so every benchmark and test clip comes from here."""
import numpy as np

SEED = 20250829


def _texture(rng, h, w, k=9):
    t = rng.integers(0, 256, size=(h + k - 1, w + k - 1, 3)).astype(np.float32)
    c = np.cumsum(np.cumsum(np.pad(t, ((1, 0), (1, 0), (0, 0))), axis=0), axis=1)
    box = (c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]) / (k * k)
    lo, hi = box.min(), box.max()
    return ((box - lo) / (hi - lo) * 255.0).astype(np.uint8)


def synthetic_clip(num_frames, height, width, seed=SEED, start=0):
    """List of `num_frames` uint8 [height,width,3] frames: frames start .. start + num_frames - 1 of the clip (frame t
    depends on t alone, so a rank of a sharded job can generate its own stretch only)."""
    margin = 128
    bg = _texture(np.random.default_rng(seed), height + 2 * margin, width + 2 * margin)
    fg = _texture(np.random.default_rng(seed + 1), height, width)
    yy, xx = np.mgrid[0:height, 0:width]
    rad = height / 6.0
    frames = []
    for t in range(start, start + num_frames):
        oy = margin + int(round(-0.75 * (t % 120)))   # periodic so long clips stay inside the margin
        ox = margin + int(round(1.5 * (t % 60))) - 45
        f = bg[oy:oy + height, ox:ox + width].copy()
        cy = height / 2.0 + 2.0 * (t % 100) - 100
        cx = width / 2.0 - 3.0 * (t % 100) + 150
        disc = (yy - cy) ** 2 + (xx - cx) ** 2 <= rad * rad
        f[disc] = fg[disc]
        frames.append(np.ascontiguousarray(f))
    return frames
