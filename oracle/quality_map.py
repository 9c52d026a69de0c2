"""ORACLE (test infrastructure, never imported by the product): numpy float32 restatement of the reference's flow
quality map, `generate_quality_frame_gpu` (reference correction_worker.py:175-208), torch's float32 operations
written out one by one.  Pinned by tests/golden/quality_map.npz, cut from the reference itself run on torch's
CPU device (tests/golden/make_quality_fixtures.py); tests/test_quality_map.py holds it to those bytes.  Where torch
fuses a multiply-add (bilinear resize, vector norm) this does too.  One step cannot be restated: on arrays above a
few thousand elements torch's CPU sqrt is MKL's vector sqrt, 1 ulp off the correctly rounded root on ~0.7 % of values;
it has not moved a byte yet (0 of 9.3 M bytes differ from the reference on 540x960 frames at full, half and odd LOD
resolutions, 0 on the fixtures), but the GPU parity bar allows for it: bytes within one level, < 1e-5 of them
different."""
import numpy as np

F = np.float32


def _fma(a, b, c):
    """float32 fused multiply-add: the float64 product of two float32 values is exact, so one float64 add and one
    rounding to float32 reproduce it (up to double rounding, never seen on these inputs)."""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(F)


def resize_flow(flow, h, w):
    """F.interpolate(mode='bilinear', align_corners=False) of an [fh,fw,2] field to [h,w,2], then the vector rescale
    (reference :183-185), as torch's CPU kernel evaluates it in this image (found by matching its output bit for bit
    on 1.5 M values): source index = fma(scale, dst + 0.5, -0.5) clamped at 0; rows first,
    t = fma(v0, 1 - l, v1 * l) along x, then the same along y."""
    fh, fw = flow.shape[:2]

    def taps(n_out, n_in):
        scale = F(n_in) / F(n_out)
        src = np.maximum(_fma(scale, np.arange(n_out, dtype=F) + F(0.5), F(-0.5)), F(0))
        i0 = np.minimum(np.floor(src).astype(np.int64), n_in - 1)
        i1 = i0 + (i0 < n_in - 1)
        l1 = np.clip((src - i0.astype(F)).astype(F), 0, 1)
        return i0, i1, F(1) - l1, l1

    y0, y1, hy0, hy1 = taps(h, fh)
    x0, x1, wx0, wx1 = taps(w, fw)
    f = flow.astype(F)
    hy0, hy1 = hy0[:, None, None], hy1[:, None, None]
    wx0, wx1 = wx0[None, :, None], wx1[None, :, None]
    top = _fma(f[y0][:, x0], wx0, f[y0][:, x1] * wx1)
    bot = _fma(f[y1][:, x0], wx0, f[y1][:, x1] * wx1)
    out = _fma(top, hy0, bot * hy1)
    out[..., 0] *= F(w / fw)
    out[..., 1] *= F(h / fh)
    return out


def quality_map(frame1, frame2, flow, threshold):
    h, w = frame1.shape[:2]
    a = frame1.astype(F) / F(255.0)
    b = frame2.astype(F) / F(255.0)
    flow = flow.astype(F)
    if flow.shape[:2] != (h, w):
        flow = resize_flow(flow, h, w)
    gy, gx = np.mgrid[0:h, 0:w]
    with np.errstate(all="ignore"):
        tx = gx.astype(F) - flow[..., 0]
        ty = gy.astype(F) - flow[..., 1]
        oob = (tx < 0) | (tx >= w) | (ty < 0) | (ty >= h)
        # .long(): truncation; NaN and out-of-range values become INT64_MIN on the host, i.e. 0 after the clamp
        xi = np.where(np.isfinite(tx) & (np.abs(tx) < 9e18), tx, -1).astype(np.int64).clip(0, w - 1)
        yi = np.where(np.isfinite(ty) & (np.abs(ty) < 9e18), ty, -1).astype(np.int64).clip(0, h - 1)
    s = b[yi, xi]
    d = a - s

    def sum3(v):
        return (v[..., 0] + v[..., 1]) + v[..., 2]

    rgb = F(1.0) - np.sqrt(sum3(d * d)) / F(1.732)
    absim = F(1.0) - sum3(np.abs(d)) / F(3)
    def norm3(v):          # linalg.vector_norm's accumulation: fused multiply-adds over the three channels
        return np.sqrt(_fma(v[..., 2], v[..., 2], _fma(v[..., 1], v[..., 1], v[..., 0] * v[..., 0])))

    na = np.maximum(norm3(a), F(1e-8))[..., None]
    ns = np.maximum(norm3(s), F(1e-8))[..., None]
    cos = (sum3((a / na) * (s / ns)) + F(1.0)) / F(2.0)
    overall = ((rgb + absim) + cos) / F(3.0)
    green = np.clip((overall - F(0.5)) * F(2.0), 0, 1)
    red = np.clip(F(1.0) - overall, 0, 1)
    good = overall > F(threshold)
    q = np.zeros((h, w, 3), dtype=F)
    q[..., 1] = np.where(good, green, 0)
    q[..., 0] = np.where(good, 0, red)
    q[oob] = (1.0, 0.0, 0.0)
    return (q * F(255)).astype(np.uint8)
