"""The engine's kernels as PyTorch custom ops: `torch.ops.vfml.*` (north_star: "hand-written HIP kernels through
PyTorch-ROCm custom ops").

The C ABI (include/vfml.h, bound in vfml/hip.py) is what the engine itself sequences - raw device pointers, channel
slices of wider buffers, fused epilogues.  This module registers the self-contained form of the hot ops with
`torch.library`, for callers that want tensors in / tensors out on the dispatcher (graph capture, `torch.compile`
custom-op boundaries, other PyTorch code):

    vfml::conv2d_nhwc(x, weight, bias, stride, pad_h, pad_w, activation) -> Tensor     K2 / K6 convolutions
    vfml::corr_volume(f1, f2, scale)                                     -> Tensor     K3 all-pairs correlation
    vfml::corr_lookup(pyramid, coords, hl, wl, radius)                   -> Tensor     K5 pyramid lookup
    vfml::convex_upsample(flow, mask)                                    -> Tensor     K8 8x convex upsampling

Only the CUDA (= HIP on ROCm) dispatch key has an implementation: on CPU tensors the dispatcher raises
NotImplementedError - there is no CPU fallback, as everywhere in the engine.  Importing this module registers the ops
(idempotent)."""
import torch

from . import hip
from .weights import pack_conv_weight

_LIB = None
_ACT = {"none": hip.EPI_NONE, "relu": hip.EPI_RELU, "tanh": hip.EPI_TANH, "sigmoid": hip.EPI_SIGMOID}


def _conv2d_nhwc(x, weight, bias, stride, pad_h, pad_w, activation):
    """x [n,h,w,cin] f32 (cin % 4 == 0), weight [cout,cin,kh,kw], bias [cout] or None -> [n,ho,wo,cout] f32, on the
    split-f16 MFMA kernel (fp32-grade: three f16 MFMAs per product)."""
    if activation not in _ACT:
        raise ValueError(f"activation must be one of {sorted(_ACT)}")
    n, h, w, cin = x.shape
    cout, cin_w, kh, kw = weight.shape
    if cin_w > cin or cin % 4:
        raise ValueError("x must carry the weight's input channels, padded to a multiple of 4")
    x = x.contiguous().float()
    flat = pack_conv_weight(weight, cin_pad=cin if cin > cin_w else None).to(x.device)
    sw = hip.SplitWeight(cout, flat.numel() // cout, x.device).fill(flat, scale=hip.SplitWeight.auto_scale(float(flat.abs().max())))
    ho = (h + 2 * pad_h - kh) // stride + 1
    wo = (w + 2 * pad_w - kw) // stride + 1
    ldo = (cout + 3) // 4 * 4
    out = torch.empty(n * ho * wo * ldo, device=x.device, dtype=torch.float32)
    hip.conv2d(x.reshape(-1), cin, cin, n, h, w, sw, None if bias is None else bias.contiguous().float(), cout, kh, kw,
               out, ldo, stride=stride, pad_h=pad_h, pad_w=pad_w, epilogue=_ACT[activation])
    return out.view(n, ho, wo, ldo)[..., :cout]


def _corr_volume(f1, f2, scale):
    """f1 [P,D], f2 [S,D] f32 (D % 32 == 0) -> scale * f1 f2^T, [P,S] f32 (a view of rows padded to 32 columns)."""
    P, D = f1.shape
    S = f2.shape[0]
    x16 = torch.empty(P * D, device=f1.device, dtype=torch.float32)
    hip.to_s16(f1.contiguous().float().reshape(-1), P, D, D, x16, D, scale=16.0)
    w = hip.SplitWeight(S, D, f1.device).fill(f2.contiguous().float().reshape(-1), scale=16.0)
    ld = (S + 31) // 32 * 32
    out = torch.empty(P * ld, device=f1.device, dtype=torch.float32)
    hip.conv2d(x16, D, D, 1, 1, P, w, None, S, 1, 1, out, ld, out_scale=float(scale) / 16.0, in_fmt=hip.FMT_S16)
    return out.view(P, ld)[:, :S]


def _corr_lookup(pyramid, coords, hl, wl, radius):
    """pyramid: per level a [P, ld_l] f32 tensor (row q = the correlation of query q with the hl[l] x wl[l] targets);
    coords [P,2] (x, y) at level 0 -> [P, levels * (2r+1)^2] bilinear window samples, RAFT's window order."""
    P = coords.shape[0]
    L = len(pyramid)
    ld = [int(p.shape[1]) for p in pyramid]
    nout = L * (2 * radius + 1) ** 2
    out = torch.empty(P * nout, device=coords.device, dtype=torch.float32)
    c = coords.contiguous().float()
    hip.corr_lookup([[p.contiguous().reshape(-1) for p in pyramid]], list(hl), list(wl), ld, radius, P, c.reshape(-1), 0, 2,
                    out, 0, nout)
    return out.view(P, nout)


def _convex_upsample(flow, mask):
    """flow [h,w,2] (pixels at 1/8 resolution), mask [h,w,576] logits (tap * 64 + sy * 8 + sx) -> [8h,8w,2] (x8)."""
    h, w, _ = flow.shape
    ys, xs = torch.meshgrid(torch.arange(h, device=flow.device, dtype=torch.float32),
                            torch.arange(w, device=flow.device, dtype=torch.float32), indexing="ij")
    coords = torch.zeros(h, w, 4, device=flow.device, dtype=torch.float32)
    coords[..., 0] = xs + flow[..., 0]
    coords[..., 1] = ys + flow[..., 1]
    out = torch.empty(8 * h, 8 * w, 2, device=flow.device, dtype=torch.float32)
    hip.convex_upsample(coords.reshape(-1), 0, 0, mask.contiguous().float().reshape(-1), 0, int(mask.shape[2]), h, w,
                        out.reshape(-1))
    return out


def register():
    global _LIB
    if _LIB is not None:
        return _LIB
    lib = torch.library.Library("vfml", "DEF")
    lib.define("conv2d_nhwc(Tensor x, Tensor weight, Tensor? bias, int stride, int pad_h, int pad_w, str activation) -> Tensor")
    lib.define("corr_volume(Tensor f1, Tensor f2, float scale) -> Tensor")
    lib.define("corr_lookup(Tensor[] pyramid, Tensor coords, int[] hl, int[] wl, int radius) -> Tensor")
    lib.define("convex_upsample(Tensor flow, Tensor mask) -> Tensor")
    lib.impl("conv2d_nhwc", _conv2d_nhwc, "CUDA")
    lib.impl("corr_volume", _corr_volume, "CUDA")
    lib.impl("corr_lookup", _corr_lookup, "CUDA")
    lib.impl("convex_upsample", _convex_upsample, "CUDA")
    _LIB = lib
    return lib


register()
