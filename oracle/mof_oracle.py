"""CPU oracle for the multi-frame optical-flow hot path (TEST INFRASTRUCTURE ONLY).

Status: PARITY UNPINNED for the model arithmetic.  The reference repository
(IvanPopov/video-flow-ml) holds no model code: everything behind
`self.model(frame_batch_padded, {})` (processing/videoflow_core.py:188) lives in the
un-vendored, un-pinned git submodule XiaoyuShi97/VideoFlow (.gitmodules:1-3, directory
empty, weights absent per .MISSING_LARGE_BLOBS:1-11) and the reference has no tests
or golden vectors.  This file therefore restates the *published* algorithm (RAFT
encoder / all-pairs correlation pyramid / lookup / convex upsampling, arranged the
way VideoFlow's MOFNet arranges them for N frames) in plain fp32 PyTorch on the CPU
and is pinned only by analytic known-answer tests (tests/test_oracle_kat.py).  The
host plumbing either side of the model (windows, tiles, .npz/.flo, cache names) IS
pinned by fixtures cut from the reference's own importable modules (tests/golden/).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The shipped engine (video-flow-ml_amd/vfml) never does.

Call-site anchors in the reference (what this restates):
  * build_network(cfg)(images[B,N,3,H,W], {}) -> (flow[B,2(N-2),2,H,W], aux)
        processing/videoflow_core.py:28,101,188
  * InputPadder(dims).pad / .unpad           processing/videoflow_core.py:29,182-183,191
  * get_cfg() attribute bag                   processing/videoflow_core.py:30,76,88-94
  * middle index pick flow[0, shape[1]//2]    processing/videoflow_core.py:194-195

Tensor layout here is PyTorch-native NCHW fp32; the engine uses NHWC on the device.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- cfg
def get_cfg():
    """Attribute bag with the fields the reference mutates
    (processing/videoflow_core.py:88-94: model, decoder_depth, corr_levels, corr_radius)."""
    return SimpleNamespace(
        model="",
        network="MOFNetStack",
        feat_dim=256,
        down_ratio=8,
        corr_levels=4,
        corr_radius=4,
        decoder_depth=12,
        # explicit input normalisation pair: net_in = input_scale * x + input_shift.
        # (2, -1) maps the [0,1] tensors the reference hands over
        # (processing/videoflow_processor.py:154) onto [-1,1].
        input_scale=2.0,
        input_shift=-1.0,
    )


# -------------------------------------------------------------------------- padder
class InputPadder:
    """Pads H,W up to a multiple of 8, split evenly (RAFT 'sintel' mode), replicate mode.
    Used at processing/videoflow_core.py:182-183,191 on a 5-D [B,T,C,H,W] tensor."""

    def __init__(self, dims, mode="sintel"):
        self.ht, self.wd = int(dims[-2]), int(dims[-1])
        pad_ht = (((self.ht // 8) + 1) * 8 - self.ht) % 8
        pad_wd = (((self.wd // 8) + 1) * 8 - self.wd) % 8
        if mode == "sintel":
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2]
        else:
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, 0, pad_ht]

    def pad(self, x):
        if not any(self._pad):
            return x
        lead = x.shape[:-3]
        y = F.pad(x.reshape(-1, *x.shape[-3:]), self._pad, mode="replicate")
        return y.reshape(*lead, *y.shape[-3:])

    def unpad(self, x):
        ht, wd = x.shape[-2:]
        c = [self._pad[2], ht - self._pad[3], self._pad[0], wd - self._pad[1]]
        return x[..., c[0]:c[1], c[2]:c[3]]


# ------------------------------------------------------------------------- encoder
class ResidualBlock(nn.Module):
    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, 3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1)
        self.norm1 = nn.InstanceNorm2d(planes)
        self.norm2 = nn.InstanceNorm2d(planes)
        self.downsample = None
        if stride != 1:
            self.norm3 = nn.InstanceNorm2d(planes)
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, 1, stride=stride), self.norm3)

    def forward(self, x):
        y = F.relu(self.norm1(self.conv1(x)))
        y = F.relu(self.norm2(self.conv2(y)))
        if self.downsample is not None:
            x = self.downsample(x)
        return F.relu(x + y)


class BasicEncoder(nn.Module):
    """RAFT residual encoder, instance norm, 1/8 resolution, `output_dim` channels."""

    def __init__(self, output_dim=256):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3)
        self.norm1 = nn.InstanceNorm2d(64)
        self.layer1 = nn.Sequential(ResidualBlock(64, 64, 1), ResidualBlock(64, 64, 1))
        self.layer2 = nn.Sequential(ResidualBlock(64, 96, 2), ResidualBlock(96, 96, 1))
        self.layer3 = nn.Sequential(ResidualBlock(96, 128, 2), ResidualBlock(128, 128, 1))
        self.conv2 = nn.Conv2d(128, output_dim, 1)

    def forward(self, x):
        x = F.relu(self.norm1(self.conv1(x)))
        x = self.layer3(self.layer2(self.layer1(x)))
        return self.conv2(x)


# --------------------------------------------------------------------- correlation
def coords_grid(batch, ht, wd, dtype=torch.float32):
    ys, xs = torch.meshgrid(torch.arange(ht), torch.arange(wd), indexing="ij")
    return torch.stack([xs, ys], dim=0).to(dtype)[None].repeat(batch, 1, 1, 1)


def bilinear_sampler(img, coords):
    """grid_sample with pixel coordinates, align_corners=True, zeros padding (RAFT)."""
    H, W = img.shape[-2:]
    xgrid, ygrid = coords.split([1, 1], dim=-1)
    xgrid = 2 * xgrid / (W - 1) - 1
    ygrid = 2 * ygrid / (H - 1) - 1
    return F.grid_sample(img, torch.cat([xgrid, ygrid], dim=-1), align_corners=True)


class CorrBlock:
    """All-pairs correlation volume + avg-pool pyramid + (2r+1)^2 window lookup.

    Window index order is RAFT's: delta = stack(meshgrid(dy, dx)) is added onto the
    (x, y) centroid, so output channel i*(2r+1)+j samples x+d[i], y+d[j]."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.num_levels, self.radius = num_levels, radius
        b, d, h, w = fmap1.shape
        corr = torch.matmul(fmap1.view(b, d, h * w).transpose(1, 2), fmap2.view(b, d, h * w))
        corr = (corr / math.sqrt(d)).reshape(b * h * w, 1, h, w)
        self.pyramid = [corr]
        for _ in range(num_levels - 1):
            corr = F.avg_pool2d(corr, 2, stride=2)
            self.pyramid.append(corr)

    def __call__(self, coords):
        r = self.radius
        coords = coords.permute(0, 2, 3, 1)
        b, h, w, _ = coords.shape
        d = torch.linspace(-r, r, 2 * r + 1, dtype=coords.dtype)
        delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1).view(1, 2 * r + 1, 2 * r + 1, 2)
        out = []
        for i, corr in enumerate(self.pyramid):
            centroid = coords.reshape(b * h * w, 1, 1, 2) / 2 ** i
            out.append(bilinear_sampler(corr, centroid + delta).view(b, h, w, -1))
        return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous()


# -------------------------------------------------------------------- update block
class MotionEncoder(nn.Module):
    """RAFT BasicMotionEncoder widened to (forward, backward) correlation and flow."""

    def __init__(self, cor_planes):
        super().__init__()
        self.convc1 = nn.Conv2d(2 * cor_planes, 256, 1)
        self.convc2 = nn.Conv2d(256, 192, 3, padding=1)
        self.convf1 = nn.Conv2d(4, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(192 + 64, 128 - 4, 3, padding=1)
        self.sel = None

    def forward(self, fflow, bflow, fcorr, bcorr):
        flow = torch.cat([fflow, bflow], dim=1)
        w = self.convc1.weight
        if self.sel is not None:    # --fast: a sub-window / sub-pyramid of the trained lookup's input columns
            half = w.shape[1] // 2
            w = torch.cat([w[:, self.sel], w[:, half + self.sel]], dim=1)
        cor = F.relu(F.conv2d(torch.cat([fcorr, bcorr], dim=1), w, self.convc1.bias))
        cor = F.relu(self.convc2(cor))
        flo = F.relu(self.convf1(flow))
        flo = F.relu(self.convf2(flo))
        out = F.relu(self.conv(torch.cat([cor, flo], dim=1)))
        return torch.cat([out, flow], dim=1)


class SepConvGRU(nn.Module):
    def __init__(self, hidden_dim=128, input_dim=384):
        super().__init__()
        c = hidden_dim + input_dim
        self.convz1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convr1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convq1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convz2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self.convr2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self.convq2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))

    def forward(self, h, x):
        for cz, cr, cq in ((self.convz1, self.convr1, self.convq1), (self.convz2, self.convr2, self.convq2)):
            hx = torch.cat([h, x], dim=1)
            z = torch.sigmoid(cz(hx))
            r = torch.sigmoid(cr(hx))
            q = torch.tanh(cq(torch.cat([r * h, x], dim=1)))
            h = (1 - z) * h + z * q
        return h


class FlowHead(nn.Module):
    def __init__(self, input_dim=128, hidden_dim=256, out_dim=4):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, out_dim, 3, padding=1)

    def forward(self, x):
        return self.conv2(F.relu(self.conv1(x)))


class MOFUpdateBlock(nn.Module):
    """Motion encoder -> temporal stack fusion over the centre frames -> SepConvGRU ->
    flow head (4 ch: d_fwd, d_bwd) and mask head (2 x 576 ch, x0.25)."""

    def __init__(self, cor_planes, hidden_dim=128):
        super().__init__()
        self.encoder = MotionEncoder(cor_planes)
        self.tprop = nn.Conv2d(3 * 128, 128, 1)
        self.gru = SepConvGRU(hidden_dim, input_dim=128 + 128 + 128)
        self.flow_head = FlowHead(hidden_dim, 256, 4)
        self.mask = nn.Sequential(nn.Conv2d(hidden_dim, 256, 3, padding=1), nn.ReLU(inplace=False),
                                  nn.Conv2d(256, 2 * 64 * 9, 1))

    def temporal(self, mf, bs):
        bm, c, h, w = mf.shape
        m = mf.view(bs, bm // bs, c, h, w)
        zero = torch.zeros_like(m[:, :1])
        prev = torch.cat([zero, m[:, :-1]], dim=1)
        nxt = torch.cat([m[:, 1:], zero], dim=1)
        return F.relu(self.tprop(torch.cat([prev, m, nxt], dim=2).view(bm, 3 * c, h, w)))

    def forward(self, net, inp, fcorr, bcorr, fflow, bflow, bs):
        mf = self.encoder(fflow, bflow, fcorr, bcorr)
        mt = self.temporal(mf, bs)
        net = self.gru(net, torch.cat([inp, mf, mt], dim=1))
        return net, 0.25 * self.mask(net), self.flow_head(net)


# ------------------------------------------------------------------------ network
def upsample_flow(flow, mask):
    """[N,2,h,w] -> [N,2,8h,8w] by convex combination of the 3x3 coarse neighbourhood."""
    n, _, h, w = flow.shape
    mask = torch.softmax(mask.view(n, 1, 9, 8, 8, h, w), dim=2)
    up = F.unfold(8 * flow, [3, 3], padding=1).view(n, 2, 9, 1, 1, h, w)
    up = torch.sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(n, 2, 8 * h, 8 * w)


class MOFNetOracle(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.hidden_dim = self.context_dim = cfg.feat_dim // 2
        self.fnet = BasicEncoder(cfg.feat_dim)
        self.cnet = BasicEncoder(cfg.feat_dim)
        # checkpoint shapes are those of the base 4-level, radius-4 lookup; a smaller configured lookup
        # (the reference's --fast, processing/videoflow_core.py:91-94) uses the matching input columns
        BL, BR = 4, 4
        self.update_block = MOFUpdateBlock(BL * (2 * BR + 1) ** 2, self.hidden_dim)
        if (cfg.corr_levels, cfg.corr_radius) != (BL, BR):
            bw, d = 2 * BR + 1, BR - cfg.corr_radius
            self.update_block.encoder.sel = torch.tensor(
                [l * bw * bw + (i + d) * bw + (j + d) for l in range(cfg.corr_levels)
                 for i in range(2 * cfg.corr_radius + 1) for j in range(2 * cfg.corr_radius + 1)])

    @torch.no_grad()
    def forward(self, images, data=None, return_lowres=False):
        cfg = self.cfg
        if getattr(cfg, "network", "MOFNetStack") == "BOFNet" and images.shape[1] > 3:
            lo = images.shape[1] // 2 - 1          # tri-frame member: centre triple of the window
            images = images[:, lo:lo + 3]
        B, N, _, H, W = images.shape
        M = N - 2
        h, w = H // 8, W // 8
        images = cfg.input_scale * images + cfg.input_shift
        fmaps = self.fnet(images.reshape(B * N, 3, H, W)).reshape(B, N, -1, h, w)
        centre = fmaps[:, 1:N - 1].reshape(B * M, -1, h, w)
        fcorr_fn = CorrBlock(centre, fmaps[:, 2:N].reshape(B * M, -1, h, w), cfg.corr_levels, cfg.corr_radius)
        bcorr_fn = CorrBlock(centre, fmaps[:, 0:N - 2].reshape(B * M, -1, h, w), cfg.corr_levels, cfg.corr_radius)
        cnet = self.cnet(images[:, 1:N - 1].reshape(B * M, 3, H, W))
        net, inp = torch.split(cnet, [self.hidden_dim, self.context_dim], dim=1)
        net, inp = torch.tanh(net), torch.relu(inp)
        coords0 = coords_grid(B * M, h, w, images.dtype)
        fcoords1, bcoords1 = coords0.clone(), coords0.clone()
        for _ in range(cfg.decoder_depth):
            fcorr, bcorr = fcorr_fn(fcoords1), bcorr_fn(bcoords1)
            net, up_mask, delta = self.update_block(net, inp, fcorr, bcorr,
                                                    fcoords1 - coords0, bcoords1 - coords0, B)
            fcoords1 = fcoords1 + delta[:, 0:2]
            bcoords1 = bcoords1 + delta[:, 2:4]
        fmask, bmask = torch.split(up_mask, [576, 576], dim=1)
        fup = upsample_flow(fcoords1 - coords0, fmask).reshape(B, M, 2, H, W)
        bup = upsample_flow(bcoords1 - coords0, bmask).reshape(B, M, 2, H, W)
        low = torch.cat([(fcoords1 - coords0).reshape(B, M, 2, h, w),
                         (bcoords1 - coords0).reshape(B, M, 2, h, w)], dim=1)
        return torch.cat([fup, bup], dim=1), low


def build_network(cfg):
    return MOFNetOracle(cfg)
