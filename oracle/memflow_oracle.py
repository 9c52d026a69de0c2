"""CPU oracle for the MemFlow pair path (TEST INFRASTRUCTURE ONLY).  Status: PARITY UNPINNED.

The reference runs MemFlow only through a generated script that imports the un-vendored, un-pinned
submodule DQiaole/MemFlow (.gitmodules:4-6, directory empty; weights absent):
processing/memflow_inference_isolated.py:54-112 — `build_network(cfg)`, value-range normalisation
(:81-85), `InputPadder(frames.shape).pad`, a fresh `InferenceCore`, the LAST TWO frames (:97),
`processor.step(pair, end=True, flow_init=None)` (:102-107), unpad.  Nothing of the network is
observable, so — exactly as for the VideoFlow path — the architecture is defined in DESIGN.md §2b and
restated here in plain fp32 PyTorch: RAFT-style encoders and correlation pyramid, an update block whose
motion features are augmented by a memory read-out.  With the memory bank always empty (fresh
InferenceCore + end=True, reference :92,104) the read-out degenerates to attention of the frame's
query over its own key/value: softmax(q k^T / sqrt(d)) v over all P cells.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from .mof_oracle import BasicEncoder, CorrBlock, FlowHead, InputPadder, SepConvGRU, coords_grid, upsample_flow  # noqa: F401


def get_cfg():
    return SimpleNamespace(restore_ckpt="", network="MemFlowNet", feat_dim=256, down_ratio=8, corr_levels=4,
                           corr_radius=4, decoder_depth=12, att_dim=128)


class PairMotionEncoder(nn.Module):
    """RAFT BasicMotionEncoder (one correlation lookup, one flow).  The motion feature is 128 wide:
    124 conv outputs, the flow pair, and two always-zero channels (the slot a multi-frame network uses
    for the backward flow) - which keeps the flow on a 16-byte boundary for the engine."""

    def __init__(self, cor_planes):
        super().__init__()
        self.convc1 = nn.Conv2d(cor_planes, 256, 1)
        self.convc2 = nn.Conv2d(256, 192, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(192 + 64, 128 - 4, 3, padding=1)

    def forward(self, flow, corr):
        cor = F.relu(self.convc2(F.relu(self.convc1(corr))))
        flo = F.relu(self.convf2(F.relu(self.convf1(flow))))
        return torch.cat([F.relu(self.conv(torch.cat([cor, flo], dim=1))), flow, torch.zeros_like(flow)], dim=1)


class MemFlowUpdateBlock(nn.Module):
    def __init__(self, cor_planes, hidden_dim=128, att_dim=128):
        super().__init__()
        self.encoder = PairMotionEncoder(cor_planes)
        self.value = nn.Conv2d(128, att_dim, 1)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.gru = SepConvGRU(hidden_dim, input_dim=128 + 128 + 128)
        self.flow_head = FlowHead(hidden_dim, 256, 2)
        self.mask = nn.Sequential(nn.Conv2d(hidden_dim, 256, 3, padding=1), nn.ReLU(inplace=False),
                                  nn.Conv2d(256, 64 * 9, 1))


class MemFlowNetOracle(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.hidden_dim = self.context_dim = cfg.feat_dim // 2
        self.fnet = BasicEncoder(cfg.feat_dim)
        self.cnet = BasicEncoder(cfg.feat_dim)
        self.query = nn.Conv2d(self.context_dim, cfg.att_dim, 1)
        self.key = nn.Conv2d(self.context_dim, cfg.att_dim, 1)
        cor_planes = cfg.corr_levels * (2 * cfg.corr_radius + 1) ** 2
        self.update_block = MemFlowUpdateBlock(cor_planes, self.hidden_dim, cfg.att_dim)

    @torch.no_grad()
    def forward(self, pair):
        """pair: [1, 2, 3, H, W] in [-1, 1] (previous, current).  Returns (flow_low [1,2,h,w], flow [1,2,H,W])."""
        cfg, ub = self.cfg, self.update_block
        B, _, _, H, W = pair.shape
        h, w = H // 8, W // 8
        fm = self.fnet(pair.reshape(2 * B, 3, H, W)).reshape(B, 2, -1, h, w)
        corr_fn = CorrBlock(fm[:, 0], fm[:, 1], cfg.corr_levels, cfg.corr_radius)
        cn = self.cnet(pair[:, 0])
        net, inp = torch.split(cn, [self.hidden_dim, self.context_dim], dim=1)
        net, inp = torch.tanh(net), torch.relu(inp)
        q = self.query(inp).flatten(2).transpose(1, 2)                       # [B, P, d]
        k = self.key(inp).flatten(2)                                         # [B, d, P]
        attn = torch.softmax(torch.matmul(q, k) / math.sqrt(cfg.att_dim), dim=-1)   # memory = this frame only
        coords0 = coords_grid(B, h, w, pair.dtype)
        coords1 = coords0.clone()
        for _ in range(cfg.decoder_depth):
            corr = corr_fn(coords1)
            motion = ub.encoder(coords1 - coords0, corr)
            v = ub.value(motion).flatten(2).transpose(1, 2)                  # [B, P, d]
            readout = torch.matmul(attn, v).transpose(1, 2).reshape(B, -1, h, w)
            motion_global = motion + ub.gamma * readout
            net = ub.gru(net, torch.cat([inp, motion, motion_global], dim=1))
            coords1 = coords1 + ub.flow_head(net)
        up = upsample_flow(coords1 - coords0, 0.25 * ub.mask(net))
        return coords1 - coords0, up


def build_network(cfg):
    return MemFlowNetOracle(cfg)


def normalise_frames(frames):
    """Value-range heuristic of reference processing/memflow_inference_isolated.py:81-85."""
    mx = frames.max().item()
    if mx > 2.0:
        return 2 * (frames / 255.0) - 1.0
    if mx > 1.0:
        return 2 * frames - 1.0
    return frames


@torch.no_grad()
def compute_flow(net, frames):
    """frames [1,T,3,H,W] (0..255 floats as the reference's processor hands over) -> [2,H,W]:
    normalise, pad, last two frames, step, unpad, drop the batch dim (reference :80-112)."""
    x = normalise_frames(frames)
    padder = InputPadder(x.shape)
    x = padder.pad(x)
    _, up = net(x[:, -2:])
    return padder.unpad(up[0])
