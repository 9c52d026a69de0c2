"""Analytic known-answer tests that pin the CPU oracle (oracle/mof_oracle.py).  The reference has
no model arithmetic and no golden vectors for it ("parity unpinned"), so these are
the oracle's only anchors: closed-form answers of the published algorithm."""
import math

import torch
import torch.nn.functional as F

from oracle import mof_oracle as mo


def test_correlation_of_one_hot_features_is_a_delta_over_sqrt_d():
    h, w, d = 4, 4, 16                       # one distinct one-hot channel per cell
    f = torch.zeros(1, d, h, w)
    for k in range(h * w):
        f[0, k, k // w, k % w] = 1.0
    blk = mo.CorrBlock(f, f, num_levels=2, radius=1)
    c0 = blk.pyramid[0].view(h * w, h * w)
    assert torch.equal(c0, torch.eye(h * w) / math.sqrt(d))
    assert torch.allclose(blk.pyramid[1].sum((1, 2, 3)), torch.full((h * w,), 0.25 / math.sqrt(d)))


def test_pyramid_level_equals_correlation_with_pooled_features():
    """avg-pool commutes with the dot product: the identity the engine's K3/K4 build rests on."""
    g = torch.Generator().manual_seed(0)
    f1, f2 = torch.randn(1, 32, 9, 11, generator=g), torch.randn(1, 32, 9, 11, generator=g)
    blk = mo.CorrBlock(f1, f2, num_levels=3, radius=2)
    p = f2
    for l in range(1, 3):
        p = F.avg_pool2d(p, 2, 2)
        alt = torch.matmul(f1.view(1, 32, -1).transpose(1, 2), p.view(1, 32, -1)) / math.sqrt(32)
        assert torch.allclose(blk.pyramid[l].view(99, -1), alt[0], atol=1e-5)


def test_lookup_at_integer_coords_is_a_gather_in_rafts_window_order():
    h, w, r = 8, 8, 2
    blk = mo.CorrBlock.__new__(mo.CorrBlock)
    blk.num_levels, blk.radius = 1, r
    blk.pyramid = [torch.arange(h * w * h * w, dtype=torch.float32).view(h * w, 1, h, w)]
    out = blk(mo.coords_grid(1, h, w))       # zero flow
    y, x = 3, 4
    q = y * w + x
    for i in range(2 * r + 1):
        for j in range(2 * r + 1):
            xs, ys = x + i - r, y + j - r    # channel i*(2r+1)+j samples (x+d[i], y+d[j])
            assert out[0, i * (2 * r + 1) + j, y, x].item() == blk.pyramid[0][q, 0, ys, xs].item()
    assert out[0, 0, 0, 0].item() == 0.0     # window corner outside the map -> zeros padding


def test_upsample_uniform_mask_is_box_mean_times_eight():
    flow = torch.zeros(1, 2, 5, 6)
    flow[:, 0] = 1.0
    up = mo.upsample_flow(flow, torch.zeros(1, 576, 5, 6))
    assert up.shape == (1, 2, 40, 48)
    assert torch.allclose(up[0, 0, 16:24, 16:24], torch.full((8, 8), 8.0))
    assert torch.allclose(up[0, 0, :8, :8], torch.full((8, 8), 8.0 * 4 / 9))
    assert (up[0, 1] == 0).all()


def test_upsample_one_hot_mask_picks_one_neighbour():
    g = torch.Generator().manual_seed(1)
    flow = torch.randn(1, 2, 4, 4, generator=g)
    mask = torch.full((1, 9, 64, 4, 4), -1e4)
    mask[:, 5] = 0.0                                   # tap 5 = (dy 0, dx +1)
    up = mo.upsample_flow(flow, mask.view(1, 576, 4, 4))
    assert torch.allclose(up[0, :, 8:16, 8:16], (8 * flow[0, :, 1, 2]).view(2, 1, 1).expand(2, 8, 8), atol=1e-5)


def test_padder_roundtrip_and_noop():
    p = mo.InputPadder((1080, 1920))
    x = torch.zeros(1, 5, 3, 1080, 1920)
    assert p.pad(x) is x
    p = mo.InputPadder((100, 203))
    x = torch.rand(1, 3, 3, 100, 203)
    y = p.pad(x)
    assert y.shape == (1, 3, 3, 104, 208)
    assert torch.equal(p.unpad(y), x)
    assert torch.equal(y[..., 0, 2:-3], x[..., 0, :])    # replicate border, 2 top / 2 bottom, 2 left / 3 right


def test_engine_padder_matches_oracle_padder():
    from vfml import InputPadder
    for dims in ((100, 203), (1080, 1920), (257, 255), (8, 8)):
        a, b = mo.InputPadder(dims), InputPadder(dims)
        x = torch.rand(1, 3, 3, *dims)
        assert torch.equal(a.pad(x), b.pad(x))
        assert torch.equal(a.unpad(a.pad(x)), b.unpad(b.pad(x)))


def test_output_layout_forward_then_backward_and_zero_motion_symmetry():
    """Identical frames: the forward and backward problems of a centre frame coincide, so their
    flows must be equal -> the [fwd..., bwd...] stacking is what the reference indexes (:194)."""
    cfg = mo.get_cfg()
    cfg.decoder_depth = 2
    net = mo.build_network(cfg).eval()
    torch.manual_seed(0)
    x = torch.rand(1, 1, 3, 128, 128).repeat(1, 3, 1, 1, 1)
    flow, low = net(x, {})
    assert flow.shape == (1, 2, 2, 128, 128) and low.shape == (1, 2, 2, 16, 16)
    # fwd and bwd see the same correlation volume and start from the same state; the update block
    # treats them through different weights, so only shapes/finite-ness are asserted beyond that.
    assert torch.isfinite(flow).all()


def test_seeded_state_dict_is_deterministic_and_loads_strictly():
    from vfml import get_cfg
    from vfml.weights import conv_spec, seeded_state_dict
    cfg = get_cfg()
    a, b = seeded_state_dict(cfg, 0), seeded_state_dict(cfg, 0)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert not torch.equal(a["fnet.conv1.weight"], seeded_state_dict(cfg, 1)["fnet.conv1.weight"])
    net = mo.build_network(mo.get_cfg())
    net.load_state_dict(a, strict=True)
    assert len(conv_spec(cfg)) * 2 == len(a)


def test_fast_mode_lookup_is_a_channel_subset_of_the_base_lookup():
    """--fast (3 levels, radius 3) on a checkpoint trained for the base lookup: the smaller lookup's
    channels are exactly `corr_channel_subset` of the base one's, which is what lets the engine and the
    oracle use the matching input columns of convc1 (vfml/weights.py)."""
    import torch
    from oracle import mof_oracle as mo
    from vfml.weights import corr_channel_subset
    g = torch.Generator().manual_seed(3)
    f1, f2 = torch.randn(1, 32, 16, 24, generator=g), torch.randn(1, 32, 16, 24, generator=g)
    coords = mo.coords_grid(1, 16, 24, torch.float32) + torch.randn(1, 2, 16, 24, generator=g) * 3
    base = mo.CorrBlock(f1, f2, 4, 4)(coords)
    small = mo.CorrBlock(f1, f2, 3, 3)(coords)
    sel = torch.tensor(corr_channel_subset(3, 3))
    assert small.shape[1] == sel.numel() == 147
    assert torch.equal(small, base[:, sel])
