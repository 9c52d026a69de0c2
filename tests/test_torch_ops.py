"""`torch.ops.vfml.*`: the hot ops registered with torch.library (vfml/torch_ops.py)."""
import math

import pytest
import torch
import torch.nn.functional as F


def test_ops_are_registered_and_have_no_cpu_implementation():
    import vfml.torch_ops  # noqa: F401  (registers)
    for name in ("conv2d_nhwc", "corr_volume", "corr_lookup", "convex_upsample"):
        assert hasattr(torch.ops.vfml, name), name
    schema = str(torch.ops.vfml.corr_lookup.default._schema)
    assert "Tensor[] pyramid" in schema and "int radius" in schema
    with pytest.raises(NotImplementedError):                      # only the CUDA (HIP) key has a kernel
        torch.ops.vfml.convex_upsample(torch.zeros(4, 4, 2), torch.zeros(4, 4, 576))
    with pytest.raises(NotImplementedError):
        torch.ops.vfml.corr_volume(torch.zeros(8, 32), torch.zeros(8, 32), 1.0)


@pytest.mark.gpu
def test_ops_match_pytorch_and_the_oracle(gpu):
    import vfml.torch_ops  # noqa: F401
    from oracle import mof_oracle as mo
    g = torch.Generator().manual_seed(61)
    # convolution
    x = torch.randn(2, 64, 21, 30, generator=g)
    wt = torch.randn(96, 64, 3, 3, generator=g) / math.sqrt(64 * 9)
    b = torch.randn(96, generator=g)
    got = torch.ops.vfml.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), wt.cuda(), b.cuda(), 2, 1, 1, "relu")
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), stride=2, padding=1)).float()
    assert tuple(got.shape) == (2, 11, 15, 96)
    assert ((got.cpu().permute(0, 3, 1, 2) - ref).abs().max() / ref.abs().max()).item() < 5e-6
    # all-pairs correlation
    f1, f2 = torch.randn(700, 256, generator=g), torch.randn(1300, 256, generator=g)
    vol = torch.ops.vfml.corr_volume(f1.cuda(), f2.cuda(), 1.0 / 16.0)
    ref = (f1.double() @ f2.double().t() / 16.0).float()
    assert ((vol.cpu() - ref).abs().max() / ref.abs().max()).item() < 5e-6
    # pyramid lookup vs the oracle's CorrBlock (grid_sample)
    h, w, levels, radius = 18, 24, 4, 4
    P = h * w
    pyr, hh, ww = [], h, w
    for _ in range(levels):
        pyr.append(torch.randn(P, 1, hh, ww, generator=g))
        hh, ww = hh // 2, ww // 2
    coords = mo.coords_grid(1, h, w) + torch.randn(1, 2, h, w, generator=g) * 5.0
    blk = mo.CorrBlock.__new__(mo.CorrBlock)
    blk.num_levels, blk.radius, blk.pyramid = levels, radius, pyr
    ref = blk(coords)[0].permute(1, 2, 0).reshape(P, -1)
    dev = []
    for p in pyr:
        s = p.shape[-2] * p.shape[-1]
        t = torch.zeros(P, (s + 31) // 32 * 32)
        t[:, :s] = p.reshape(P, s)
        dev.append(t.cuda())
    got = torch.ops.vfml.corr_lookup(dev, coords[0].permute(1, 2, 0).reshape(P, 2).cuda(), [p.shape[-2] for p in pyr],
                                     [p.shape[-1] for p in pyr], radius)
    assert (got.cpu() - ref).abs().max().item() < 2e-5
    # convex upsampling vs the oracle
    flow = torch.randn(1, 2, h, w, generator=g) * 3
    mask = torch.randn(1, 576, h, w, generator=g) * 2
    up = torch.ops.vfml.convex_upsample(flow[0].permute(1, 2, 0).contiguous().cuda(),
                                        mask[0].permute(1, 2, 0).contiguous().cuda())
    ref = mo.upsample_flow(flow, mask)[0].permute(1, 2, 0)
    assert (up.cpu() - ref).abs().max().item() < 2e-5
