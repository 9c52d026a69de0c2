"""Flow -> 8-bit image encoders against vectors cut from the reference's own
encoding/flow_encoders.py (tests/golden/make_encoder_fixtures.py): bytes, so bit-exact - the numpy host path on
CPU, the HIP kernel on the GPU, including NaN / inf / on-the-clamp inputs."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd"))

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "flow_encoders.npz"))


def _cases():
    from encoding import GamedevFlowEncoder, MotionVectorsRG8FlowEncoder, MotionVectorsRGB8FlowEncoder
    for name in ("flow", "small"):
        f = GOLD[name]
        h, w = f.shape[:2]
        yield f"gamedev_{name}", GamedevFlowEncoder(), f, w, h
        yield f"gamedev_1080_{name}", GamedevFlowEncoder(), f, 1920, 1080
        yield f"gamedev_s50c5_{name}", GamedevFlowEncoder(scale_factor=50.0, clamp_range=5.0), f, w, h
        for c in (64.0, 32.0, 2.5):
            yield f"rg8_{c}_{name}", MotionVectorsRG8FlowEncoder(clamp_range=c), f, w, h
            yield f"rgb8_{c}_{name}", MotionVectorsRGB8FlowEncoder(clamp_range=c), f, w, h


def test_host_encoders_reproduce_the_reference_bytes():
    n = 0
    for key, enc, f, w, h in _cases():
        got = enc.encode(f.copy(), w, h)
        assert got.dtype == np.uint8 and got.shape == f.shape[:2] + (3,)
        assert np.array_equal(got, GOLD[key]), key
        n += 1
    assert n == 18


def test_host_decoders_and_factory():
    from encoding import (FlowEncoderFactory, MotionVectorsRG8FlowEncoder, MotionVectorsRGB8FlowEncoder, decode_motion_vectors,
                          encode_flow, encode_motion_vectors)
    enc8 = GOLD["enc8"]
    for c in (64.0, 32.0):
        assert np.array_equal(MotionVectorsRG8FlowEncoder(clamp_range=c).decode(enc8), GOLD[f"rg8_dec_{c}"], equal_nan=True)
        assert np.array_equal(MotionVectorsRGB8FlowEncoder(clamp_range=c).decode(enc8), GOLD[f"rgb8_dec_{c}"], equal_nan=True)
        assert np.array_equal(decode_motion_vectors(enc8, c, "rg8"), GOLD[f"rg8_dec_{c}"], equal_nan=True)
    d = GOLD["default_clamps"]
    assert (MotionVectorsRG8FlowEncoder().clamp_range, MotionVectorsRGB8FlowEncoder().clamp_range) == (d[0], d[1])
    f = GOLD["flow"]
    assert np.array_equal(encode_flow(f.copy(), f.shape[1], f.shape[0]), GOLD["gamedev_flow"])
    assert np.array_equal(encode_motion_vectors(f.copy(), 64.0, "rg8"), GOLD["rg8_64.0_flow"])
    assert np.array_equal(encode_motion_vectors(f.copy(), 32.0), GOLD["rgb8_32.0_flow"])
    ours = set(FlowEncoderFactory.get_available_formats())
    assert ours == set(GOLD["factory_formats"].tolist()) - {"hsv", "torchvision"}
    with pytest.raises(ValueError, match="Unsupported format"):
        FlowEncoderFactory.create_encoder("nope")
    with pytest.raises(ValueError, match="not part of this build"):
        FlowEncoderFactory.create_encoder("hsv")


@pytest.mark.gpu
def test_gpu_encoders_reproduce_the_reference_bytes(gpu):
    n = 0
    for key, enc, f, w, h in _cases():
        got = enc.encode(torch.from_numpy(f).cuda(), w, h)
        assert got.is_cuda and got.dtype == torch.uint8
        assert np.array_equal(got.cpu().numpy(), GOLD[key]), key
        n += 1
    assert n == 18


@pytest.mark.gpu
def test_gpu_encoder_on_a_full_field_equals_host(gpu):
    """1080p field: device kernel == host numpy path (which is pinned to the reference above)."""
    from encoding import MotionVectorsRGB8FlowEncoder, GamedevFlowEncoder
    g = torch.Generator().manual_seed(5)
    f = (torch.randn(1080, 1920, 2, generator=g) * 20).numpy()
    for enc in (MotionVectorsRGB8FlowEncoder(), GamedevFlowEncoder()):
        assert np.array_equal(enc.encode(torch.from_numpy(f).cuda(), 1920, 1080).cpu().numpy(), enc.encode(f.copy(), 1920, 1080))
