"""TAA blend with flow reprojection against vectors cut from the reference's own
effects/taa_processor.py (tests/golden/make_taa_fixtures.py): whole sequences, so that the history feedback and
its dtype changes (float32 on the second frame, float64 after) are covered; NaN / inf / out-of-image flow vectors;
luminance jumps where the float32 weights underflow.  Host path: identical arithmetic, tolerance 1e-9 on 0..255.
HIP kernel: every step but exp() is reproduced exactly; tolerance 1e-3 on 0..255 (observed ~1e-5)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd"))

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "taa.npz"))
N = 5


def _sequences(wrap):
    """(key, result) for every fixture, `wrap` moving the inputs to where the path under test wants them."""
    from effects import TAAComparisonProcessor, TAAProcessor, apply_taa_effect
    frames, flows = GOLD["frames"], GOLD["flows"]
    for tag, kw in (("bilateral", dict(use_bilateral=True)), ("bilinear", dict(use_bilateral=False))):
        p = TAAProcessor(alpha=0.1)
        for i in range(N):
            yield f"{tag}_{i}", p.apply_taa(wrap(frames[i]), None if i == 0 else wrap(flows[i]), use_flow=True,
                                            sequence_id="s", **kw)
    p = TAAProcessor(alpha=0.25, bilateral_sigma_color=8.0)
    for i in range(N):
        yield f"sigma8_{i}", p.apply_taa(wrap(frames[i]), wrap(flows[i]), sequence_id="q")
    p = TAAProcessor(alpha=0.1)
    for i in range(N):
        yield f"simple_{i}", p.apply_simple_taa(wrap(frames[i]))
    yield "simple_on_f64", TAAProcessor(0.3).apply_taa(wrap(frames[4]), None, previous_taa_frame=wrap(GOLD["bilateral_3"]),
                                                      use_flow=False)
    yield "explicit_prev", TAAProcessor(0.1).apply_taa(wrap(frames[4].astype(np.float32)), wrap(flows[4]),
                                                     previous_taa_frame=wrap(GOLD["bilinear_3"]))
    yield "bilinear_on_f64", TAAProcessor(0.1).apply_taa(wrap(frames[4]), wrap(flows[4]),
                                                       previous_taa_frame=wrap(GOLD["bilateral_3"]), use_bilateral=False)
    yield "effect_fn", apply_taa_effect(wrap(frames[3]), wrap(flows[3]), previous_taa_frame=wrap(GOLD["simple_2"]), alpha=0.2)
    c = TAAComparisonProcessor(alpha=0.15)
    for i in range(3):
        a, b = c.apply_comparison(wrap(frames[i]), wrap(flows[i]))
        yield f"cmp_flow_{i}", a
        yield f"cmp_simple_{i}", b


def _check(key, got, tol):
    want = GOLD[key]
    assert got.shape == want.shape and got.dtype == want.dtype, (key, got.dtype, want.dtype)
    assert np.isfinite(got).all(), key
    err = np.abs(got.astype(np.float64) - want.astype(np.float64)).max()
    assert err <= tol, (key, err)
    return err


def test_host_taa_reproduces_the_reference_sequences():
    with np.errstate(all="ignore"):
        n = sum(1 for key, got in _sequences(lambda a: a.copy()) if _check(key, got, 1e-9) is not None)
    assert n == 4 * N + 4 + 6


def test_host_samplers_outside_the_image():
    from effects import TAAProcessor
    got = TAAProcessor()._bilinear_sample(GOLD["bilateral_2"], GOLD["xs"], GOLD["ys"])
    assert got.dtype == np.float32 and np.array_equal(got, GOLD["bilinear_sampler"])


def test_history_bookkeeping():
    from effects import TAAProcessor
    p = TAAProcessor(alpha=0.5)
    f = GOLD["frames"][0]
    first = p.apply_taa(f, sequence_id="a")
    assert first.dtype == np.float32 and np.array_equal(first, f) and p.get_history("a") is first
    assert p.get_history("b") is None
    p.apply_taa(f, sequence_id="b")
    p.reset_history("a")
    assert p.get_history("a") is None and p.get_history("b") is not None
    p.reset_history()
    assert p.history == {}
    p.set_alpha(1.0)
    with pytest.raises(ValueError, match="between 0.0 and 1.0"):
        p.set_alpha(1.5)


@pytest.mark.gpu
def test_hip_taa_matches_the_reference_sequences():
    dev = torch.device("cuda:0")
    worst = {}
    for key, got in _sequences(lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)):
        assert got.is_cuda
        worst[key] = _check(key, got.cpu().numpy(), 1e-3)
    assert len(worst) == 4 * N + 4 + 6
    # the paths without exp() are exact
    for key in [f"bilinear_{i}" for i in range(N)] + [f"simple_{i}" for i in range(N)] + ["simple_on_f64", "bilinear_on_f64"]:
        assert worst[key] == 0.0, (key, worst[key])
    print("max |HIP - reference| over the bilateral sequences:", max(worst.values()))


@pytest.mark.gpu
def test_hip_taa_full_size_against_the_host_path():
    """1080p, three steps (float32 history, then float64): device and host paths agree; flow 0 with alpha 1 returns the
    frame; a constant history survives any flow."""
    from effects import TAAProcessor
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    h, w = 1080, 1920
    frames = rng.integers(0, 256, size=(3, h, w, 3), dtype=np.uint8)
    flows = (rng.standard_normal((3, h, w, 2)) * 4).astype(np.float32)
    host, devp = TAAProcessor(0.1), TAAProcessor(0.1)
    for i in range(3):
        a = host.apply_taa(frames[i], flows[i])
        b = devp.apply_taa(torch.from_numpy(frames[i]).to(dev), torch.from_numpy(flows[i]).to(dev))
        assert b.dtype == (torch.float32 if i == 0 else torch.float64)
        err = np.abs(b.cpu().numpy() - a).max(axis=2)
        print(f"step {i}: max err {err.max():.3g}, pixels over 1e-3: {(err > 1e-3).sum()}, over 1e-6: {(err > 1e-6).sum()}")
        assert err.max() < 1e-3
    flat = torch.full((h, w, 3), 77.0, dtype=torch.float64, device=dev)
    cur = torch.from_numpy(frames[0]).to(dev)
    out = TAAProcessor(0.0).apply_taa(cur, torch.from_numpy(flows[0]).to(dev), previous_taa_frame=flat)
    assert torch.allclose(out, flat, rtol=0, atol=1e-9)
    out = TAAProcessor(1.0).apply_taa(cur, torch.zeros(h, w, 2, device=dev), previous_taa_frame=flat)
    assert torch.equal(out, cur.double())


@pytest.mark.gpu
def test_hip_taa_rejects_bad_arguments():
    from vfml import hip
    dev = torch.device("cuda:0")
    cur = torch.zeros(8, 8, 3, dtype=torch.uint8, device=dev)
    hist = torch.zeros(8, 8, 3, dtype=torch.float32, device=dev)
    with pytest.raises(ValueError, match="does not match"):
        hip.taa_blend(cur, torch.zeros(8, 7, 2, device=dev), hist, hip.TAA_BILATERAL, 0.1)
    with pytest.raises(ValueError, match="expected"):
        hip.taa_blend(cur, None, hist.half(), hip.TAA_SIMPLE, 0.1)
    with pytest.raises(RuntimeError, match="2x2"):
        hip.taa_blend(cur[:1].contiguous(), torch.zeros(1, 8, 2, device=dev), hist[:1].contiguous(), hip.TAA_BILATERAL, 0.1)
