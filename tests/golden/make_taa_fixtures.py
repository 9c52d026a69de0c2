"""Cut golden vectors from the REFERENCE's TAA processor (build container only).

    python tests/golden/make_taa_fixtures.py [/root/reference]

effects/taa_processor.py of the reference imports cv2 at module level without using it on this path; a stub module
stands in.  Inputs and the reference's outputs are stored as data (taa.npz); nothing of the reference's source
text is.  The GPU box never runs this file.
"""
import os
import sys
import types

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, REF)
    from effects.taa_processor import TAAComparisonProcessor, TAAProcessor, apply_taa_effect
    rng = np.random.default_rng(20250830)
    h, w, n = 31, 43, 5
    # a drifting texture + noise, so that reprojection has something to follow
    base = rng.integers(0, 256, size=(h + 16, w + 16, 3)).astype(np.float32)
    k = np.ones(5, np.float32) / 5
    for ax in (0, 1):
        base = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), ax, base)
    frames = np.stack([np.clip(base[8 + i:8 + i + h, 8 - i:8 - i + w] + rng.normal(0, 6, (h, w, 3)), 0, 255)
                       for i in range(n)]).astype(np.uint8)
    frames[2, :6, :6] = 255                                   # luminance jumps: float32 weights underflow to 0 here
    frames[1, :6, :6] = 0
    flows = (rng.standard_normal((n, h, w, 2)) * 1.5 + np.array([1.0, -1.0])).astype(np.float32)
    flows[1, 0, :6] = [[np.nan, 0], [0, np.nan], [np.inf, 1], [1, -np.inf], [1e9, -1e9], [-0.5, -0.5]]
    flows[2, -1, -4:] = [[5, 5], [0.25, 0.75], [-100, 3], [0, 0]]
    flows[3, 5] = 0.0
    out = {"frames": frames, "flows": flows}
    with np.errstate(all="ignore"):
        for tag, kw in (("bilateral", dict(use_bilateral=True)), ("bilinear", dict(use_bilateral=False))):
            p = TAAProcessor(alpha=0.1)
            for i in range(n):
                r = p.apply_taa(frames[i], None if i == 0 else flows[i], use_flow=True, sequence_id="s", **kw)
                out[f"{tag}_{i}"] = r
        p = TAAProcessor(alpha=0.25, bilateral_sigma_color=8.0)
        for i in range(n):
            out[f"sigma8_{i}"] = p.apply_taa(frames[i], flows[i], sequence_id="q")
        p = TAAProcessor(alpha=0.1)
        for i in range(n):
            out[f"simple_{i}"] = p.apply_simple_taa(frames[i])
        # simple blend on a float64 history, explicit history argument, float32 current frame
        out["simple_on_f64"] = TAAProcessor(0.3).apply_taa(frames[4], None, previous_taa_frame=out["bilateral_3"],
                                                            use_flow=False)
        out["explicit_prev"] = TAAProcessor(0.1).apply_taa(frames[4].astype(np.float32), flows[4],
                                                           previous_taa_frame=out["bilinear_3"])
        out["bilinear_on_f64"] = TAAProcessor(0.1).apply_taa(frames[4], flows[4], previous_taa_frame=out["bilateral_3"],
                                                             use_bilateral=False)
        out["effect_fn"] = apply_taa_effect(frames[3], flows[3], previous_taa_frame=out["simple_2"], alpha=0.2)
        c = TAAComparisonProcessor(alpha=0.15)
        for i in range(3):
            a, b = c.apply_comparison(frames[i], flows[i])
            out[f"cmp_flow_{i}"], out[f"cmp_simple_{i}"] = a, b
        # the samplers on their own, with coordinates outside the image
        xs = rng.uniform(-3, w + 3, (h, w))
        ys = rng.uniform(-3, h + 3, (h, w))
        out["xs"], out["ys"] = xs, ys
        out["bilinear_sampler"] = TAAProcessor()._bilinear_sample(out["bilateral_2"], xs, ys)
    np.savez_compressed(os.path.join(HERE, "taa.npz"), **out)
    print("wrote taa.npz:", len(out), "arrays;", {k: str(v.dtype) for k, v in out.items() if k.endswith("_2")})


if __name__ == "__main__":
    main()
