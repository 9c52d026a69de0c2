"""Cut golden vectors from the REFERENCE's flow quality map (build container only).

    python tests/golden/make_quality_fixtures.py [/root/reference]

correction_worker.py of the reference imports cv2 and the VideoFlow submodule's flow writer at module level;
`generate_quality_frame_gpu` uses neither, so empty stand-in modules satisfy the imports.  It is run on
torch's CPU device (the function takes the device as an argument).  Inputs and the reference's outputs are stored
as data (quality_map.npz); nothing of the reference's source text is.  The GPU box never runs this file.
"""
import os
import sys
import types

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    for name in ("VideoFlow", "VideoFlow.core", "VideoFlow.core.utils", "VideoFlow.core.utils.frame_utils"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["VideoFlow.core.utils.frame_utils"].writeFlow = None
    sys.path.insert(0, REF)
    from correction_worker import generate_quality_frame_gpu
    rng = np.random.default_rng(20250831)
    h, w = 45, 61
    base = rng.integers(0, 256, size=(h + 8, w + 8, 3)).astype(np.float32)
    k = np.ones(3, np.float32) / 3
    for ax in (0, 1):
        base = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), ax, base)
    frame1 = np.clip(base[4:4 + h, 4:4 + w] + rng.normal(0, 3, (h, w, 3)), 0, 255).astype(np.uint8)
    frame2 = np.clip(base[3:3 + h, 6:6 + w] + rng.normal(0, 3, (h, w, 3)), 0, 255).astype(np.uint8)   # moved by (-2, +1)
    frame1[0, :4] = [[0, 0, 0], [255, 255, 255], [0, 0, 0], [1, 0, 0]]
    frame2[0, :4] = [[0, 0, 0], [255, 255, 255], [255, 255, 255], [0, 0, 1]]
    flow = (np.array([-2.0, 1.0]) + rng.standard_normal((h, w, 2)) * 0.4).astype(np.float32)
    flow[0, :4] = 0.0
    flow[1, :6] = [[np.inf, 0], [0, -np.inf], [1e9, 0], [-1e9, 2], [100, 0], [0, -100]]
    flow[2, :3] = [[-0.999, 0], [2.0, 2.0], [1.5, 2.5]]
    noise = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    out = {"frame1": frame1, "frame2": frame2, "flow": flow, "noise": noise}
    dev = torch.device("cpu")
    for thr in (0.9, 0.75):
        out[f"map_{thr}"] = generate_quality_frame_gpu(frame1, frame2, flow.copy(), dev, thr)
        out[f"map_noise_{thr}"] = generate_quality_frame_gpu(frame1, noise, flow.copy(), dev, thr)
    out["map_same"] = generate_quality_frame_gpu(frame1, frame1, np.zeros_like(flow), dev, 0.9)
    # fields at LOD resolution (the visualizer falls back to them): resized inside the function
    for tag, (fh, fw) in (("half", ((h + 1) // 2, (w + 1) // 2)), ("quarter", (12, 16)), ("tiny", (1, 1))):
        lod = (np.array([-1.0, 0.5]) * (fw / w) + rng.standard_normal((fh, fw, 2)) * 0.2).astype(np.float32)
        out[f"lod_{tag}"] = lod
        out[f"map_lod_{tag}"] = generate_quality_frame_gpu(frame1, frame2, lod.copy(), dev, 0.9)
    np.savez_compressed(os.path.join(HERE, "quality_map.npz"), **out)
    print("wrote quality_map.npz:", len(out), "arrays; torch", torch.__version__)


if __name__ == "__main__":
    main()
