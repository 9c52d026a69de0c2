"""Cut golden vectors from the REFERENCE's motion-vector / gamedev flow encoders (build container only).

    python tests/golden/make_encoder_fixtures.py [/root/reference]

encoding/flow_encoders.py of the reference imports cv2 at module level (only its HSV encoder uses it);
a stub module stands in.  Inputs and the reference's outputs are stored as data (flow_encoders.npz);
nothing of the reference's source text is.  The GPU box never runs this file.
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
    from encoding.flow_encoders import (FlowEncoderFactory, GamedevFlowEncoder, MotionVectorsRG8FlowEncoder,
                                        MotionVectorsRGB8FlowEncoder)
    rng = np.random.default_rng(20250829)
    h, w = 37, 53
    flow = (rng.standard_normal((h, w, 2)) * 30.0).astype(np.float32)
    flow[0, :8] = [[0, 0], [1e-7, -1e-7], [64, -64], [63.999, 64.001], [32, 32], [-32, 0], [0, 31.9999], [1e6, -1e6]]
    flow[1, :6] = [[np.nan, 1], [1, np.nan], [np.inf, 0], [0, -np.inf], [np.inf, np.inf], [-np.inf, np.nan]]
    flow[2, :4] = [[22.627417, 22.627417], [-22.627417, 22.627417], [32, 1e-3], [19.2, 25.6]]   # |v| ~ clamp 32
    small = (rng.standard_normal((h, w, 2)) * 0.7).astype(np.float32)
    enc8 = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    enc8[0, :4] = [[0, 0, 0], [255, 255, 255], [128, 128, 0], [127, 128, 255]]
    out = {"flow": flow, "small": small, "enc8": enc8}
    with np.errstate(all="ignore"):
        for name, f in (("flow", flow), ("small", small)):
            out[f"gamedev_{name}"] = GamedevFlowEncoder().encode(f.copy(), w, h)
            out[f"gamedev_1080_{name}"] = GamedevFlowEncoder().encode(f.copy(), 1920, 1080)
            out[f"gamedev_s50c5_{name}"] = GamedevFlowEncoder(scale_factor=50.0, clamp_range=5.0).encode(f.copy(), w, h)
            for c in (64.0, 32.0, 2.5):
                out[f"rg8_{c}_{name}"] = MotionVectorsRG8FlowEncoder(clamp_range=c).encode(f.copy(), w, h)
                out[f"rgb8_{c}_{name}"] = MotionVectorsRGB8FlowEncoder(clamp_range=c).encode(f.copy(), w, h)
        for c in (64.0, 32.0):
            out[f"rg8_dec_{c}"] = MotionVectorsRG8FlowEncoder(clamp_range=c).decode(enc8)
            out[f"rgb8_dec_{c}"] = MotionVectorsRGB8FlowEncoder(clamp_range=c).decode(enc8)
    out["factory_formats"] = np.array(FlowEncoderFactory.get_available_formats())
    out["default_clamps"] = np.array([MotionVectorsRG8FlowEncoder().clamp_range, MotionVectorsRGB8FlowEncoder().clamp_range,
                                      GamedevFlowEncoder().clamp_range, GamedevFlowEncoder().scale_factor], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "flow_encoders.npz"), **out)
    print("wrote flow_encoders.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
