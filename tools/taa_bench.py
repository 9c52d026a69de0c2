"""vfml_taa_blend on a 1080p frame: time and achieved HBM bandwidth per mode (dev tool, GPU only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd"))
import numpy as np, torch
from vfml import hip
from effects import TAAProcessor
h, w = 1080, 1920
cur = torch.randint(0, 256, (h, w, 3), dtype=torch.uint8).cuda()
flow = (torch.randn(h, w, 2) * 3).cuda()
for mode, name, hist_dt, per_px in ((hip.TAA_BILATERAL, "bilateral f64 history", torch.float64, 3 + 8 + 24 + 24),
                                    (hip.TAA_BILATERAL, "bilateral f32 history", torch.float32, 3 + 8 + 12 + 24),
                                    (hip.TAA_BILINEAR, "bilinear f32 history", torch.float32, 3 + 8 + 12 + 12),
                                    (hip.TAA_SIMPLE, "simple f32 history", torch.float32, 3 + 12 + 12)):
    hist = (torch.rand(h, w, 3) * 255).to(hist_dt).cuda()
    for _ in range(3): hip.taa_blend(cur, flow, hist, mode, 0.1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): hip.taa_blend(cur, flow, hist, mode, 0.1)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"{name:24s} {us:7.1f} us per 1080p frame = {h * w * per_px / us / 1e3:7.1f} GB/s ({h * w * per_px / 1e6:.0f} MB algorithmic)")
p = TAAProcessor(0.1)
hc, hf = cur.cpu().numpy(), flow.cpu().numpy()
p.apply_taa(hc, hf); p.apply_taa(hc, hf)
t0 = time.time(); p.apply_taa(hc, hf); print(f"host numpy path (bilateral, f64 history): {(time.time() - t0) * 1e3:.0f} ms")
