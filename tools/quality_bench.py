"""vfml_flow_quality_map on 1080p frames: time and achieved HBM bandwidth (dev tool, GPU only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd"))
import torch
from vfml import hip
h, w = 1080, 1920
f1 = torch.randint(0, 256, (h, w, 3), dtype=torch.uint8).cuda()
f2 = torch.randint(0, 256, (h, w, 3), dtype=torch.uint8).cuda()
for name, fl in (("full-resolution field", (torch.randn(h, w, 2) * 3).cuda()), ("LOD-1 field (540x960)", torch.randn(540, 960, 2).cuda())):
    for _ in range(3): hip.flow_quality_map(f1, f2, fl, 0.9)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): hip.flow_quality_map(f1, f2, fl, 0.9)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    byts = h * w * 9 + fl.numel() * 4
    print(f"{name:24s} {us:7.1f} us per 1080p map = {byts / us / 1e3:7.1f} GB/s ({byts / 1e6:.1f} MB algorithmic)")
