"""PCIe-inclusive rates of the 1080p seq5 job (dev tool, GPU only): never bench.py's `value`.

(a) the reference's own host API, one call per field: T float32 frames up (124 MB), one field down
    (16.6 MB), synchronous  (processing/videoflow_processor.py:122-185 of the reference);
(b) the drop-in's job loop: the uint8 clip uploaded once, fields computed from HBM, every field
    copied to host memory as it finishes.
"""
import contextlib, io, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from processing.flow_inference import VideoFlowInference
from vfml import get_cfg
from vfml.synth import synthetic_clip
from vfml.weights import write_seeded_checkpoint

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
tmp = tempfile.mkdtemp()
write_seeded_checkpoint(tmp, get_cfg(), seed=0)
os.chdir(tmp)
with contextlib.redirect_stdout(io.StringIO()):
    eng = VideoFlowInference("cuda", sequence_length=5)
    eng.load_model()
frames = synthetic_clip(n, 1080, 1920)
proc = eng.get_processor()

eng.compute_optical_flow(frames, 2); torch.cuda.synchronize()
t0 = time.time()
ta = []
for i in range(n):
    eng.compute_optical_flow(frames, i)
    ta.append(time.time())
dt = time.time() - t0
steady = (ta[-1] - ta[3]) / (n - 4)
print(f"(a) reference host API (numpy frames in, numpy field out, one synchronous call per field): {n / dt:6.2f} fields/s "
      f"over the {n}-field job, {1 / steady:6.2f} fields/s ({steady * 1e3:.1f} ms/field) after the first 4 fields")

t0 = time.time()
clip = proc.upload_clip(frames)
host = torch.empty((n, 1080, 1920, 2), dtype=torch.float32).pin_memory()
marks = [time.time()]
for i in range(n):
    host[i].copy_(proc.compute_optical_flow_resident(clip, i), non_blocking=True)
    if os.environ.get("PER_FIELD"):
        torch.cuda.synchronize(); marks.append(time.time())
torch.cuda.synchronize()
dt = time.time() - t0
if os.environ.get("PER_FIELD"):
    print("    setup %.0f ms; per field ms: %s" % ((marks[0] - t0) * 1e3, " ".join(f"{(b - a) * 1e3:.0f}" for a, b in zip(marks, marks[1:]))))
print(f"(b) u8 clip uploaded once ({clip.numel() / 1e6:.0f} MB), fields copied to pinned host memory as they finish: "
      f"{n / dt:6.2f} fields/s ({dt / n * 1e3:.1f} ms/field)")
