"""Where the first fields of a job spend their time (dev tool, GPU only): a FRESH process loads the model, then computes the
first fields of a 1080p seq-5 job one at a time with a device synchronise after each, printing wall ms per field, the
bytes torch's allocator took from the driver meanwhile and the number of hipMalloc-backed segments.

    python tools/cold_start.py [fields=10] [HxW=1080x1920]

VFML_RESERVE_GB / VFML_WARM (processing/videoflow_core.py) change what load_model() does ahead of the job."""
import contextlib
import io
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

os.environ.setdefault("VFML_PRECISION", "mixed")
from vfml import get_cfg  # noqa: E402
from vfml.runner import ClipFeeder, run_sharded  # noqa: E402
from vfml.synth import synthetic_clip  # noqa: E402
from vfml.weights import write_seeded_checkpoint  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
H, W = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1080x1920").split("x"))
work = tempfile.mkdtemp(prefix="vfml_cold_")
write_seeded_checkpoint(work, get_cfg(), seed=0)
os.chdir(work)
from processing.videoflow_processor import VideoFlowProcessor  # noqa: E402

clip_np = synthetic_clip(n + 4, H, W)
t0 = time.perf_counter()
torch.cuda.init()
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
t_ctx = time.perf_counter() - t0
t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    proc = VideoFlowProcessor("cuda", sequence_length=5)
    proc.load_model()
torch.cuda.synchronize()
t_load = time.perf_counter() - t0
print(f"context {1e3 * t_ctx:.0f} ms, load_model {1e3 * t_load:.0f} ms, reserved after load {torch.cuda.memory_reserved() / 2**30:.1f} GiB")


def segs():
    return torch.cuda.memory_stats().get("num_device_alloc", 0)


t0 = time.perf_counter()
feeder = ClipFeeder(clip_np, "cuda")
torch.cuda.synchronize()
print(f"ClipFeeder (clip + pinned ring) {1e3 * (time.perf_counter() - t0):.0f} ms")
tot0 = time.perf_counter()
for i in range(n):
    r0, s0 = torch.cuda.memory_reserved(), segs()
    t0 = time.perf_counter()
    out = run_sharded(proc, None, [i], feeder=feeder)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"field {i}: {1e3 * dt:7.1f} ms   +{(torch.cuda.memory_reserved() - r0) / 2**30:6.2f} GiB reserved, +{segs() - s0} hipMalloc")
print(f"{n} fields one by one: {time.perf_counter() - tot0:.3f} s")
# the same fields again as one job (caches warm, graph captured): the steady-state loop
torch.cuda.synchronize()
t0 = time.perf_counter()
run_sharded(proc, None, list(range(n)), feeder=feeder)
torch.cuda.synchronize()
print(f"the same {n} fields as one warm job: {1e3 * (time.perf_counter() - t0) / n:.1f} ms per field")
