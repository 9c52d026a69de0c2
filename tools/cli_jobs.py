"""The three whole-job configurations of BASELINE.json through flow_processor.py on one GPU (dev tool, GPU only):
synthetic clips, seeded weights, compressed .npz cache, --skip-lods.  Prints the CLI's own timing lines."""
import os, sys, tempfile, contextlib, io, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd"))
import torch
import flow_processor
from vfml import get_cfg
from vfml.memflow_net import memflow_cfg, seeded_memflow_state_dict
from vfml.weights import write_seeded_checkpoint

work = tempfile.mkdtemp(prefix="vfml_jobs_", dir=os.environ.get("TMPDIR", "/tmp"))
write_seeded_checkpoint(work, get_cfg(), seed=0)
os.makedirs(os.path.join(work, "MemFlow_ckpt"))
torch.save(seeded_memflow_state_dict(memflow_cfg(), 0), os.path.join(work, "MemFlow_ckpt", "MemFlowNet_sintel.pth"))
os.chdir(work)
jobs = {"C2": ["--input", "synthetic:1920x1080x300", "--sequence-length", "5"],
        "C4": ["--input", "synthetic:1920x1080x100", "--model", "memflow"],
        "C3": ["--input", "synthetic:3840x2160x24", "--sequence-length", "5", "--tile"]}
for name in (sys.argv[1:] or list(jobs)):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = flow_processor.main(jobs[name] + ["--output", os.path.join(work, "out_" + name), "--device", "cuda", "--interactive",
                                               "--skip-lods"])
    lines = [l for l in buf.getvalue().splitlines() if re.search(r"fields/s|frames/s|field", l)]
    print(name, "rc", rc)
    for l in lines[-6:]:
        print("   ", l.strip()[:200])
    sys.stdout.flush()
