"""Determinism soak of the two-stream engine: the fields of a streamed 1080p clip computed with the encoder prefetch on a side
stream (and, second half, with the flow half of the motion encoder on a third) against the one-stream engine, bit for bit,
PASSES times over in one process.

    python tools/exp/prefetch_soak.py [passes, default 30]"""
import contextlib, io, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd")); sys.path.insert(0, ROOT)
import torch
from vfml import build_network, get_cfg
from vfml.cfg import DEFAULT_MIXED_CORR_VOLUME, DEFAULT_MIXED_PLAN
from vfml.synth import synthetic_clip
from vfml.weights import seeded_state_dict
from processing.videoflow_processor import VideoFlowProcessor
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cfg = get_cfg(); cfg.precision = "mixed"; cfg.mfma_plan = dict(DEFAULT_MIXED_PLAN); cfg.corr_volume = DEFAULT_MIXED_CORR_VOLUME      # (what VideoFlowCore runs by default)
net = build_network(cfg); net.load_state_dict(seeded_state_dict(cfg, 0)); net = net.cuda().eval()
with contextlib.redirect_stdout(io.StringIO()):
    proc = VideoFlowProcessor("cuda", sequence_length=5)
proc.core.model = net
clip = proc.upload_clip(synthetic_clip(11, 1080, 1920))
order = list(range(2, 9))
os.environ["VFML_PREFETCH"] = "0"; os.environ["VFML_FLOW_BRANCH"] = "0"
net.clear_feature_cache()
ref = [proc.compute_optical_flow_resident(clip, i).clone() for i in order]
bad = 0
for p in range(passes):
    os.environ["VFML_PREFETCH"] = "1"; os.environ["VFML_FLOW_BRANCH"] = "1" if p >= passes // 2 else "0"
    net.clear_feature_cache()
    got = [proc.compute_optical_flow_resident(clip, i).clone() for i in order]
    torch.cuda.synchronize()
    for i, a, b in zip(order, ref, got):
        if not torch.equal(a, b):
            bad += 1
            print(f"pass {p} field {i}: {int((a != b).sum())} values differ, max {float((a - b).abs().max()):.3g} px", flush=True)
    if p % 5 == 4:
        print(f"pass {p + 1}/{passes}: {bad} fields differ so far", flush=True)
print(f"{passes * len(order)} fields compared, {bad} differ")
sys.exit(1 if bad else 0)
