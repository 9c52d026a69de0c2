"""Per-layer timing of the conv/GEMM launches of one 1080p field (dev tool, GPU only)."""
import sys, os, collections, contextlib, io, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd")); sys.path.insert(0, ROOT)
import torch
from vfml import hip, get_cfg, build_network
from vfml.weights import seeded_state_dict
from vfml.synth import synthetic_clip
import numpy as np

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1080, 1920)
cfg = get_cfg(); cfg.precision = prec
net = build_network(cfg); net.load_state_dict(seeded_state_dict(cfg, 0)); net.cuda().eval()
clip = torch.from_numpy(np.stack(synthetic_clip(5, H, W))).cuda()
net.forward_u8(clip); torch.cuda.synchronize()
recs = []
orig = hip.conv2d
def wrapped(in0, c0, ld0, n, h, w, weight, bias, cout, kh, kw, out, ldo, **kw_):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(in0, c0, ld0, n, h, w, weight, bias, cout, kh, kw, out, ldo, **kw_); e1.record()
    s = kw_.get("stride", 1)
    ho = (h + 2 * kw_.get("pad_h", 0) - kh) // s + 1; wo = (w + 2 * kw_.get("pad_w", 0) - kw) // s + 1
    ctot = c0 + kw_.get("c1", 0)
    recs.append(((n * ho * wo, cout, kh * kw * ctot, f"{kh}x{kw}s{s} c{ctot}->{cout}"), 2.0 * n * ho * wo * kh * kw * ctot * cout, e0, e1))
hip.conv2d = wrapped
import vfml.network as nw
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); net.forward_u8(clip); t1.record(); torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, fl, e0, e1 in recs:
    d = agg.setdefault(key, [0, 0.0, 0.0]); d[0] += 1; d[1] += fl; d[2] += e0.elapsed_time(e1)
tot = sum(d[2] for d in agg.values())
print(f"precision {prec}: step {t0.elapsed_time(t1):.1f} ms, conv launches {len(recs)}, conv total {tot:.1f} ms")
print(f"{'M':>9} {'cout':>6} {'K':>6}  {'layer':22s} {'calls':>5} {'ms':>8} {'TF/s':>7} {'%':>5}")
for (M, cout, K, name), (n, fl, ms) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print(f"{M:9d} {cout:6d} {K:6d}  {name:22s} {n:5d} {ms:8.2f} {fl/ms/1e9:7.1f} {100*ms/tot:5.1f}")
