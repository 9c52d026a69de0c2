"""Which pyramid levels may be stored as f16 inside the mixed plan's EPE budget (dev tool, GPU only): for every
cfg.corr_volume setting, mean / max EPE at 1920x1080 of the mixed plan against the engine's exact-f32 arithmetic, three
weight seeds x T in {3, 5} (the protocol of tests/test_gpu_e2e.py::test_mixed_plan_stays_within_its_budget_at_1080p).

    python tools/corr_volume_plan.py [settings ...]        default: f32 f16@3 f16@2 f16@1 f16"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vfml import build_network, get_cfg  # noqa: E402
from vfml.cfg import DEFAULT_MIXED_PLAN  # noqa: E402
from vfml.synth import synthetic_clip  # noqa: E402
from vfml.weights import seeded_state_dict  # noqa: E402

settings = sys.argv[1:] or ["f32", "f16@3", "f16@2", "f16@1", "f16"]
H, W = 1080, 1920
clip = torch.from_numpy(np.stack(synthetic_clip(5, H, W))).cuda()
rows = {s: [] for s in settings}
for seed in (0, 1, 2):
    sd = seeded_state_dict(get_cfg(), seed)

    def net(prec, vol):
        c = get_cfg()
        c.precision, c.corr_volume = prec, vol
        if prec == "mixed":
            c.mfma_plan = dict(DEFAULT_MIXED_PLAN)
        n = build_network(c)
        n.load_state_dict(sd)
        return n.cuda().eval()

    ref_net = net("f32", "f32")
    refs = {}
    for T in (3, 5):
        win = clip[1:4] if T == 3 else clip
        r = ref_net.forward_u8(win, return_lowres=False)[0]
        refs[T] = r[0, r.shape[1] // 2].permute(1, 2, 0).cpu()
    ref_net.release_workspace()
    del ref_net
    torch.cuda.empty_cache()
    for s in settings:
        n = net("mixed", s)
        for T in (3, 5):
            win = clip[1:4] if T == 3 else clip
            g = n.forward_u8(win, return_lowres=False)[0]
            g = g[0, g.shape[1] // 2].permute(1, 2, 0).cpu()
            e = (g - refs[T]).pow(2).sum(-1).sqrt()
            rows[s].append((seed, T, float(e.mean()), float(e.max())))
            print(f"corr_volume {s:6s} seed {seed} T={T}: mean EPE {float(e.mean()):.3e} px, max {float(e.max()):.3e} px", flush=True)
        n.release_workspace()
        del n
        torch.cuda.empty_cache()
print("\n| corr_volume | worst mean EPE (px) | mean of means | worst max |")
print("|---|---|---|---|")
for s in settings:
    m = [r[2] for r in rows[s]]
    print(f"| {s} | {max(m):.3e} | {sum(m) / len(m):.3e} | {max(r[3] for r in rows[s]):.3e} |")
