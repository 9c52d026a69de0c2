#!/usr/bin/env python3
"""Per-layer precision sensitivity of one flow field (dev tool, GPU only): for every layer of the network, the mean
end-point error the picked flow takes when THAT layer alone runs with 2 or 1 MFMAs per product instead of 3
(cfg.precision='mixed', cfg.mfma_plan={layer: n}), against the all-3 field, plus the time the field then takes.
Then a greedy plan: cheapest error per saved millisecond first, until the budget is spent; the combined plan is
re-measured (errors do not add exactly).

    python tools/precision_plan.py [--height 1080 --width 1920 --seq 5 --seed 0 --budget 5e-5] [--json out.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from vfml import build_network, get_cfg  # noqa: E402
from vfml.synth import synthetic_clip  # noqa: E402
from vfml.weights import seeded_state_dict  # noqa: E402

UB = "update_block"
LAYERS = [f"{UB}.encoder.convc1", f"{UB}.encoder.convc2", f"{UB}.encoder.convf1", f"{UB}.encoder.convf2",
          f"{UB}.encoder.conv", f"{UB}.tprop",
          f"{UB}.gru.convzr1.iter", f"{UB}.gru.convq1.iter", f"{UB}.gru.convzr2.iter", f"{UB}.gru.convq2.iter",
          f"{UB}.gru.convzr1.ctx", f"{UB}.gru.convq1.ctx", f"{UB}.gru.convzr2.ctx", f"{UB}.gru.convq2.ctx",
          f"{UB}.flow_head.conv1", f"{UB}.flow_head.conv2", f"{UB}.mask.0", f"{UB}.mask.2",
          "fnet", "cnet", "corr"]


# Candidate plans for a plain-f16-grade arithmetic that stays inside the 1e-3 px contract (BASELINE config 5, "fp16"): one MFMA
# per product wherever the rounding of a WEIGHT does not reach the flow linearly; the context encoder (1.4e-3 px on its own at
# 1080p), the per-frame context parts of the gates and the feature encoder keep all three terms (they run once per frame, not
# per iteration); the layers on the linear flow path take their activations as plain f16 and keep the weights' lo half ("2a").
_KEEP3 = {"cnet": 3, "fnet": 3, f"{UB}.gru.convzr1.ctx": 3, f"{UB}.gru.convq1.ctx": 3, f"{UB}.gru.convzr2.ctx": 3,
          f"{UB}.gru.convq2.ctx": 3}
CANDIDATES = {
    "all-1": {"": 1},
    "all-1, cnet 3": {"": 1, "cnet": 3},
    "all-1, encoders+ctx 3": {"": 1, **_KEEP3},
    "P1: + conv, fh1 at 2a": {"": 1, **_KEEP3, f"{UB}.encoder.conv": "2a", f"{UB}.flow_head.conv1": "2a"},
    "P2: + convc2, fh2 at 2a": {"": 1, **_KEEP3, f"{UB}.encoder.conv": "2a", f"{UB}.flow_head.conv1": "2a",
                               f"{UB}.encoder.convc2": "2a", f"{UB}.flow_head.conv2": "2a"},
    "P3: P2 + convc1, q2 at 2a": {"": 1, **_KEEP3, f"{UB}.encoder.conv": "2a", f"{UB}.flow_head.conv1": "2a",
                                 f"{UB}.encoder.convc2": "2a", f"{UB}.flow_head.conv2": "2a",
                                 f"{UB}.encoder.convc1": "2a", f"{UB}.gru.convq2.iter": "2a"},
    "P4: P3, fnet at 1": {"": 1, **_KEEP3, "fnet": 1, f"{UB}.encoder.conv": "2a", f"{UB}.flow_head.conv1": "2a",
                          f"{UB}.encoder.convc2": "2a", f"{UB}.flow_head.conv2": "2a",
                          f"{UB}.encoder.convc1": "2a", f"{UB}.gru.convq2.iter": "2a"},
    "MOF default mixed plan": None,       # filled in main()
}


def main():
    from vfml.cfg import DEFAULT_MIXED_PLAN
    CANDIDATES["MOF default mixed plan"] = dict(DEFAULT_MIXED_PLAN)
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--seq", type=int, default=5)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--budget", type=float, default=5e-5, help="mean EPE (px) the greedy plan may spend")
    ap.add_argument("--json", default=None)
    ap.add_argument("--arch", default="mof", choices=["mof", "bof"], help="bof: the tri-frame network (centre triple of the window)")
    ap.add_argument("--candidates", action="store_true",
                    help="skip the per-layer sweep: measure the named candidate plans (CANDIDATES) against the all-3 field")
    args = ap.parse_args()

    cfg = get_cfg()
    cfg.precision = "mixed"
    cfg.mfma_plan = {}
    if args.arch == "bof":
        cfg.network = "BOFNet"
    net = build_network(cfg)
    net.load_state_dict(seeded_state_dict(cfg, args.seed))
    net.cuda().eval()
    frames = synthetic_clip(args.seq + 3, args.height, args.width)
    clip = torch.from_numpy(np.stack(frames)).cuda()

    def field(plan, start=0):
        cfg.mfma_plan = dict(plan)
        net.clear_feature_cache()
        f, _ = net.forward_u8(clip[start:start + args.seq], return_lowres=False, pick_only=True)
        # (pick_only: [1, 1, 2, H, W]; the tri-frame network returns both flows - the reference's pick is shape[1] // 2)
        return f[0, f.shape[1] // 2].permute(1, 2, 0).clone()

    def timed(plan):
        """ms per field in the sliding steady state (encoders / pyramids of the overlap cached, as in a job)."""
        cfg.mfma_plan = dict(plan)
        net.clear_feature_cache()
        keys = [("t", i) for i in range(args.seq + 3)]
        ts = []
        for s in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            net.forward_u8(clip[s:s + args.seq], return_lowres=False, pick_only=True, frame_keys=keys[s:s + args.seq])
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return min(ts[2:])

    def epe(a, b):
        return float((a - b).pow(2).sum(-1).sqrt().mean())

    ref = field({})
    t_ref = timed({})
    print(f"{args.arch} {args.width}x{args.height} seq {args.seq} seed {args.seed}: all-3 field {t_ref:.2f} ms (steady state), "
          f"|flow| mean {float(ref.abs().mean()):.3f} px")
    if args.candidates:
        out = {}
        for name, plan in CANDIDATES.items():
            e, t = epe(field(plan), ref), timed(plan)
            out[name] = {"plan": plan, "epe_vs_all3": e, "ms": t}
            print(f"  {name:28s}: dEPE {e:.3e} px vs all-3, field {t:7.2f} ms (all-3 {t_ref:.2f})", flush=True)
        if args.json:
            with open(args.json, "w") as f:
                json.dump({"args": vars(args), "t_ref_ms": t_ref, "candidates": out}, f, indent=1)
        return
    rows = []
    for layer in LAYERS:
        for n in (2, "2a", 1):
            if layer == "corr" and n != 1:
                continue
            e = epe(field({layer: n}), ref)
            t = timed({layer: n})
            rows.append({"layer": layer, "mfma": n, "epe": e, "ms": t, "saved_ms": t_ref - t})
            print(f"  {layer:34s} mfma {n!s:>2}: dEPE {e:.3e} px  field {t:7.2f} ms  saved {t_ref - t:6.2f} ms", flush=True)

    # greedy: per layer the options are ("2a", 2, 1); take steps in order of error per saved ms
    plan, spent, saved = {}, 0.0, {}
    opts = sorted([r for r in rows if r["saved_ms"] > 0.02], key=lambda r: r["epe"] / r["saved_ms"])
    for r in opts:
        cur = plan.get(r["layer"])
        if cur is not None and saved[r["layer"]] >= r["saved_ms"]:
            continue
        prev = next((q["epe"] for q in rows if q["layer"] == r["layer"] and q["mfma"] == cur), 0.0)
        if spent - prev + r["epe"] > args.budget:
            continue
        spent += r["epe"] - prev
        plan[r["layer"]] = r["mfma"]
        saved[r["layer"]] = r["saved_ms"]
    e_plan = epe(field(plan), ref)
    t_plan = timed(plan)
    print(f"greedy plan (budget {args.budget:.1e}, sum of single-layer errors {spent:.2e}): {json.dumps(plan)}")
    print(f"  combined: dEPE {e_plan:.3e} px vs all-3, field {t_plan:.2f} ms (all-3 {t_ref:.2f} ms)")
    ub2a = {UB: "2a", "corr": 1}
    print(f"  update block '2a' + corr 1: dEPE {epe(field(ub2a), ref):.3e} px, field {timed(ub2a):.2f} ms")
    for name, p in (("all-2w", {"": 2}), ("all-2a", {"": "2a"}), ("all-1", {"": 1})):
        print(f"  {name}: dEPE {epe(field(p), ref):.3e} px, field {timed(p):.2f} ms")
    if args.json:
        with open(args.json, "w") as f:
            json.dump({"args": vars(args), "t_ref_ms": t_ref, "rows": rows, "plan": plan, "plan_epe": e_plan,
                       "plan_ms": t_plan}, f, indent=1)


if __name__ == "__main__":
    main()
