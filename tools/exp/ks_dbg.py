import sys, os, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd"))
from vfml import hip
g = torch.Generator().manual_seed(3)
P, S, D = 128, 2000, 4192
f1, f2 = torch.randn(P, D, generator=g) / 8, torch.randn(S, D, generator=g) / 8
ld = (S + 31) // 32 * 32
t = torch.empty(f1.numel(), device="cuda"); hip.to_s16(f1.cuda().reshape(-1), P, D, D, t, D)
w = hip.SplitWeight(S, D, torch.device("cuda")).fill(f2.cuda().reshape(-1).contiguous(), scale=16.0)
ref = (f1.double() @ f2.double().t()).float()
refh = (f1[:, :2112].double() @ f2[:, :2112].double().t()).float()
for ws in (None, torch.full((P * ld + S * P,), float("nan"), device="cuda")):
    out = torch.full((P * ld,), 7.0, device="cuda")
    hip.conv2d(t, D, D, 1, 1, P, w, None, S, 1, 1, out, ld, in_fmt=hip.FMT_S16, ksplit_ws=ws)
    got = out.view(P, ld).cpu()[:, :S]
    print("ws" if ws is not None else "no ws", "err vs full", ((got - ref).abs().max() / ref.abs().max()).item(),
          "vs first half", ((got - refh).abs().max() / ref.abs().max()).item(), "nan", torch.isnan(got).sum().item())
    if ws is not None:
        w2 = ws[:P * ld].view(P, ld).cpu()[:, :S]
        print("  ws vs second half", ((w2 - (ref - refh)).abs().max() / ref.abs().max()).item(), torch.isnan(w2).sum().item())
