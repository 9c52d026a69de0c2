"""Where do the disturbed window samples of two_stream_lookup.py come from?  Integer coordinates (every output IS one texel)
and f32 output, so a wrong value can be searched for in the volume."""
import sys, os, math, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd")); sys.path.insert(0, ROOT)
from vfml import hip
g = torch.Generator(device="cuda").manual_seed(4)
h, w, R, L = 135, 240, 4, 4
P = h * w
hl = [h >> l for l in range(L)]; wl = [w >> l for l in range(L)]
ld = [(a * b + 31) // 32 * 32 for a, b in zip(hl, wl)]
vol = [torch.randn(P * l, device="cuda", generator=g) for l in ld]
coords = (torch.rand(P, 4, device="cuda", generator=g) * torch.tensor([w, h, w, h], device="cuda")).floor()
if os.environ.get("DIAG_FRAC", "0") == "1":
    coords = (coords / 8).floor() * 8 + 4      # x.5 at level 3, integer below
elif os.environ.get("DIAG_FRAC", "0") == "2":
    coords = coords + 0.37
else:
    coords = (coords / 8).floor() * 8          # multiples of 8: integer at every level
coords = coords.reshape(-1).contiguous()
nch = 336
out = torch.zeros(P * nch, device="cuda")
fmt = hip.FMT_S16 if os.environ.get("DIAG_S16", "0") == "1" else hip.FMT_F32
def lookup(): hip.corr_lookup(vol, hl, wl, ld, R, P, coords, 0, 4, out, 0, nch, out_fmt=fmt)
f1 = torch.randn(32400 * 256, device="cuda"); rows = torch.empty_like(f1); hip.to_s16(f1, 32400, 256, 256, rows, 256, scale=16.0)
wg = hip.SplitWeight(8040, 256, torch.device("cuda")).fill(torch.randn(8040 * 256, device="cuda"), scale=16.0)
og = torch.zeros(32400 * 8064, device="cuda")
agg = lambda: hip.conv2d(rows, 256, 256, 1, 1, 32400, wg, None, 8040, 1, 1, og, 8064, in_fmt=hip.FMT_S16)
side = torch.cuda.Stream()
out.zero_(); lookup(); torch.cuda.synchronize(); ref = out.clone()
# sanity: the reference itself is the gathered texels
agg(); torch.cuda.synchronize()
shown = 0; nbad = 0
for it in range(20):
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(8): agg()
    outs = []
    for _ in range(10):
        out.zero_(); lookup(); outs.append(out.clone())
    torch.cuda.synchronize()
    for o in outs:
        d = (o != ref).nonzero().flatten()
        if d.numel() == 0: continue
        nbad += 1
        if shown >= 3: continue
        shown += 1
        qs = (d // nch); cs = d % nch
        print(f"run: {d.numel()} words differ over {qs.unique().numel()} queries; query range {int(qs.min())}..{int(qs.max())}")
        uq = qs.unique()
        print("  queries:", uq[:40].tolist())
        print("  blocks (q//4):", (uq // 4).unique()[:40].tolist())
        print("  channels histogram by level:", [int(((cs >= 81 * l) & (cs < 81 * (l + 1))).sum()) for l in range(4)], "pad:", int((cs >= 324).sum()))
        rem = cs % 81
        print("  i histogram:", [int(((rem // 9) == t).sum()) for t in range(9)], " j histogram:", [int(((rem % 9) == t).sum()) for t in range(9)])
        for k in range(min(12, d.numel())):
            q, c = int(qs[k]), int(cs[k])
            if fmt != hip.FMT_F32 or c >= 324:
                print(f"   q={q} word={c} got={float(o[d[k]]):.6g} exp={float(ref[d[k]]):.6g}"); continue
            l = c // 81; rem = c % 81; i, j = rem // 9, rem % 9
            got, exp = float(o[d[k]]), float(ref[d[k]])
            row = vol[l][q * ld[l]:(q + 1) * ld[l]]
            where = (row == got).nonzero().flatten().tolist()
            cx, cy = float(coords[q * 4]) / (1 << l), float(coords[q * 4 + 1]) / (1 << l)
            ex, ey = int(cx) - R + i, int(cy) - R + j
            # other rows / levels
            if os.environ.get("DIAG_FRAC", "0") != "0":
                fx, fy = cx - math.floor(cx), cy - math.floor(cy)
                ex, ey = math.floor(cx) - R + i, math.floor(cy) - R + j
                W = [(1 - fx) * (1 - fy), fx * (1 - fy), (1 - fx) * fy, fx * fy]
                T = []
                for (dx, dy) in ((0, 0), (1, 0), (0, 1), (1, 1)):
                    X, Y = ex + dx, ey + dy
                    T.append(float(row[Y * wl[l] + X]) if 0 <= X < wl[l] and 0 <= Y < hl[l] else 0.0)
                mine = sum(a * b for a, b in zip(W, T))
                repl = [(T[t] + (got - exp) / W[t]) if W[t] else float("nan") for t in range(4)]
                print(f"   q={q} c={c} level={l} (i={i},j={j}) f=({fx:.3f},{fy:.3f}) T={['%.5g' % t for t in T]} got={got:.6g} exp={exp:.6g} host={mine:.6g} "
                      f"replaced-texel candidates={['%.5g' % t for t in repl]}")
                continue
            hit = []
            if not where and got != 0.0:
                for l2 in range(4):
                    m = (vol[l2] == got).nonzero().flatten()
                    if m.numel():
                        hit.append((l2, int(m[0]) // ld[l2], int(m[0]) % ld[l2]))
            print(f"   q={q} c={c} level={l} (i={i},j={j}) expects texel ({ex},{ey}) idx {ey * wl[l] + ex}: got={got:.6g} exp={exp:.6g} "
                  f"got found in own row at {where[:4]} elsewhere {hit[:2]}")
print(f"{nbad}/200 launches differ")
