"""Would the lookup be faster on a volume whose per-query level images are stored in small 2-D tiles (a 10 x 10 window then
touches fewer 64 / 128-byte lines than ten 40-byte row segments)?  Timing only: random volumes, vfml_corr_lookup's vol_tile
(tile = 2^tws x 2^ths texels; the volume's rows are in the tile order of the query grid as well).

    python tools/exp/lookup_tiled.py"""
import sys, os, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd")); sys.path.insert(0, ROOT)
from vfml import hip
g = torch.Generator(device="cuda").manual_seed(4)
h, w, R, L = 135, 240, 4, 4
P = h * w
hl = [h >> l for l in range(L)]; wl = [w >> l for l in range(L)]
nch = 336
NM = 3
out = torch.zeros(NM * P * nch, device="cuda")
# coordinates like the engine's: the query's own position plus a smooth flow of a few pixels
ys, xs = torch.meshgrid(torch.arange(h, device="cuda", dtype=torch.float32), torch.arange(w, device="cuda", dtype=torch.float32), indexing="ij")
base = torch.stack([xs, ys, xs, ys], -1).reshape(P, 4)
coords = (base.repeat(NM, 1) + 6.0 * torch.randn(NM * P, 4, device="cuda", generator=g)).reshape(-1).contiguous()
for fmt, fname, es in ((hip.FMT_F32, "f32 volume", 4), (hip.FMT_F16, "f16 volume", 2)):
    for tws, ths in ((0, 0), (3, 2), (2, 2), (4, 1), (3, 1), (2, 3), (3, 3), (4, 2)):
        TW, TH = 1 << tws, 1 << ths
        ld = [(((a + TH - 1) // TH * TH) * ((b + TW - 1) // TW * TW) + 31) // 32 * 32 for a, b in zip(hl, wl)]
        # a tiled volume has one ROW per tile position of the query grid too (query q reads row tile_position(q)): whole tiles
        rows = ((h + TH - 1) // TH * TH) * ((w + TW - 1) // TW * TW)
        assert rows >= P and all(l >= ((a + TH - 1) // TH * TH) * ((b + TW - 1) // TW * TW) for l, a, b in zip(ld, hl, wl))
        vols = []
        for m in range(NM):
            vols.append([torch.empty(rows * l, device="cuda", dtype=torch.float32 if es == 4 else torch.float16).normal_() for l in ld])
        def go():
            hip.corr_lookup(vols, hl, wl, ld, R, P, coords, 0, 4, out, 0, nch, out_fmt=hip.FMT_S16, vol_fmt=fmt, vol_tile=tws + 16 * ths)
        for _ in range(3): go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): go()
        e1.record(); torch.cuda.synchronize()
        print(f"{fname} tile {TW:2d} x {TH}: {1000 * e0.elapsed_time(e1) / 20:7.1f} us per launch ({NM} maps x {P} queries)", flush=True)
        del vols
