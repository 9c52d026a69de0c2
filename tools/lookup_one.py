"""One 1080p-sized correlation lookup launch, repeated (target for rocprofv3 --pmc). GPU only."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd")); sys.path.insert(0, ROOT)
import torch
from vfml import hip
h, w, L, R, M = 135, 240, 4, 4, 3
P = h * w
hl, wl = [h], [w]
for l in range(1, L):
    hl.append(hl[-1] // 2); wl.append(wl[-1] // 2)
# the engine's layout: level images and volume rows in 4 x 8 tiles (LOOKUP_ROW_MAJOR=1: row-major)
vt = None if os.environ.get("LOOKUP_ROW_MAJOR") == "1" else hip.VolTile(2, 3)
Sl = [vt.count(a, b) if vt else a * b for a, b in zip(hl, wl)]
rows = vt.count(h, w) if vt else P          # a tiled volume has a row per tile position of the query grid
ld = [(s + 31) // 32 * 32 for s in Sl]
ld = [n if (n // 32) % 2 else n + 32 for n in ld]
TILE = vt.code if vt else 0
assert rows >= P and all(l >= s for l, s in zip(ld, Sl))
maps = [[torch.randn(rows * ld[l], device="cuda") for l in range(L)] for _ in range(M)]
coords = torch.empty(M * P * 4, device="cuda")
hip.coords_init(coords, M, h, w)
coords += torch.randn_like(coords) * 2.0
out = torch.empty(M * P * 656, device="cuda")
for fmt in (hip.FMT_S16, hip.FMT_F32):
    for _ in range(3):
        hip.corr_lookup(maps, hl, wl, ld, R, P, coords, 0, 4, out, 0, 656, out_fmt=fmt, vol_tile=TILE)
torch.cuda.synchronize()
alg = M * P * (L * 100 * 4 + L * 81 * 4)
for name, fmt in (("S16", hip.FMT_S16), ("F32", hip.FMT_F32)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        hip.corr_lookup(maps, hl, wl, ld, R, P, coords, 0, 4, out, 0, 656, out_fmt=fmt, vol_tile=TILE)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"lookup[{name}] {M*P} queries: {ms*1000:.1f} us, algorithmic {alg/1e6:.1f} MB -> {alg/ms/1e9:.2f} TB/s")
