"""Finding of round 2 (profiles/r02_kernel_anatomy.md section 7): vfml_corr_lookup returned slightly different windows for a
few dozen of 32 400 queries (pyramid levels 2-3) while one of this library's MFMA convolution / GEMM kernels ran on ANOTHER
stream - a packed-f32 multiply reading a ds_read2 result straight behind its s_waitcnt (two_stream_lookup_diag.py shows
which operand; fixed by `bilinear4` + -fno-slp-vectorize for flow_ops.hip).  This is the reproducer, now a regression check:
every line must say 0/100.

    python tools/exp/two_stream_lookup.py
    VFML_LIB=<a library built with -DVFML_LOOKUP_OLD_ARITH> python tools/exp/two_stream_lookup.py     # the old behaviour"""
import sys, os, math, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd")); sys.path.insert(0, ROOT)
from vfml import hip
from vfml.weights import pack_conv_weight
g = torch.Generator(device="cuda").manual_seed(4)
h, w, R, L = 135, 240, 4, 4
P = h * w
hl = [h >> l for l in range(L)]; wl = [w >> l for l in range(L)]
ld = [(a * b + 31) // 32 * 32 for a, b in zip(hl, wl)]
vol = [torch.randn(P * l, device="cuda", generator=g) for l in ld]
coords = (torch.rand(P, 4, device="cuda", generator=g) * torch.tensor([w, h, w, h], device="cuda")).reshape(-1)
nch = 336
out = torch.zeros(P * nch, device="cuda")
def lookup(fmt): hip.corr_lookup(vol, hl, wl, ld, R, P, coords, 0, 4, out, 0, nch, out_fmt=fmt)
def conv(cin, cout, k, H, W, s16=True, per_tap=True, mfma=3):
    x = torch.randn(H * W * cin, device="cuda")
    if s16:
        x16 = torch.empty_like(x); hip.to_s16(x, H * W, cin, cin, x16, cin); x = x16
    wt = torch.randn(cout, cin, k, k, device="cuda") / math.sqrt(cin * k * k)
    wc = pack_conv_weight(wt, cblock=True) if s16 else pack_conv_weight(wt)
    wa = hip.SplitWeight(cout, wc.numel() // cout, x.device).fill(wc.cuda(), scale=hip.SplitWeight.auto_scale(float(wt.abs().max())))
    wa.order = hip.KORDER_CBLOCK if s16 else hip.KORDER_TAP
    o = torch.empty(H * W * cout, device="cuda")
    fmt = hip.FMT_S16 if s16 else hip.FMT_F32
    return lambda: hip.conv2d(x, cin, cin, 1, H, W, wa, None, cout, k, k, o, cout, pad_h=k // 2, pad_w=k // 2, in_fmt=fmt, per_tap=per_tap)
aggs = {
    "dma 1x1 c256->256 (no masked taps)": conv(256, 256, 1, 405, 240),
    "dma 3x3 c256->256": conv(256, 256, 3, 405, 240),
    "register-staged 3x3 c64 (f32 src, no LDS-DMA)": conv(64, 64, 3, 540, 960, s16=False),
    "persistent GEMM (LDS-DMA) 32400x8040x256": None,
}
f1 = torch.randn(32400 * 256, device="cuda"); rows = torch.empty_like(f1); hip.to_s16(f1, 32400, 256, 256, rows, 256, scale=16.0)
wg = hip.SplitWeight(8040, 256, torch.device("cuda")).fill(torch.randn(8040 * 256, device="cuda"), scale=16.0)
og = torch.zeros(32400 * 8064, device="cuda")
aggs["persistent GEMM (LDS-DMA) 32400x8040x256"] = lambda: hip.conv2d(rows, 256, 256, 1, 1, 32400, wg, None, 8040, 1, 1, og, 8064, in_fmt=hip.FMT_S16)
side = torch.cuda.Stream()
for fmt, fname in ((hip.FMT_S16, "S16 out"), (hip.FMT_F32, "f32 out")):
    out.zero_(); lookup(fmt); torch.cuda.synchronize(); ref = out.clone()
    for name, fn in aggs.items():
        fn(); torch.cuda.synchronize(); bad = 0
        for it in range(10):
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                for _ in range(8): fn()
            for _ in range(10):
                out.zero_(); lookup(fmt); bad += int(not torch.equal(out, ref))
            torch.cuda.synchronize()
        print(f"lookup ({fname}) beside {name:48s}: {bad}/100 differ")
