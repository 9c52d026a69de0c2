"""Round 3: do the update block's convolutions run faster as three per-frame chains on three streams (graph branches)
than as one launch over the three centre frames?  Every layer between two temporal fusions is frame-independent; a chain
of dependent launches per frame would let one frame's epilogue / launch gap / prologue run beside the other frames' K loops
instead of all 512 resident workgroups reaching those phases together.  Timing only (random data).  GPU only.

    python tools/exp/frame_branches.py            # the four three-MFMA layers of an iteration, 12 iterations
"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd"))
sys.path.insert(0, ROOT)
import torch
from vfml import hip
from vfml.weights import pack_conv_weight

H8, W8, NF = 135, 240, 3
P = H8 * W8
dev = torch.device("cuda")
torch.manual_seed(0)
# (name, cin, cout, kh, kw, mfma): one iteration's launches of the mixed plan, in order (the motion encoder's flow half
# and the small kernels left out)
LAYERS = [("convc1", 672, 256, 1, 1, 3), ("convc2", 256, 192, 3, 3, 3), ("conv", 256, 128, 3, 3, 3),
          ("tprop*", 128, 128, 1, 1, "2a"),      # (stand-in: the fusion itself joins the frames)
          ("zr1", 384, 256, 1, 5, 1), ("q1", 384, 128, 1, 5, "2a"), ("zr2", 384, 256, 5, 1, 1), ("q2", 384, 128, 5, 1, 3),
          ("fh1", 128, 256, 3, 3, 3), ("fh2", 256, 36, 1, 1, 3)]
if os.environ.get("FB_SWEEP1X1", "0") == "1":      # what bounds the 1x1 motion-encoder layer: columns (MFMA work) or rows (bytes)?
    LAYERS = [(f"k{k}n{n}", k, n, 1, 1, 3) for k in (352, 672) for n in (64, 128, 192, 256)]
if os.environ.get("FB_ONLY", ""):
    LAYERS = [l for l in LAYERS if l[0] in os.environ["FB_ONLY"].split(",")]
if os.environ.get("FB_ONLY1X1", "0") == "1":
    LAYERS = [l for l in LAYERS if l[3] * l[4] == 1]
if os.environ.get("FB_ONLY3", "0") == "1":
    LAYERS = [l for l in LAYERS if l[5] == 3]


def weight(cin, cout, kh, kw, mfma):
    wt = torch.randn(cout * kh * kw * cin, device=dev) / math.sqrt(cin * kh * kw)
    c64 = mfma == 1 and cin % 64 == 0
    wc = pack_conv_weight(wt.reshape(cout, kh, kw, cin).permute(0, 3, 1, 2), cblock=64 if c64 else True)
    w = hip.SplitWeight(cout, wc.numel() // cout, dev).fill(wc, scale=hip.SplitWeight.auto_scale(float(wt.abs().max())))
    w.order = hip.KORDER_CBLOCK64 if c64 else hip.KORDER_CBLOCK
    return w


LD = 768
src = torch.empty(NF * P * LD, device=dev)
hip.to_s16(torch.randn(NF * P * LD, device=dev), NF * P, LD, LD, src, LD)
outs = [torch.empty(NF * P * 256, device=dev) for _ in LAYERS]
bias = torch.randn(256, device=dev)
W = [weight(*l[1:]) for l in LAYERS]


PROJ = os.environ.get("FB_PROJ", "0") == "1"       # fh1 with the projection epilogue (vfml_conv_desc.proj_out) instead of its store
if PROJ:
    w36 = torch.randn(36 * 256, device=dev) / 48.0
    W36 = hip.SplitWeight(36, 256, dev).fill(w36, scale=hip.SplitWeight.auto_scale(float(w36.abs().max())))
    W36.order = hip.KORDER_CBLOCK
    parts = torch.empty(2 * NF * P * 36, device=dev)


def chain(f0, nf, iters):
    for _ in range(iters):
        for (name, cin, cout, kh, kw, mfma), w, o in zip(LAYERS, W, outs):
            kw_ = dict(proj=W36, proj_out=parts, ld_proj=36) if PROJ and name == "fh1" else {}
            hip.conv2d(src, cin, LD, nf, H8, W8, w, bias, cout, kh, kw, o, 256, in0_off=f0 * P * LD, out_off=f0 * P * 256,
                       pad_h=kh // 2, pad_w=kw // 2, epilogue=hip.EPI_RELU, in_fmt=hip.FMT_S16, out_fmt=hip.FMT_S16, mfma=mfma,
                       **kw_)


side = [torch.cuda.Stream() for _ in range(NF - 1)]


def joint(iters):
    chain(0, NF, iters)


def branches(iters, join_every=1):
    """per-frame chains; the frames meet every `join_every` iterations (the temporal fusion)"""
    main = torch.cuda.current_stream()
    for _ in range(iters // join_every):
        fork = torch.cuda.Event()
        fork.record()
        joins = []
        for f, s in enumerate(side, start=1):
            with torch.cuda.stream(s):
                s.wait_event(fork)
                chain(f, 1, join_every)
                e = torch.cuda.Event()
                e.record()
                joins.append(e)
        chain(0, 1, join_every)
        for e in joins:
            main.wait_event(e)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


def graphed(fn):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()                       # warm
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            fn()
    torch.cuda.synchronize()
    return g.replay


ITERS = int(os.environ.get("FB_ITERS", 12))
if os.environ.get("FB_PER_LAYER", "0") == "1":      # every layer on its own: 12 back-to-back launches over the three frames
    ALL, ALLW, ALLO = LAYERS, W, outs
    tiles = os.environ.get("FB_TILES", "").split(";")        # VFML_DMA_TILE values to compare ("" = the dispatcher's choice)
    for i, l in enumerate(ALL):
        LAYERS, W, outs = [l], [ALLW[i]], [ALLO[i]]
        ref = None
        for tile in tiles:
            os.environ.pop("VFML_DMA_TILE", None)
            if tile:
                os.environ["VFML_DMA_TILE"] = tile
            ms = timed(graphed(lambda: joint(ITERS)))
            same = ""
            if ref is None:
                ref = outs[0].clone()
            else:
                same = "  same bits" if torch.equal(ref.view(torch.int32), outs[0].view(torch.int32)) else "  DIFFERENT"
            g = 2 * NF * P * l[1] * l[2] * l[3] * l[4] / 1e9
            # (cout columns of the 256-wide output rows: the rest is never written)
            chk = int(outs[0].view(NF * P, 256)[:, :l[2]].contiguous().view(torch.int32).to(torch.int64).sum())
            print(f"{l[0]:8s} cin {l[1]:4d} cout {l[2]:4d} {l[3]}x{l[4]} mfma {l[5]!s:3s} tile {tile or 'auto':8s} "
                  f"{ms / ITERS * 1e3:8.1f} us  {g * ITERS / ms:7.1f} TFLOP/s{same}  checksum {chk:x}", flush=True)
    sys.exit(0)
gf = sum(2 * NF * P * l[1] * l[2] * l[3] * l[4] for l in LAYERS) * ITERS / 1e9
print(f"{len(LAYERS)} layers x {ITERS} iterations, {gf:.0f} GFLOP", flush=True)
for label, fn in (("joint, eager", lambda: joint(ITERS)),
                  ("branches (join per iteration), eager", lambda: branches(ITERS)),
                  ("joint, graph", graphed(lambda: joint(ITERS))),
                  ("branches (join per iteration), graph", graphed(lambda: branches(ITERS))),
                  ("branches (never join), graph", graphed(lambda: branches(ITERS, ITERS)))):
    ms = timed(fn)
    print(f"{label:42s} {ms:8.3f} ms   {gf / ms:7.1f} TFLOP/s", flush=True)
