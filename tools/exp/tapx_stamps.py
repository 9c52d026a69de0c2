#!/usr/bin/env python3
"""Where a wave's cycles go in conv_gemm_tapx_kernel (experiment build, GPU only): rebuilds conv_gemm_tapx.hip with
`-include tools/exp/tapx_hooks.h` (the in-kernel stamps live there, not in the product source) into tools/exp/libvfml_stamps.so (the other objects are the shipped ones), runs one 1080p update-block
shape and prints, per wave of one mid-grid workgroup, the mean cycles per K step spent (X) reading the weight fragments
and waiting at the barrier that frees the weight stage, (M) in the MFMA phase (LDS-DMA issues included), (W) waiting for
the step's pieces (vmcnt), (Y) at the barrier that closes the step.

    python tools/exp/tapx_stamps.py [kh kw cin cout]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "video-flow-ml_amd", "vfml", "csrc")
TAG = next((a[6:] for a in sys.argv if a.startswith("--tag=")), "")
LIB = os.path.join(ROOT, "tools", "exp", f"libvfml_stamps{TAG}.so")


def build():
    obj = os.path.join(ROOT, "tools", "exp", f"tapx_stamps{TAG}.o")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-include",
                    os.path.join(ROOT, "tools", "exp", "tapx_hooks.h")] + [f for f in sys.argv if f.startswith("-D")] + ["-c",
                    os.path.join(CSRC, "conv_gemm_tapx.hip"), "-o", obj], check=True)
    others = [os.path.join(CSRC, "_obj", f) for f in sorted(os.listdir(os.path.join(CSRC, "_obj")))
              if f.endswith(".o") and not f.startswith("conv_gemm_tapx")]
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB, obj] + others, check=True)


if __name__ == "__main__":
    if not os.path.exists(LIB) or "--build" in sys.argv or "--build-only" in sys.argv:
        build()
    if "--build-only" in sys.argv:
        sys.exit(0)
    os.environ["VFML_LIB"] = LIB
    sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    from vfml import hip
    import conv_microbench as mb
    args = [int(v) for v in sys.argv[1:] if v.lstrip("-").isdigit()]
    kh, kw, cin, cout = args if len(args) == 4 else (1, 5, 512, 256)
    mb.bench(f"{kh}x{kw} c{cin}->{cout}", 3, 135, 240, cin, cout, kh, kw, s16=True, reps=40)
    out = (ctypes.c_ulonglong * 64)()
    assert hip.lib().vfml_debug_tapx_stamps(out) == 0
    for w in range(4):
        n = out[w * 8 + 4] or 1
        x, m, wt, y = (out[w * 8 + k] / n for k in range(4))
        print(f"wave {w}: {n} steps; per step: weights+barrier X {x:7.0f}  MFMA phase {m:7.0f}  vmcnt wait {wt:6.0f}  barrier Y {y:6.0f}"
              f"  total {x + m + wt + y:7.0f} cycles; K loop {out[w * 8 + 5]} shader clocks in {out[w * 8 + 6] / 100.0:.1f} us"
              f" = {out[w * 8 + 5] / max(1, out[w * 8 + 6]) * 0.1:.3f} GHz")
    tl = (ctypes.c_ulonglong * (4096 * 4))()
    assert hip.lib().vfml_debug_tapx_timeline(tl) == 0
    rows = [(tl[4 * b], tl[4 * b + 1], tl[4 * b + 2], tl[4 * b + 3], b) for b in range(4096) if tl[4 * b]]
    t0 = min(r[0] for r in rows)
    print(f"timeline of the last launch ({len(rows)} workgroups; us from the first entry): block: entry, K loop start, K loop end, exit")
    import statistics
    for r in sorted(rows)[::max(1, len(rows) // 24)]:
        print(f"  block {r[4]:5d}: {(r[0] - t0) / 100:7.1f} {(r[1] - t0) / 100:7.1f} {(r[2] - t0) / 100:7.1f} {(r[3] - t0) / 100:7.1f}")
    print("  mean us: entry->K", statistics.mean((r[1] - r[0]) / 100 for r in rows), " K loop", statistics.mean((r[2] - r[1]) / 100 for r in rows),
          " epilogue", statistics.mean((r[3] - r[2]) / 100 for r in rows), " last exit", max(r[3] - t0 for r in rows) / 100)
