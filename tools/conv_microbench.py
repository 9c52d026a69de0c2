"""Micro-benchmark of vfml_conv2d[_split] on chosen shapes (dev tool, GPU only)."""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd")); sys.path.insert(0, ROOT)
import torch
from vfml import hip

def bench(name, n, h, w, cin, cout, kh, kw, ld=None, prec="f16x3", reps=20, s16=False, cblock=None, gemm=False, stride=1, f32_out=False):
    cblock = (os.environ.get("MB_CBLOCK", "1") == "1") if cblock is None else cblock
    ld = ld or cin
    x = torch.randn(n * h * w * ld, device="cuda")
    fmt = hip.FMT_S16 if s16 else hip.FMT_F32
    if s16:
        x16 = torch.empty_like(x)
        hip.to_s16(x, n * h * w, ld, ld, x16, ld)
        x = x16
        name += " [S16]"
    wt = torch.randn(cout * kh * kw * cin, device="cuda") / math.sqrt(cin * kh * kw)
    b = None if os.environ.get("MB_NOBIAS") else torch.randn(cout, device="cuda")     # MB_NOBIAS: no global load in the epilogue
    mf = int(os.environ.get("MB_MFMA", "3"))      # 1: one MFMA per product (64-channel steps where the shape allows)
    h64 = mf == 1 and s16 and cblock and cin % 64 == 0 and kh * kw > 1 and not gemm
    if prec != "f32" and s16 and cblock:
        from vfml.weights import pack_conv_weight
        wc = pack_conv_weight(wt.reshape(cout, kh, kw, cin).permute(0, 3, 1, 2), cblock=64 if h64 else True)
        wobj = hip.SplitWeight(cout, wc.numel() // cout, x.device).fill(wc, scale=hip.SplitWeight.auto_scale(float(wt.abs().max())))
        wobj.order = hip.KORDER_CBLOCK64 if h64 else hip.KORDER_CBLOCK
        name += " cb64" if h64 else " cb"
    else:
        wobj = wt if prec == "f32" else hip.SplitWeight(cout, kh * kw * cin, x.device).fill(wt, scale=hip.SplitWeight.auto_scale(float(wt.abs().max())))
    out = torch.empty(n * h * w * cout * (2 if os.environ.get("MB_DUP") else 1), device="cuda")
    def run():
        if gemm:     # as the correlation volume is built: no bias, no activation, plain f32 out
            hip.conv2d(x, cin, ld, n, h, w, wobj, None, cout, kh, kw, out, cout, out_scale=1.0 / 16.0, in_fmt=fmt)
        else:
            hip.conv2d(x, cin, ld, n, h, w, wobj, b, cout, kh, kw, out, cout, stride=stride, pad_h=kh // 2, pad_w=kw // 2,
                       epilogue=hip.EPI_NONE if f32_out else hip.EPI_RELU, in_fmt=fmt, out_fmt=hip.FMT_F32 if f32_out else fmt,
                       mfma=mf if (mf == 3 or (s16 and cblock)) else 3)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * n * ((h - 1) // stride + 1) * ((w - 1) // stride + 1) * kh * kw * cin * cout
    print(f"{name:46s} M={n*h*w:8d} K={kh*kw*cin:5d} cout={cout:5d}  {ms*1000:8.1f} us  {fl/ms/1e9:7.1f} TF/s")

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "enc":
    # the residual-block convs of one 1080p frame's encoder: float32 rows split while staged (shipped) vs split rows
    for s16 in (False, True):
        bench("layer1 3x3 c64->64 540x960", 1, 540, 960, 64, 64, 3, 3, s16=s16, f32_out=True)
        bench("layer2 3x3 c64->96 s2", 1, 540, 960, 64, 96, 3, 3, s16=s16, f32_out=True, stride=2)
        bench("layer2 1x1 c64->96 s2", 1, 540, 960, 64, 96, 1, 1, s16=s16, f32_out=True, stride=2)
        bench("layer2 3x3 c96->96 270x480", 1, 270, 480, 96, 96, 3, 3, s16=s16, f32_out=True)
        bench("layer3 3x3 c96->128 s2", 1, 270, 480, 96, 128, 3, 3, s16=s16, f32_out=True, stride=2)
        bench("layer3 3x3 c128->128 135x240", 1, 135, 240, 128, 128, 3, 3, s16=s16, f32_out=True)
        bench("conv2 1x1 c128->256 135x240", 1, 135, 240, 128, 256, 1, 1, s16=s16, f32_out=True)
    x = torch.randn(540 * 960 * 64, device="cuda"); y = torch.empty_like(x)
    for _ in range(3): hip.to_s16(x, 540 * 960, 64, 64, y, 64)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): hip.to_s16(x, 540 * 960, 64, 64, y, 64)
    e1.record(); torch.cuda.synchronize()
    print(f"to_s16 540x960x64: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
    sys.exit(0)

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "s16":
    bench("gru 1x5 c512->256", 3, 135, 240, 512, 256, 1, 5, s16=True)
    bench("gru 1x5 c512->256 ld768", 3, 135, 240, 512, 256, 1, 5, ld=768, s16=True)
    bench("1x5 c512->128", 3, 135, 240, 512, 128, 1, 5, s16=True)
    bench("3x3 c256->256", 3, 135, 240, 256, 256, 3, 3, s16=True)
    bench("3x3 c256->192", 3, 135, 240, 256, 192, 3, 3, s16=True)
    bench("1x1 c656->256", 3, 135, 240, 656, 256, 1, 1, s16=True)
    if os.environ.get("MB_MFMA", "3") == "3":
        bench("gemm 32400x32400x256", 1, 1, 32400, 256, 32400, 1, 1, reps=5, s16=True, gemm=True)
        bench("gemm 32400x8040x256", 1, 1, 32400, 256, 8040, 1, 1, reps=5, s16=True, gemm=True)
    sys.exit(0)

if __name__ == "__main__":
    bench("gru 1x5 c512->256 (3x135x240)", 3, 135, 240, 512, 256, 1, 5)
    bench("gru 1x5 c512->256 (3x135x240) ld768", 3, 135, 240, 512, 256, 1, 5, ld=768)
    bench("same M, tiny images (760x8x16): A L2-resident", 760, 8, 16, 512, 256, 1, 5)
    bench("1x1 c2560->256 (K same, no tap reuse)", 3, 135, 240, 2560, 256, 1, 1)
    bench("1x1 c512->256", 3, 135, 240, 512, 256, 1, 1)
    bench("1x5 c512->128", 3, 135, 240, 512, 128, 1, 5)
    bench("3x3 c256->256", 3, 135, 240, 256, 256, 3, 3)
    bench("3x3 c256->256 tiny images", 760, 8, 16, 256, 256, 3, 3)
    bench("3x3 c64->64 half-res 5 frames", 5, 540, 960, 64, 64, 3, 3)
    bench("gemm 32400x32400x256", 1, 1, 32400, 256, 32400, 1, 1, reps=5)
    bench("gemm 32400x4096x256", 1, 1, 32400, 256, 4096, 1, 1, reps=5)
    bench("f32 gru 1x5 c512->256", 3, 135, 240, 512, 256, 1, 5, prec="f32")
