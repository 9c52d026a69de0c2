"""Run one conv shape repeatedly (target for rocprofv3 --pmc). GPU only."""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd")); sys.path.insert(0, ROOT)
import torch
from vfml import hip
n, h, w, cin, cout, kh, kw = 3, 135, 240, 512, 256, 1, 5
ld = int(os.environ.get('LD', 768))
x = torch.randn(n * h * w * ld, device="cuda")
wt = torch.randn(cout * kh * kw * cin, device="cuda") / math.sqrt(cin * kh * kw)
b = torch.randn(cout, device="cuda")
from vfml.weights import pack_conv_weight
mfma = int(os.environ.get("MB_MFMA", "3"))          # 1: one MFMA per product over 64-channel steps (the gate layers of the mixed plan)
wc = pack_conv_weight(wt.reshape(cout, kh, kw, cin).permute(0, 3, 1, 2), cblock=64 if mfma == 1 else True)
wobj = hip.SplitWeight(cout, wc.numel() // cout, x.device).fill(wc, scale=hip.SplitWeight.auto_scale(float(wt.abs().max())))
wobj.order = hip.KORDER_CBLOCK64 if mfma == 1 else hip.KORDER_CBLOCK
out = torch.empty(n * h * w * cout, device="cuda")
fmt = hip.FMT_S16 if os.environ.get("S16", "1") == "1" else hip.FMT_F32
if fmt == hip.FMT_S16:
    x16 = torch.empty_like(x)
    hip.to_s16(x, n * h * w, ld, ld, x16, ld)
    x = x16
for _ in range(5):
    hip.conv2d(x, cin, ld, n, h, w, wobj, b, cout, kh, kw, out, cout, pad_h=kh // 2, pad_w=kw // 2, epilogue=hip.EPI_RELU,
               in_fmt=fmt, out_fmt=fmt, mfma=mfma)
torch.cuda.synchronize()
