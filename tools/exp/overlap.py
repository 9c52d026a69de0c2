"""Dev experiment: overlap of LDS-DMA fills with MFMAs and LDS reads (GPU only)."""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "overlap.so"))
L.run_overlap.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
out = torch.zeros(1024, device="cuda")
span = 2 << 20
src = torch.zeros(span, dtype=torch.uint8, device="cuda")
names = {1: "DMA only", 2: "MFMA only", 3: "DMA + MFMA", 6: "ds_read + MFMA", 7: "DMA + ds_read + MFMA",
         9: "DMA + barrier", 11: "DMA + MFMA + barrier", 14: "ds_read + MFMA + barrier", 15: "DMA + ds_read + MFMA + barrier"}
def run(mode, iters=2000, grid=512):
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: L.run_overlap(mode, ctypes.c_void_p(src.data_ptr()), span, iters, grid, ctypes.c_void_p(out.data_ptr()), st)
    assert f() == 0; torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"{names[mode]:34s}: {ms*1e3:8.0f} us  ({ms*1e3/iters*1000:7.1f} ns per iteration; 8 KiB/wave DMA, 24 MFMA/wave)", flush=True)
for m in (1, 2, 3, 6, 7, 9, 11, 14, 15):
    run(m)
