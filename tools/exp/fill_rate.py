"""Dev experiment: LDS-DMA fill rate per CU by piece shape and source size (GPU only)."""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "fill_rate.so"))
L.run_fill_rate.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
def run(span, row_stride, iters=2000, grid=512):
    src = torch.zeros(span, dtype=torch.uint8, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: L.run_fill_rate(ctypes.c_void_p(src.data_ptr()), span, row_stride, iters, grid, ctypes.c_void_p(sink.data_ptr()), st)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    byts = grid * 4 * iters * 8 * 1024
    print(f"span {span>>20:5d} MiB row_stride {row_stride:5d}: {ms*1e3:8.0f} us  {byts/ms/1e9:6.2f} TB/s  {byts/ms/1e6/256:6.1f} GB/s per CU", flush=True)
for span in (2 << 20, 16 << 20, 256 << 20):
    for rs in (128, 256, 1536, 3072):
        run(span, rs)
run(2 << 20, 128, grid=256)
run(2 << 20, 3072, grid=256)
