"""Dev experiment: achieved HBM write bandwidth of tiled store patterns (GPU only)."""
import ctypes, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "store_pattern.so"))
L.run_store_pattern.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
rows, cols = 32400, 32400
def _chk(rc):
    assert rc == 0, rc
def run(ld, tr, tc, order, nt, grid=512, reps=3):
    out = torch.empty(rows * ld, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: _chk(L.run_store_pattern(ctypes.c_void_p(out.data_ptr()), rows, cols, ld, tr, tc, order, nt, grid, st))
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"ld={ld} tile {tr:4d}x{tc:5d} order={order} nt={nt} grid={grid}: {ms*1e3:8.0f} us  {rows*cols*4/ms/1e9:5.2f} TB/s", flush=True)
for ld in (32416, 32768, 32800):
    for tr, tc in ((128, 128), (128, 256), (32, 512), (16, 1024)):
        for order in (0, 1, 2):
            run(ld, tr, tc, order, 1)
run(32416, 128, 128, 0, 0)
run(32416, 128, 128, 0, 1, grid=2048)
