"""The correlation-volume GEMMs of a 1080p field at the row-major (32400 rows; 32400 / 8040 / 1980 / 480 columns) and the
4 x 8-tiled (32640; 32640 / 8640 / 2400 / 512) geometry: us per launch, one MFMA per product, f32 volumes, level 0 also with
the transposed second output.

    python tools/exp/volume_gemm_shapes.py"""
import sys, os, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd")); sys.path.insert(0, ROOT)
from vfml import hip
D = 256
def odd(n, unit=32):
    """n rounded up to a multiple of `unit` floats whose quotient is odd: the row stride is then an odd multiple of 128 bytes"""
    n = (n + unit - 1) // unit * unit
    return n if (n // unit) % 2 else n + unit


def run(M, Ns, tag, ldf):
    f1 = torch.randn(M * D, device="cuda"); rows = torch.empty_like(f1); hip.to_s16(f1, M, D, D, rows, D, scale=16.0)
    for N in Ns:
        ld = ldf(N)
        assert ld >= N
        wg = hip.SplitWeight(N, D, torch.device("cuda")).fill(torch.randn(N * D, device="cuda"), scale=16.0)
        out = torch.empty(M * ld, device="cuda")
        ld_t = ldf(M)
        assert ld_t >= M
        outs_t = [None] + ([torch.empty(N * ld_t, device="cuda")] if N == Ns[0] else [])      # transposed: N rows of ld_t >= M
        for out_t in outs_t:
            def go():
                hip.conv2d(rows, D, D, 1, 1, M, wg, None, N, 1, 1, out, ld, out_scale=1.0 / 256, in_fmt=hip.FMT_S16, mfma=1,
                           out_t=out_t, ld_out_t=ld_t if out_t is not None else 0)
            for _ in range(2): go()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): go()
            e1.record(); torch.cuda.synchronize()
            us = 100 * e0.elapsed_time(e1)
            print(f"{tag} M={M} N={N} ld={ld} {'+ transposed ld_t=%d' % ld_t if out_t is not None else '            '}: {us:8.1f} us  "
                  f"{2.0 * M * N * D / us / 1e6:6.1f} TFLOP/s", flush=True)
        del out, outs_t, wg


r32 = lambda n: (n + 31) // 32 * 32
run(32400, [32400, 8040, 1980, 480], "row-major, ld = 32 k    ", r32)
run(32640, [32640, 8640, 2400, 512], "tiled 4x8, ld = 32 k    ", r32)
run(32640, [32640, 8640, 2400, 512], "tiled 4x8, ld = 32 * odd", odd)
