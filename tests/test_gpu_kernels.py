"""Per-kernel parity: each C-ABI entry point (include/vfml.h) against the same op of the CPU
oracle / plain PyTorch fp32 on identical seeded inputs.  Tolerances are fp32 rounding-level."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def nhwc(x):  # NCHW cpu -> flat NHWC device
    return x.permute(0, 2, 3, 1).contiguous().cuda().reshape(-1)


def from_nhwc(flat, n, h, w, c):
    return flat.view(n, h, w, c).permute(0, 3, 1, 2).cpu()


def rel_err(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


PRECISIONS = ["f32", "f16x3"]     # exact f32 MFMA / split-f16 (3 f16 MFMAs per product)
CONV_TOL = {"f32": 2e-6, "f16x3": 5e-6}


def as_weight(flat, cout, precision, order=0):
    """flat [cout][K] f32 (host or device) -> what hip.conv2d takes for that precision; `order` says
    which K order `flat` was packed in (hip.KORDER_*)."""
    from vfml import hip
    flat = flat.cuda().contiguous()
    if precision == "f32":
        return flat
    w = hip.SplitWeight(cout, flat.numel() // cout, flat.device).fill(
        flat, scale=hip.SplitWeight.auto_scale(float(flat.abs().max())))
    w.order = order
    return w


@pytest.mark.parametrize("cin,cout,kh,kw,stride,ph,pw,H,W,n", [
    (4, 64, 7, 7, 2, 3, 3, 64, 80, 2),      # encoder stem (cin 3 padded to 4)
    (64, 96, 3, 3, 2, 1, 1, 32, 40, 2),     # strided residual conv, cout not a tile multiple
    (96, 96, 3, 3, 1, 1, 1, 17, 23, 1),     # ragged spatial size
    (128, 256, 1, 1, 1, 0, 0, 16, 20, 3),   # 1x1
    (64, 96, 1, 1, 2, 0, 0, 32, 40, 2),     # strided 1x1 shortcut
    (512, 256, 1, 5, 1, 0, 2, 16, 20, 2),   # GRU horizontal
    (512, 128, 5, 1, 1, 2, 0, 16, 20, 2),   # GRU vertical
    (256, 4, 3, 3, 1, 1, 1, 16, 20, 3),     # flow head, 4 outputs
    (256, 124, 3, 3, 1, 1, 1, 16, 20, 1),   # motion encoder tail
    (648, 256, 1, 1, 1, 0, 0, 16, 20, 1),   # corr reduce (K not a multiple of 32)
])
@pytest.mark.parametrize("precision", PRECISIONS)
def test_conv2d_matches_torch(gpu, precision, cin, cout, kh, kw, stride, ph, pw, H, W, n):
    from vfml import hip
    from vfml.weights import pack_conv_weight
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin, H, W, generator=g)
    wt = torch.randn(cout, cin, kh, kw, generator=g) / math.sqrt(cin * kh * kw)
    b = torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(x, wt, b, stride=stride, padding=(ph, pw)))
    ho, wo = ref.shape[-2:]
    out = torch.full((n * ho * wo * cout,), float("nan"), device=gpu)
    hip.conv2d(nhwc(x), cin, cin, n, H, W, as_weight(pack_conv_weight(wt), cout, precision), b.cuda(), cout, kh, kw,
               out, cout, stride=stride, pad_h=ph, pad_w=pw, epilogue=hip.EPI_RELU)
    got = from_nhwc(out, n, ho, wo, cout)
    assert torch.isfinite(got).all()
    ref64 = F.relu(F.conv2d(x.double(), wt.double(), b.double(), stride=stride, padding=(ph, pw))).float()
    assert rel_err(got, ref64) < CONV_TOL[precision]
    assert rel_err(got, ref) < 2 * CONV_TOL[precision]


@pytest.mark.parametrize("precision", PRECISIONS)
def test_conv2d_two_sources_slices_and_gru_epilogues(gpu, precision):
    """cat([r*h, x]) input from two buffers, output into a channel slice, GRU gate epilogues."""
    from vfml import hip
    from vfml.weights import pack_conv_weight
    g = torch.Generator().manual_seed(2)
    n, H, W = 2, 12, 20
    h = torch.tanh(torch.randn(n, 128, H, W, generator=g))
    x = torch.randn(n, 384, H, W, generator=g)
    wz, wr, wq = [torch.randn(128, 512, 1, 5, generator=g) / math.sqrt(512 * 5) for _ in range(3)]
    bz, br, bq = [torch.randn(128, generator=g) * 0.1 for _ in range(3)]
    hx = torch.cat([h, x], 1)
    z = torch.sigmoid(F.conv2d(hx, wz, bz, padding=(0, 2)))
    r = torch.sigmoid(F.conv2d(hx, wr, br, padding=(0, 2)))
    q = torch.tanh(F.conv2d(torch.cat([r * h, x], 1), wq, bq, padding=(0, 2)))
    href = (1 - z) * h + z * q
    # one state buffer per cell: [ z | r*h | h | x ]  (the layout the engine uses; the two sources of
    # the q convolution are channel slices of the same allocation)
    LD, Z, RH, HH, X = 768, 0, 128, 256, 384
    G = torch.zeros(n, H, W, LD)
    G[..., HH:] = hx.permute(0, 2, 3, 1)
    G = G.cuda().reshape(-1)
    wzr = as_weight(torch.cat([pack_conv_weight(wz), pack_conv_weight(wr)]), 256, precision)
    hip.conv2d(G, 512, LD, n, H, W, wzr, torch.cat([bz, br]).cuda(), 256, 1, 5, G, LD, in0_off=HH, out_off=Z,
               pad_w=2, epilogue=hip.EPI_GRU_ZR, split=128, aux0=G, ld_aux0=LD, aux0_off=HH)
    got = G.view(n, H, W, LD).permute(0, 3, 1, 2).cpu()
    assert rel_err(got[:, Z:Z + 128], z) < 2e-6
    assert rel_err(got[:, RH:RH + 128], r * h) < 2e-6
    hip.conv2d(G, 128, LD, n, H, W, as_weight(pack_conv_weight(wq), 128, precision), bq.cuda(), 128, 1, 5, G, LD,
               in0_off=RH, out_off=HH, in1=G, c1=384, ld1=LD, in1_off=X, pad_w=2, epilogue=hip.EPI_GRU_Q,
               aux0=G, ld_aux0=LD, aux0_off=Z, aux1=G, ld_aux1=LD, aux1_off=HH)
    got = G.view(n, H, W, LD).permute(0, 3, 1, 2).cpu()
    assert rel_err(got[:, HH:HH + 128], href) < 4e-6
    assert torch.equal(got[:, X:], x)  # the x slice is untouched


@pytest.mark.parametrize("precision", PRECISIONS)
def test_conv2d_as_gemm_correlation(gpu, precision):
    """out[q][s] = <f1[q], f2[s]> / sqrt(D) with a padded leading dimension (K3)."""
    from vfml import hip
    g = torch.Generator().manual_seed(3)
    P, S, D = 300, 77, 256
    f1, f2 = torch.randn(P, D, generator=g), torch.randn(S, D, generator=g)
    ld = 96
    out = torch.zeros(P * ld, device=gpu)
    # rows 10.. of a larger target matrix: exercises the row offset the pyramid build uses
    f2_all = torch.cat([torch.randn(10, D, generator=g), f2])
    hip.conv2d(f1.cuda().reshape(-1), D, D, 1, 1, P, as_weight(f2_all.reshape(-1), S + 10, precision), None, S, 1, 1,
               out, ld, out_scale=1.0 / 16.0, weight_off=10 if precision == "f16x3" else 10 * D)
    ref = (f1.double() @ f2.double().t() / 16.0).float()
    got = out.view(P, ld).cpu()
    assert rel_err(got[:, :S], ref) < CONV_TOL[precision]
    assert (got[:, S:] == 0).all()


@pytest.mark.parametrize("bias,epi", [(False, "none"), (True, "relu")])
def test_wide_gemm_persistent_path(gpu, bias, epi):
    """Split-row rows x wide plain-f32 output: the persistent form of the LDS-DMA kernel (more tiles than
    resident workgroups, ragged last row and column tiles, 16-byte row stores, optional bias)."""
    from vfml import hip
    g = torch.Generator().manual_seed(21)
    P, S, D = 1300, 6404, 256
    f1, f2 = torch.randn(P, D, generator=g), torch.randn(S, D, generator=g)
    b = torch.randn(S, generator=g) if bias else None
    ld = (S + 31) // 32 * 32
    x16 = torch.empty(P * D, device=gpu)
    hip.to_s16(f1.cuda().reshape(-1), P, D, D, x16, D)
    out = torch.full((P * ld,), 7.0, device=gpu)
    hip.conv2d(x16, D, D, 1, 1, P, as_weight(f2.reshape(-1), S, "f16x3"), b.cuda() if bias else None, S, 1, 1, out, ld,
               out_scale=1.0 / 16.0, epilogue=hip.EPI_RELU if epi == "relu" else hip.EPI_NONE, in_fmt=hip.FMT_S16)
    ref = f1.double() @ f2.double().t()
    if bias:
        ref = ref + b.double()
    ref = ref / 16.0
    if epi == "relu":
        ref = ref.clamp_min(0)
    got = out.view(P, ld).cpu()
    assert rel_err(got[:, :S], ref.float()) < CONV_TOL["f16x3"]
    assert (got[:, S:] == 7.0).all()          # nothing written past cout


def test_wide_gemm_with_transposed_second_output(gpu):
    """out[q][s] = <f1[q], f2[s]>/16 and, from the same pass, out_t[s][q] (the correlation volume of the frame
    pair read the other way round): ragged edge tiles in both directions, nothing written outside.  With each
    operand's split-row and split-plane forms cut from the same scaled copy (as the engine does), the reverse
    problem computed directly with VFML_CONV_SWAP_CROSS is bit-identical to the transposed output."""
    from vfml import hip
    g = torch.Generator().manual_seed(23)
    P, S, D = 1300, 1412, 256
    f1, f2 = torch.randn(P, D, generator=g), torch.randn(S, D, generator=g)
    ld, ldt = (S + 31) // 32 * 32, (P + 31) // 32 * 32
    SC = 16.0                                                       # both operands carry x16, undone by out_scale

    def rows(f):
        t = torch.empty(f.numel(), device=gpu)
        hip.to_s16((f * SC).cuda().reshape(-1), f.shape[0], D, D, t, D)
        return t

    def planes(f):
        return hip.SplitWeight(f.shape[0], D, torch.device("cuda")).fill(f.cuda().reshape(-1).contiguous(), scale=SC)

    out = torch.full((P * ld,), 7.0, device=gpu)
    out_t = torch.full((S * ldt,), 9.0, device=gpu)
    hip.conv2d(rows(f1), D, D, 1, 1, P, planes(f2), None, S, 1, 1, out, ld, out_scale=1.0 / 16.0 / SC,
               in_fmt=hip.FMT_S16, out_t=out_t, ld_out_t=ldt)
    ref = (f1.double() @ f2.double().t() / 16.0).float()
    got, got_t = out.view(P, ld).cpu(), out_t.view(S, ldt).cpu()
    assert rel_err(got[:, :S], ref) < CONV_TOL["f16x3"]
    assert torch.equal(got_t[:, :P], got[:, :S].t())            # the same accumulators, stored twice
    assert (got[:, S:] == 7.0).all() and (got_t[:, P:] == 9.0).all()
    rev = torch.zeros(S * ldt, device=gpu)
    hip.conv2d(rows(f2), D, D, 1, 1, S, planes(f1), None, P, 1, 1, rev, ldt, out_scale=1.0 / 16.0 / SC,
               in_fmt=hip.FMT_S16, swap_cross=True)
    assert torch.equal(rev.view(S, ldt).cpu()[:, :P], got_t[:, :P])
    with pytest.raises(RuntimeError, match="out_t"):              # only the GEMM form has it
        hip.conv2d(rows(f1), D, D, 1, 1, P, planes(f2[:64]), None, 64, 1, 1, out, ld,
                   in_fmt=hip.FMT_S16, out_t=out_t, ld_out_t=ldt)


@pytest.mark.parametrize("P,S", [(128, 1024), (516, 1028), (3000, 1500), (308, 5000), (4100, 1024), (8, 1152)])
def test_gemm_form_sizes(gpu, P, S):
    """The persistent GEMM form over tile counts below, at and above the resident slots' granularity (1, 5, 8k+r,
    ... tiles), with and without the transposed output."""
    from vfml import hip
    g = torch.Generator().manual_seed(24)
    D = 64
    f1, f2 = torch.randn(P, D, generator=g), torch.randn(S, D, generator=g)
    ref = (f1.double() @ f2.double().t()).float()
    x16 = torch.empty(P * D, device=gpu)
    hip.to_s16(f1.cuda().reshape(-1), P, D, D, x16, D)
    w = as_weight(f2.reshape(-1), S, "f16x3")
    ld, ldt = (S + 31) // 32 * 32, (P + 31) // 32 * 32
    for dual in (False, True):
        out = torch.full((P * ld,), 3.0, device=gpu)
        out_t = torch.full((S * ldt,), 4.0, device=gpu) if dual else None
        hip.conv2d(x16, D, D, 1, 1, P, w, None, S, 1, 1, out, ld, in_fmt=hip.FMT_S16, out_t=out_t, ld_out_t=ldt if dual else 0)
        got = out.view(P, ld).cpu()
        assert rel_err(got[:, :S], ref) < CONV_TOL["f16x3"], (P, S, dual)
        assert (got[:, S:] == 3.0).all()
        if dual:
            gt = out_t.view(S, ldt).cpu()
            assert torch.equal(gt[:, :P], got[:, :S].t()) and (gt[:, P:] == 4.0).all()


def test_gemm_rows_of_a_source_larger_than_one_descriptor(gpu):
    """1x1 over one split-row source whose rows span 2.2 GB (> the 2 GiB a buffer descriptor covers): the
    LDS-DMA kernel rebases its descriptor per tile.  (MemFlow's attention read-out: 4.2 GB of attention rows.)"""
    from vfml import hip
    g = torch.Generator().manual_seed(22)
    rows, ld, c, cout = 17000, 32768, 64, 128
    x = torch.randn(rows, c, generator=g)
    wt = torch.randn(cout, c, generator=g) / 8.0
    src = torch.zeros(rows * ld, device=gpu)                       # 2.2 GB, channels 0..63 of every row used
    hip.to_s16(x.cuda().reshape(-1), rows, c, c, src, ld)
    out = torch.zeros(rows * cout, device=gpu)
    hip.conv2d(src, c, ld, rows, 1, 1, as_weight(wt.reshape(-1), cout, "f16x3"), None, cout, 1, 1, out, cout,
               in_fmt=hip.FMT_S16)
    ref = (x.double() @ wt.double().t()).float()
    got = out.view(rows, cout).cpu()
    assert rel_err(got, ref) < CONV_TOL["f16x3"]
    assert rel_err(got[-200:], ref[-200:]) < CONV_TOL["f16x3"]      # the rows beyond 2 GiB


def s16_decode(flat, rows, ld, c):
    """split rows (FMT_S16) device buffer -> f32 [rows, c] on the host."""
    u = flat.view(torch.float16).view(rows, ld // 8, 2, 8).float().cpu()
    return (u[:, :, 0] + u[:, :, 1]).reshape(rows, ld)[:, :c]


def test_to_s16_roundtrip(gpu):
    from vfml import hip
    g = torch.Generator().manual_seed(12)
    x = torch.randn(50, 20, generator=g) * 3
    dst = torch.zeros(50 * 24, device=gpu)
    hip.to_s16(x.cuda().reshape(-1), 50, 20, 20, dst, 24)
    rec = s16_decode(dst, 50, 24, 24)
    assert (rec[:, 20:] == 0).all()
    assert ((rec[:, :20] - x).abs() <= 2.0 ** -21 * x.abs() + 2.0 ** -24).all()


@pytest.mark.parametrize("cblock", [False, True])
@pytest.mark.parametrize("cin,cout,kh,kw,stride", [(64, 128, 3, 3, 1), (256, 124, 3, 3, 1), (128, 64, 1, 1, 1),
                                                   (512, 256, 1, 5, 1), (72, 192, 3, 3, 1), (256, 4, 3, 3, 1),
                                                   (40, 36, 5, 1, 1), (656, 256, 1, 1, 1),
                                                   (64, 96, 3, 3, 2),       # strided, uniform-step loader
                                                   (32, 128, 5, 5, 1),      # 25 taps: tap masks beyond 9 bits
                                                   (96, 128, 1, 1, 2)])     # strided 1x1 is not the GEMM-rows case
def test_conv2d_split_rows_in_and_out(gpu, cin, cout, kh, kw, stride, cblock):
    """Split-row (S16) activations in, split-row activations out == the f32-in/f32-out result, with
    the weights in tap order and in channel-block order (incl. channel counts that are not multiples
    of 32 and every tile width of the LDS-DMA kernel)."""
    from vfml import hip
    from vfml.weights import pack_conv_weight
    g = torch.Generator().manual_seed(13)
    n, H, W = 2, 13, 21
    x = torch.randn(n, cin, H, W, generator=g)
    wt = torch.randn(cout, cin, kh, kw, generator=g) / math.sqrt(cin * kh * kw)
    b = torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), stride=stride, padding=(kh // 2, kw // 2))).float()
    ho, wo = ref.shape[-2:]
    w = as_weight(pack_conv_weight(wt, cblock=cblock), cout, "f16x3", order=int(cblock))
    x16 = torch.empty(n * H * W * cin, device=gpu)
    hip.to_s16(nhwc(x), n * H * W, cin, cin, x16, cin)
    ldo = (cout + 7) // 8 * 8 + 8
    out = torch.zeros(n * ho * wo * ldo, device=gpu)
    hip.conv2d(x16, cin, cin, n, H, W, w, b.cuda(), cout, kh, kw, out, ldo, stride=stride, pad_h=kh // 2, pad_w=kw // 2,
               epilogue=hip.EPI_RELU, in_fmt=hip.FMT_S16, out_fmt=hip.FMT_S16)
    got = s16_decode(out, n * ho * wo, ldo, ldo)
    assert (got[:, (cout + 3) // 4 * 4:] == 0).all()            # nothing written past cout
    got = got[:, :cout].view(n, ho, wo, cout).permute(0, 3, 1, 2)
    assert rel_err(got, ref) < CONV_TOL["f16x3"]


def test_split_row_conv_random_shapes(gpu):
    """40 seeded random split-row convolutions (channels, taps, stride, padding, image size, K order, epilogue,
    ragged cout and pixel counts) against float64 PyTorch: sweeps the LDS-DMA kernel's loaders, tile widths
    and edge handling beyond the shapes the networks use."""
    import random
    from vfml import hip
    from vfml.weights import pack_conv_weight
    rnd = random.Random(20250829)
    g = torch.Generator().manual_seed(16)
    for case in range(40):
        cin = rnd.choice([32, 40, 64, 72, 96, 128, 160])
        cout = rnd.choice([4, 20, 33, 64, 100, 128, 136, 192, 260])
        kh, kw = rnd.choice([(1, 1), (3, 3), (1, 5), (5, 1), (3, 1), (5, 5), (1, 3)])
        stride = rnd.choice([1, 1, 1, 2])
        ph, pw = rnd.choice([0, kh // 2]), rnd.choice([0, kw // 2])
        n, H, W = rnd.choice([1, 2, 3]), rnd.randint(kh + 2, 19), rnd.randint(kw + 2, 23)
        cblock = rnd.random() < 0.6
        relu = rnd.random() < 0.5
        x = torch.randn(n, cin, H, W, generator=g)
        wt = torch.randn(cout, cin, kh, kw, generator=g) / math.sqrt(cin * kh * kw)
        b = torch.randn(cout, generator=g)
        ref = F.conv2d(x.double(), wt.double(), b.double(), stride=stride, padding=(ph, pw))
        ref = (F.relu(ref) if relu else ref).float()
        ho, wo = ref.shape[-2:]
        ld = cin + rnd.choice([0, 8, 24])
        x16 = torch.zeros(n * H * W * ld, device=gpu)
        hip.to_s16(nhwc(x), n * H * W, cin, cin, x16, ld)
        w = as_weight(pack_conv_weight(wt, cblock=cblock), cout, "f16x3", order=int(cblock))
        ldo = cout + rnd.choice([0, 3, 8])
        out = torch.full((n * ho * wo * ldo,), 5.0, device=gpu)
        hip.conv2d(x16, cin, ld, n, H, W, w, b.cuda(), cout, kh, kw, out, ldo, stride=stride, pad_h=ph, pad_w=pw,
                   epilogue=hip.EPI_RELU if relu else hip.EPI_NONE, in_fmt=hip.FMT_S16)
        got = out.view(n, ho, wo, ldo).cpu()
        tag = f"case {case}: cin {cin} cout {cout} {kh}x{kw} s{stride} p{ph},{pw} n{n} {H}x{W} ld{ld} ldo{ldo} cblock {cblock}"
        assert (got[..., cout:] == 5.0).all(), tag
        assert rel_err(got[..., :cout].permute(0, 3, 1, 2), ref) < CONV_TOL["f16x3"], tag


@pytest.mark.parametrize("same_ld", [True, False])
@pytest.mark.parametrize("cblock", [False, True])
def test_two_split_row_sources_odd_step_count(gpu, cblock, same_ld):
    """cat([a, b]) from two split-row sources, 3x1 taps over 32 + 64 channels = 9 K steps (an odd count: the
    kernel's rounding-up step must add nothing), sources as slices of one buffer (one row stride: the
    uniform-step loader) or with different row strides (the general loader)."""
    from vfml import hip
    from vfml.weights import pack_conv_weight
    g = torch.Generator().manual_seed(15)
    n, H, W, ca, cb, cout = 2, 9, 14, 32, 64, 136
    a, b = torch.randn(n, ca, H, W, generator=g), torch.randn(n, cb, H, W, generator=g)
    wt = torch.randn(cout, ca + cb, 3, 1, generator=g) / math.sqrt((ca + cb) * 3)
    bias = torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(torch.cat([a, b], 1).double(), wt.double(), bias.double(), padding=(1, 0))).float()
    P = n * H * W
    lda, ldb = (128, 128) if same_ld else (32, 72)
    if same_ld:
        buf = torch.zeros(P * 128, device=gpu)
        hip.to_s16(nhwc(a), P, ca, ca, buf, 128, dst_off=8)
        hip.to_s16(nhwc(b), P, cb, cb, buf, 128, dst_off=48)
        srca, offa, srcb, offb = buf, 8, buf, 48
    else:
        buf = torch.zeros(P * (32 + 72), device=gpu)
        hip.to_s16(nhwc(a), P, ca, ca, buf, 32)
        hip.to_s16(nhwc(b), P, cb, cb, buf, 72, dst_off=P * 32)
        srca, offa, srcb, offb = buf, 0, buf, P * 32
    w = as_weight(pack_conv_weight(wt, cblock=cblock), cout, "f16x3", order=int(cblock))
    out = torch.zeros(P * cout, device=gpu)
    hip.conv2d(srca, ca, lda, n, H, W, w, bias.cuda(), cout, 3, 1, out, cout, in0_off=offa, in1=srcb, c1=cb, ld1=ldb,
               in1_off=offb, pad_h=1, epilogue=hip.EPI_RELU, in_fmt=hip.FMT_S16)
    got = out.view(n, H, W, cout).permute(0, 3, 1, 2).cpu()
    assert rel_err(got, ref) < CONV_TOL["f16x3"]


@pytest.mark.parametrize("cblock", [False, True])
def test_gru_epilogues_in_split_rows(gpu, cblock):
    """The engine's state buffer in split rows: gates read h / z as S16 aux operands, write S16."""
    from vfml import hip
    from vfml.weights import pack_conv_weight
    g = torch.Generator().manual_seed(14)
    n, H, W = 1, 10, 16
    h = torch.tanh(torch.randn(n, 128, H, W, generator=g))
    x = torch.randn(n, 384, H, W, generator=g)
    wz, wr, wq = [torch.randn(128, 512, 5, 1, generator=g) / math.sqrt(512 * 5) for _ in range(3)]
    bz, br, bq = [torch.randn(128, generator=g) * 0.1 for _ in range(3)]
    hx = torch.cat([h, x], 1)
    z = torch.sigmoid(F.conv2d(hx, wz, bz, padding=(2, 0)))
    r = torch.sigmoid(F.conv2d(hx, wr, br, padding=(2, 0)))
    q = torch.tanh(F.conv2d(torch.cat([r * h, x], 1), wq, bq, padding=(2, 0)))
    href = (1 - z) * h + z * q
    LD, Z, RH, HH, X = 768, 0, 128, 256, 384
    P = n * H * W
    G = torch.zeros(P * LD, device=gpu)
    hip.to_s16(nhwc(hx), P, 512, 512, G, LD, dst_off=HH)
    S = hip.FMT_S16
    wzr = as_weight(torch.cat([pack_conv_weight(wz, cblock=cblock), pack_conv_weight(wr, cblock=cblock)]), 256, "f16x3",
                    order=int(cblock))
    hip.conv2d(G, 512, LD, n, H, W, wzr, torch.cat([bz, br]).cuda(), 256, 5, 1, G, LD, in0_off=HH, out_off=Z, pad_h=2,
               epilogue=hip.EPI_GRU_ZR, split=128, aux0=G, ld_aux0=LD, aux0_off=HH, in_fmt=S, out_fmt=S, aux_fmt=S)
    hip.conv2d(G, 128, LD, n, H, W, as_weight(pack_conv_weight(wq, cblock=cblock), 128, "f16x3", order=int(cblock)),
               bq.cuda(), 128, 5, 1, G, LD,
               in0_off=RH, out_off=HH, in1=G, c1=384, ld1=LD, in1_off=X, pad_h=2, epilogue=hip.EPI_GRU_Q,
               aux0=G, ld_aux0=LD, aux0_off=Z, aux1=G, ld_aux1=LD, aux1_off=HH, in_fmt=S, out_fmt=S, aux_fmt=S)
    got = s16_decode(G, P, LD, LD).view(n, H, W, LD).permute(0, 3, 1, 2)
    assert rel_err(got[:, Z:Z + 128], z) < 3e-6
    assert rel_err(got[:, RH:RH + 128], r * h) < 3e-6
    assert rel_err(got[:, HH:HH + 128], href) < 5e-6
    assert rel_err(got[:, X:], x) < 1e-6


def test_lookup_and_flow_in_split_rows(gpu):
    from oracle import mof_oracle as mo
    from vfml import hip
    h, w, radius, levels = 16, 20, 4, 4
    pyr = _pyramid_inputs(21, h, w, levels)
    coords = mo.coords_grid(1, h, w) + torch.randn(1, 2, h, w, generator=torch.Generator().manual_seed(22)) * 3
    blk = mo.CorrBlock.__new__(mo.CorrBlock)
    blk.num_levels, blk.radius, blk.pyramid = levels, radius, pyr
    ref = blk(coords)[0].permute(1, 2, 0).reshape(h * w, -1)
    P = h * w
    hl = [p.shape[-2] for p in pyr]
    wl = [p.shape[-1] for p in pyr]
    ld = [a * b for a, b in zip(hl, wl)]
    dev = [p.reshape(P, -1).cuda().reshape(-1) for p in pyr]
    c4 = torch.zeros(P, 4)
    c4[:, :2] = coords[0].permute(1, 2, 0).reshape(P, 2)
    out = torch.full((P * 664,), 7.0, device=gpu)          # [.. 328 | 328 | 8 spare]
    hip.corr_lookup(dev, hl, wl, ld, radius, P, c4.cuda().reshape(-1), 0, 4, out, 328, 664, out_fmt=hip.FMT_S16)
    got = s16_decode(out, P, 664, 664)
    assert (got[:, 328:652] - ref).abs().max().item() < 2e-5
    assert (got[:, 652:656] == 0).all()                     # zero-filled up to the whole unit
    # flow into the second quad of a unit, first quad untouched
    coords1 = torch.empty(P * 4, device=gpu)
    hip.coords_init(coords1, 1, h, w)
    delta = torch.randn(P, 4, generator=torch.Generator().manual_seed(23))
    buf = torch.zeros(P * 16, device=gpu)
    hip.to_s16(torch.ones(P * 4, device=gpu), P, 4, 4, buf, 16, dst_off=8)
    hip.coords_update(coords1, delta.cuda().reshape(-1), 1, h, w, flow_b=buf, ld_b=16, flow_b_off=12,
                      fmt_b=hip.FMT_S16)
    dec = s16_decode(buf, P, 16, 16)
    assert (dec[:, 8:12] == 1).all() and (dec[:, :8] == 0).all()
    grid = mo.coords_grid(1, h, w)[0].permute(1, 2, 0).reshape(P, 2)
    want = (torch.cat([grid, grid], 1) + delta) - torch.cat([grid, grid], 1)
    assert ((dec[:, 12:16] - want).abs() <= 2.0 ** -21 * want.abs() + 2.0 ** -24).all()


def test_conv2d_rejects_bad_arguments(gpu):
    from vfml import hip
    x = torch.zeros(64, device=gpu)
    for w in (x, hip.SplitWeight(8, 4, gpu).fill(x)):
        with pytest.raises(RuntimeError, match="multiples of 4"):
            hip.conv2d(x, 3, 3, 1, 4, 4, w, None, 4, 1, 1, x, 4)
        with pytest.raises(RuntimeError, match="ldo"):
            hip.conv2d(x, 4, 4, 1, 4, 4, w, None, 8, 1, 1, x, 4)
    with pytest.raises(RuntimeError, match="rounded up to 32"):
        hip.conv2d(x, 4, 4, 1, 4, 4, hip.SplitWeight(8, 4, gpu).fill(x), None, 8, 3, 3, x, 8)


def test_split_f16_reconstructs_to_22_bits(gpu):
    """(hi + lo)/scale reproduces x to 2^-21 relative (plus an absolute floor of one f16 subnormal
    step, where lo itself is subnormal), across 10 orders of magnitude; K padded to 8 with zeros."""
    from vfml import hip
    g = torch.Generator().manual_seed(11)
    x = torch.randn(37, 13, generator=g) * torch.logspace(-6, 4, 13)
    sw = hip.SplitWeight(37, 13, gpu).fill(x.cuda().reshape(-1), scale=0.25)
    assert sw.kp == 32
    hi = sw.hi.view(37, 32).cpu().double()
    lo = sw.lo.view(37, 32).cpu().double()
    rec = (hi + lo) / 0.25
    assert (rec[:, 13:] == 0).all()
    err = (rec[:, :13] - x.double()).abs()
    assert (err <= 2.0 ** -21 * x.double().abs() + 2.0 ** -22).all()


@pytest.mark.parametrize("kind", ["u8", "f32"])
def test_frames_to_nhwc4(gpu, kind):
    from vfml import hip
    g = torch.Generator().manual_seed(4)
    n, H, W = 3, 16, 24
    u8 = torch.randint(0, 256, (n, H, W, 3), generator=g, dtype=torch.uint8)
    x01 = (u8.float() / 255.0).permute(0, 3, 1, 2).contiguous()     # what the reference uploads
    ref = 2.0 * x01 - 1.0
    dst = torch.empty(n * H * W * 4, device=gpu)
    hip.frames_to_nhwc4(u8.cuda() if kind == "u8" else x01.cuda(), n, H, W, 2.0, -1.0, dst)
    got = dst.view(n, H, W, 4).cpu()
    assert torch.equal(got[..., :3].permute(0, 3, 1, 2), ref)
    assert (got[..., 3] == 0).all()


@pytest.mark.parametrize("c,hw", [(64, 5000), (96, 333), (128, 4097)])
def test_instnorm(gpu, c, hw):
    from vfml import hip
    g = torch.Generator().manual_seed(5)
    n = 2
    x = torch.randn(n, c, hw, 1, generator=g) * 3 + 1.5
    res = torch.randn(n, c, hw, 1, generator=g)
    xd, rd = nhwc(x), nhwc(res)
    st = torch.empty(n * c * 2, device=gpu)
    st2 = torch.empty(n * c * 2, device=gpu)
    ws = torch.empty(hip.instnorm_workspace_bytes(n, hw, c) // 8 + 1, device=gpu, dtype=torch.float64)
    hip.instnorm_stats(xd, n, hw, c, st, ws)
    hip.instnorm_stats(rd, n, hw, c, st2, ws)
    out = torch.empty_like(xd)
    y = F.relu(F.instance_norm(x))
    hip.instnorm_apply(xd, st, n, hw, c, out)
    assert rel_err(from_nhwc(out, n, hw, 1, c), y) < 3e-6
    hip.instnorm_apply(xd, st, n, hw, c, out, res=rd)
    assert rel_err(from_nhwc(out, n, hw, 1, c), F.relu(res + y)) < 3e-6
    hip.instnorm_apply(xd, st, n, hw, c, out, res=rd, res_stats=st2)
    assert rel_err(from_nhwc(out, n, hw, 1, c), F.relu(F.instance_norm(res) + y)) < 3e-6
    # split-row results (and split-row residual): what the encoder's LDS-DMA convolutions read
    o16 = torch.zeros_like(xd)
    dec = lambda t: s16_decode(t, n * hw, c, c).view(n, hw, 1, c).permute(0, 3, 1, 2)
    hip.instnorm_apply(xd, st, n, hw, c, o16, out_fmt=hip.FMT_S16)
    assert rel_err(dec(o16), y) < 3e-6
    r16 = torch.empty_like(xd)
    hip.to_s16(rd, n * hw, c, c, r16, c)
    hip.instnorm_apply(xd, st, n, hw, c, o16, res=r16, out_fmt=hip.FMT_S16)
    assert rel_err(dec(o16), F.relu(res + y)) < 3e-6
    hip.instnorm_apply(xd, st, n, hw, c, o16, res=rd, res_stats=st2, out_fmt=hip.FMT_S16)
    assert rel_err(dec(o16), F.relu(F.instance_norm(res) + y)) < 3e-6


@pytest.mark.parametrize("src16", [False, True])
@pytest.mark.parametrize("n,cin,cout,k,stride,H,W", [(1, 64, 64, 3, 1, 37, 53), (1, 4, 64, 7, 2, 70, 90), (1, 64, 96, 3, 2, 61, 45),
                                                     (2, 96, 96, 3, 1, 16, 24), (1, 64, 96, 1, 2, 50, 38), (3, 128, 128, 3, 1, 8, 16),
                                                     (1, 128, 256, 1, 1, 33, 41), (1, 96, 128, 3, 2, 270, 480)])
def test_conv_leaves_instnorm_partials(gpu, n, cin, cout, k, stride, H, W, src16):
    """conv2d(stats_part=...) + instnorm_finalize == conv2d + instnorm_stats on the stored result (the encoder's
    fused path): ragged last row tile, column tiles that end inside a tile, several images with whole tiles each;
    f32 sources (the register-staged kernel, blocks of 128 pixels) and split-row sources (the LDS-DMA kernel: blocks
    of 32 pixels, every tile shape the dispatcher picks for these sizes)."""
    from vfml import hip
    g = torch.Generator().manual_seed(n * 1000 + cout + k)
    x = torch.randn(n, cin, H, W, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g)
    pad = k // 2
    ho, wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    hw = ho * wo
    if src16 and cin < 32:
        pytest.skip("split-row sources have at least 32 channels")
    from vfml.weights import pack_conv_weight
    wobj = as_weight(pack_conv_weight(wt, cblock=src16), cout, "f16x3", order=int(src16))
    out_a = torch.empty(n * hw * cout, device=gpu)
    out_b = torch.empty_like(out_a)
    rows = hip.STATS_ROWS_S16 if src16 else hip.STATS_ROWS_F32
    chunks = (hw + rows - 1) // rows
    part = torch.full((n * chunks * cout * 2 + 8,), float("nan"), device=gpu, dtype=torch.float64)
    xd = nhwc(x)
    fmt = hip.FMT_F32
    if src16:
        x16 = torch.empty_like(xd)
        hip.to_s16(xd, n * H * W, cin, cin, x16, cin)
        xd, fmt = x16, hip.FMT_S16
    hip.conv2d(xd, cin, cin, n, H, W, wobj, b.to(gpu), cout, k, k, out_a, cout, stride=stride, pad_h=pad, pad_w=pad,
               stats_part=part, in_fmt=fmt)
    hip.conv2d(xd, cin, cin, n, H, W, wobj, b.to(gpu), cout, k, k, out_b, cout, stride=stride, pad_h=pad, pad_w=pad,
               in_fmt=fmt)
    assert torch.equal(out_a, out_b)
    assert torch.isnan(part[n * chunks * cout * 2:]).all() and not torch.isnan(part[:n * chunks * cout * 2]).any()
    st_a = torch.empty(n * cout * 2, device=gpu)
    st_b = torch.empty_like(st_a)
    hip.instnorm_finalize(part, n, chunks, cout, hw, st_a)
    ws = torch.empty(hip.instnorm_workspace_bytes(n, hw, cout) // 8 + 1, device=gpu, dtype=torch.float64)
    hip.instnorm_stats(out_b, n, hw, cout, st_b, ws)
    assert torch.allclose(st_a, st_b, rtol=2e-7, atol=1e-9), (st_a - st_b).abs().max().item()
    ref = from_nhwc(out_b, n, ho, wo, cout).double()
    assert torch.allclose(st_a.view(n, cout, 2)[..., 0].cpu().double(), ref.mean(dim=(2, 3)), rtol=1e-5, atol=1e-6)


def test_conv_stats_partials_reject_straddling_tiles(gpu):
    from vfml import hip
    x = torch.zeros(2 * 10 * 10 * 64, device=gpu)
    wobj = as_weight(torch.zeros(64 * 9 * 64, device=gpu) + 0.01, 64, "f16x3")
    out = torch.empty(2 * 100 * 64, device=gpu)
    part = torch.empty(2 * 1 * 64 * 2, device=gpu, dtype=torch.float64)
    with pytest.raises(RuntimeError, match="straddle"):
        hip.conv2d(x, 64, 64, 2, 10, 10, wobj, None, 64, 3, 3, out, 64, pad_h=1, pad_w=1, stats_part=part)
    with pytest.raises(RuntimeError, match="stats_part|vfml_conv2d"):
        hip.conv2d(x, 64, 64, 1, 10, 10, torch.zeros(64 * 9 * 64, device=gpu), None, 64, 3, 3, out, 64, pad_h=1, pad_w=1,
                   stats_part=part)


def test_avgpool2x2_floor(gpu):
    from vfml import hip
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 8, 7, 9, generator=g)
    out = torch.empty(2 * 3 * 4 * 8, device=gpu)
    hip.avgpool2x2(nhwc(x), 2, 7, 9, 8, out)
    assert torch.allclose(from_nhwc(out, 2, 3, 4, 8), F.avg_pool2d(x, 2, 2), rtol=0, atol=1e-6)


def _pyramid_inputs(seed, h, w, levels, nq_maps=1):
    g = torch.Generator().manual_seed(seed)
    P = h * w
    corr0 = torch.randn(nq_maps * P, 1, h, w, generator=g)
    pyr = [corr0]
    for _ in range(levels - 1):
        pyr.append(F.avg_pool2d(pyr[-1], 2, 2))
    return pyr


@pytest.mark.parametrize("radius,levels", [(4, 4), (3, 3)])
def test_corr_lookup_matches_grid_sample(gpu, radius, levels):
    from oracle import mof_oracle as mo
    from vfml import hip
    h, w = 18, 24
    pyr = _pyramid_inputs(7, h, w, levels)
    g = torch.Generator().manual_seed(8)
    coords = mo.coords_grid(1, h, w) + torch.randn(1, 2, h, w, generator=g) * 6.0   # some land far outside
    blk = mo.CorrBlock.__new__(mo.CorrBlock)
    blk.num_levels, blk.radius, blk.pyramid = levels, radius, pyr
    ref = blk(coords)                                        # [1, L*win, h, w]
    P = h * w
    hl = [p.shape[-2] for p in pyr]
    wl = [p.shape[-1] for p in pyr]
    ld = [(a * b + 31) // 32 * 32 for a, b in zip(hl, wl)]
    dev = []
    for p, l in zip(pyr, ld):
        t = torch.full((P, l), float("nan"))
        t[:, :p.shape[-2] * p.shape[-1]] = p.reshape(P, -1)
        dev.append(t.cuda().reshape(-1))
    c4 = torch.zeros(P, 4)
    c4[:, 2:] = coords[0].permute(1, 2, 0).reshape(P, 2)     # use the (bwd) slot at +2
    nch = levels * (2 * radius + 1) ** 2
    out = torch.full((P * (nch + 4),), float("nan"), device=gpu)
    hip.corr_lookup(dev, hl, wl, ld, radius, P, c4.cuda().reshape(-1), 2, 4, out, 4, nch + 4)
    got = out.view(P, nch + 4)[:, 4:].cpu().view(h, w, nch).permute(2, 0, 1)[None]
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < 2e-5


def test_corr_lookup_several_maps_in_separate_allocations(gpu):
    """Each query map (correlation problem) brings its own pyramid allocation; rows are map-major."""
    from vfml import hip
    h, w, r = 12, 16, 2
    P = h * w
    g = torch.Generator().manual_seed(31)
    maps = [[torch.randn(P, P, generator=g).cuda().reshape(-1), torch.randn(P, P // 4, generator=g).cuda().reshape(-1)]
            for _ in range(3)]
    coords = (torch.rand(3 * P, 4, generator=g) * torch.tensor([w, h, w, h])).cuda().reshape(-1)
    hl, wl, ld = [h, h // 2], [w, w // 2], [P, P // 4]
    together = torch.empty(3 * P * 50, device=gpu)
    hip.corr_lookup(maps, hl, wl, ld, r, P, coords, 0, 4, together, 0, 50)
    for m in range(3):
        alone = torch.empty(P * 50, device=gpu)
        hip.corr_lookup(maps[m], hl, wl, ld, r, P, coords, m * P * 4, 4, alone, 0, 50)
        assert torch.equal(together.view(3, P, 50)[m], alone.view(P, 50))


def test_corr_lookup_integer_coords_is_exact_gather(gpu):
    """Known answer: at integer coordinates the window is a plain gather, zeros outside."""
    from vfml import hip
    h, w, r = 16, 16, 4
    P = h * w
    corr = torch.arange(P * P, dtype=torch.float32).view(P, P)
    coords = torch.zeros(P, 4)
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    coords[:, 0], coords[:, 1] = xs.reshape(-1).float(), ys.reshape(-1).float()
    out = torch.empty(P * 81, device=gpu)
    hip.corr_lookup([corr.cuda().reshape(-1)], [h], [w], [P], r, P, coords.cuda().reshape(-1), 0, 4, out, 0, 81)
    got = out.view(P, 9, 9).cpu()
    q = 5 * w + 7
    for i in range(9):
        for j in range(9):
            x, y = 7 + i - r, 5 + j - r                      # channel (i,j): x + d[i], y + d[j]
            want = corr[q, y * w + x].item() if 0 <= x < w and 0 <= y < h else 0.0
            assert got[q, i, j].item() == want


@pytest.mark.parametrize("radius,levels,vol16,out16", [(4, 4, False, True), (4, 4, True, False), (3, 3, False, False), (2, 2, False, False)])
@pytest.mark.parametrize("tws,ths", [(2, 3), (3, 2), (0, 1)])
def test_corr_lookup_reads_tiled_volumes(gpu, radius, levels, vol16, out16, tws, ths):
    """vol_tile (include/vfml.h): level images stored as 2^tws x 2^ths tiles under rows in the tile order of the query grid
    give bit for bit the windows of the row-major volume - fixed-radius kernels (f32 / f16 volumes, f32 / split-row
    output) and the generic one (radius 2); ragged sizes, so edge tiles are partly empty (filled with NaN: never read)."""
    from vfml import hip
    h, w = 21, 27
    P = h * w
    g = torch.Generator().manual_seed(77)
    hl, wl = [h >> l for l in range(levels)], [w >> l for l in range(levels)]
    ld = [(a * b + 31) // 32 * 32 for a, b in zip(hl, wl)]
    vt = hip.VolTile(tws, ths)
    dt = torch.float16 if vol16 else torch.float32
    plain, tiled, ldt = [], [], []
    for l in range(levels):
        v = torch.randn(P, hl[l] * wl[l], generator=g).to(dt)
        t = torch.full((P, ld[l]), float("nan"), dtype=dt)
        t[:, :hl[l] * wl[l]] = v
        plain.append(t.cuda().reshape(-1))
        n = vt.count(hl[l], wl[l])
        ldt.append((n + 31) // 32 * 32)
        cols = torch.full((P, ldt[-1]), float("nan"), dtype=dt)
        cols[:, vt.position(hl[l], wl[l], "cpu")] = v
        rows = torch.full((vt.count(h, w), ldt[-1]), float("nan"), dtype=dt)
        rows[vt.position(h, w, "cpu")] = cols
        tiled.append(rows.cuda().reshape(-1))
    coords = (torch.rand(P, 4, generator=g) * torch.tensor([w, h, w, h]) + torch.randn(P, 4, generator=g) * 3.0).cuda().reshape(-1)
    nch = levels * (2 * radius + 1) ** 2
    ldo = (nch + 7) // 8 * 8
    fmt = hip.FMT_S16 if out16 else hip.FMT_F32
    vf = hip.FMT_F16 if vol16 else hip.FMT_F32
    a = torch.zeros(P * ldo, device=gpu)
    b = torch.zeros(P * ldo, device=gpu)
    hip.corr_lookup(plain, hl, wl, ld, radius, P, coords, 0, 4, a, 0, ldo, out_fmt=fmt, vol_fmt=vf)
    hip.corr_lookup(tiled, hl, wl, ldt, radius, P, coords, 0, 4, b, 0, ldo, out_fmt=fmt, vol_fmt=vf, vol_tile=vt.code)
    assert torch.isfinite(a.view(P, ldo)[:, :nch] if not out16 else a.view(torch.float16)).all()
    assert torch.equal(a, b)
    with pytest.raises(RuntimeError, match="whole tiles"):
        hip.corr_lookup(tiled, hl, wl, ld, radius, P, coords, 0, 4, b, 0, ldo, out_fmt=fmt, vol_fmt=vf, vol_tile=vt.code + 16 * 3)


def test_coords_and_upsample(gpu):
    from oracle import mof_oracle as mo
    from vfml import hip
    g = torch.Generator().manual_seed(9)
    n, h, w = 2, 10, 12
    coords1 = torch.empty(n * h * w * 4, device=gpu)
    hip.coords_init(coords1, n, h, w)
    delta = torch.randn(n, h, w, 4, generator=g)
    flow = torch.empty(n * h * w * 4, device=gpu)
    wide = torch.zeros(n * h * w * 8, device=gpu)
    hip.coords_update(coords1, delta.cuda().reshape(-1), n, h, w, flow_a=flow, ld_a=4, flow_b=wide, ld_b=8,
                      flow_b_off=4)
    grid = mo.coords_grid(n, h, w).permute(0, 2, 3, 1)
    c1 = torch.cat([grid + delta[..., :2], grid + delta[..., 2:]], -1)
    assert torch.equal(coords1.view(n, h, w, 4).cpu(), c1)
    want_flow = c1 - torch.cat([grid, grid], -1)
    assert torch.equal(flow.view(n, h, w, 4).cpu(), want_flow)
    assert torch.equal(wide.view(n, h, w, 8).cpu()[..., 4:], want_flow)
    mask = torch.randn(n, 1152, h, w, generator=g) * 2
    md = nhwc(mask)
    for d in range(2):
        for m in range(n):
            out = torch.empty(8 * h * 8 * w * 2, device=gpu)
            hip.convex_upsample(coords1, m * h * w * 4, 2 * d, md, m * h * w * 1152 + d * 576, 1152, h, w, out)
            ref = mo.upsample_flow(want_flow[m:m + 1, ..., 2 * d:2 * d + 2].permute(0, 3, 1, 2),
                                   mask[m:m + 1, d * 576:(d + 1) * 576])
            got = out.view(8 * h, 8 * w, 2).permute(2, 0, 1).cpu()[None]
            assert (got - ref).abs().max().item() < 2e-5


def test_upsample_uniform_mask_known_answer(gpu):
    """Uniform logits -> every sub-pixel is the mean of the 3x3 (zero-padded) neighbourhood x8."""
    from vfml import hip
    h, w = 6, 7
    coords1 = torch.empty(h * w * 4, device=gpu)
    hip.coords_init(coords1, 1, h, w)
    delta = torch.zeros(h, w, 4)
    delta[..., 0] = 1.0                                       # fwd flow = (1, 0) everywhere
    hip.coords_update(coords1, delta.cuda().reshape(-1), 1, h, w)
    mask = torch.zeros(h * w * 576, device=gpu)
    out = torch.empty(8 * h * 8 * w * 2, device=gpu)
    hip.convex_upsample(coords1, 0, 0, mask, 0, 576, h, w, out)
    got = out.view(8 * h, 8 * w, 2).cpu()
    assert torch.allclose(got[8 * 2:8 * 3, 8 * 3:8 * 4, 0], torch.full((8, 8), 8.0), atol=1e-5)   # interior
    assert torch.allclose(got[0:8, 0:8, 0], torch.full((8, 8), 8.0 * 4 / 9), atol=1e-5)         # corner: 4 of 9 taps
    assert (got[..., 1] == 0).all()


def test_flow_lod_pyramid_bit_exact_vs_reference_fixture(gpu):
    """GPU LOD generator == the reference's per-pixel loop (fixtures cut from storage/cache_manager.py)."""
    import os
    import numpy as np
    from storage import LODGenerator
    A = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_plumbing.npz"))
    for tag in "abcd":
        lods = LODGenerator.generate_lods(torch.from_numpy(A[f"lod_{tag}_in"]).cuda(), 4)
        for k, l in enumerate(lods):
            assert isinstance(l, np.ndarray) and np.array_equal(l, A[f"lod_{tag}_{k}"]), (tag, k)
    big = torch.randn(1080, 1920, 2, generator=torch.Generator().manual_seed(41))
    cpu = LODGenerator.generate_lods(big.numpy(), 5)
    dev = LODGenerator.generate_lods(big.cuda(), 5)
    assert all(np.array_equal(a, b) for a, b in zip(cpu, dev))


def _f16(t):
    return t.half().double()


@pytest.mark.parametrize("mfma", [2, "2a", 1])
@pytest.mark.parametrize("cin,cout,kh,kw,stride,src16,cblock", [
    (64, 192, 3, 3, 1, True, True),       # 128 x 192 / 192 x 128 tiles, uniform-step loader
    (96, 128, 1, 5, 1, True, False),      # tap order: the general loader
    (72, 100, 3, 3, 2, True, True),       # channel count not a multiple of 32, strided, ragged cout
    (128, 64, 3, 3, 1, True, True),       # 128 x 64 tiles
    (256, 4, 3, 3, 1, True, True),        # 128 x 32 tiles (flow head)
    (64, 96, 3, 3, 2, False, False),      # f32 sources: the register-staged kernel
    (4, 64, 7, 7, 2, False, False),       # encoder stem
])
def test_conv2d_reduced_mfma_counts_drop_exactly_the_lo_terms(gpu, mfma, cin, cout, kh, kw, stride, src16, cblock):
    """VFML_CONV_MFMA2 / _MFMA2A / _MFMA1: the result equals a float64 convolution of the operands with the weights
    (2, 1) and / or the activations ("2a", 1) rounded to ONE f16, to nearest - i.e. exactly the a*w_lo and / or a_lo*w
    terms are gone, with an unbiased rounding - and differs from the full-precision result by that rounding."""
    from vfml import hip
    from vfml.weights import pack_conv_weight
    g = torch.Generator().manual_seed(31 + {2: 2, "2a": 3, 1: 1}[mfma])
    n, H, W = 2, 17, 22
    x = torch.randn(n, cin, H, W, generator=g)
    wt = torch.randn(cout, cin, kh, kw, generator=g) / math.sqrt(cin * kh * kw)
    b = torch.randn(cout, generator=g)
    ph, pw = kh // 2, kw // 2
    w = as_weight(pack_conv_weight(wt, cblock=cblock), cout, "f16x3", order=int(cblock))
    # the kernel sees the weights times the power-of-two split scale, rounded to f16, then divides the scale out
    wq = _f16(wt * w.scale) / w.scale if mfma in (2, 1) else wt.double()
    xq = _f16(x) if mfma in ("2a", 1) else x.double()
    emu = F.conv2d(xq, wq, b.double(), stride=stride, padding=(ph, pw)).float()
    full = F.conv2d(x.double(), wt.double(), b.double(), stride=stride, padding=(ph, pw)).float()
    ho, wo = emu.shape[-2:]
    out = torch.full((n * ho * wo * cout,), float("nan"), device=gpu)
    if src16:
        src = torch.empty(n * H * W * cin, device=gpu)
        hip.to_s16(nhwc(x), n * H * W, cin, cin, src, cin)
    else:
        src = nhwc(x)
    hip.conv2d(src, cin, cin, n, H, W, w, b.cuda(), cout, kh, kw, out, cout, stride=stride, pad_h=ph, pad_w=pw,
               in_fmt=hip.FMT_S16 if src16 else hip.FMT_F32, mfma=mfma)
    got = from_nhwc(out, n, ho, wo, cout)
    assert torch.isfinite(got).all()
    assert rel_err(got, emu) < CONV_TOL["f16x3"], rel_err(got, emu)
    d = rel_err(got, full)
    assert 5e-6 < d < 3e-3, d        # the f16 rounding is there, and is no more than that


def test_gemm_form_reduced_mfma_counts(gpu):
    """The persistent GEMM form (correlation volume) with two-plane weights at 2 and 1 MFMAs per product, incl. the
    transposed second output, which at one MFMA per product is bit-identical to the direct reverse product."""
    from vfml import hip
    g = torch.Generator().manual_seed(41)
    P, S, D = 644, 1284, 256
    f1, f2 = torch.randn(P, D, generator=g), torch.randn(S, D, generator=g)
    x16 = torch.empty(P * D, device=gpu)
    hip.to_s16(f1.cuda().reshape(-1) * 16.0, P, D, D, x16, D)
    y16 = torch.empty(S * D, device=gpu)
    hip.to_s16(f2.cuda().reshape(-1) * 16.0, S, D, D, y16, D)
    w2 = hip.SplitWeight(S, D, gpu).fill(f2.cuda().reshape(-1), scale=16.0)
    w1 = hip.SplitWeight(P, D, gpu).fill(f1.cuda().reshape(-1), scale=16.0)
    ld, ldt = (S + 31) // 32 * 32, (P + 31) // 32 * 32
    for mfma in (2, 1):
        out = torch.zeros(P * ld, device=gpu)
        out_t = torch.zeros(S * ldt, device=gpu)
        hip.conv2d(x16, D, D, 1, 1, P, w2, None, S, 1, 1, out, ld, out_scale=1.0 / 16.0, in_fmt=hip.FMT_S16, mfma=mfma,
                   out_t=out_t, ld_out_t=ldt)
        a = _f16(f1 * 16.0) / 16.0 if mfma == 1 else f1.double()
        emu = (a @ (_f16(f2 * 16.0) / 16.0).t()).float()
        got = out.view(P, ld)[:, :S].cpu()
        assert rel_err(got, emu) < CONV_TOL["f16x3"]
        assert torch.equal(out_t.view(S, ldt)[:, :P].cpu(), got.t())
        if mfma == 1:
            rev = torch.zeros(S * ldt, device=gpu)
            hip.conv2d(y16, D, D, 1, 1, S, w1, None, P, 1, 1, rev, ldt, out_scale=1.0 / 16.0, in_fmt=hip.FMT_S16, mfma=1)
            assert torch.equal(rev.view(S, ldt)[:, :P].cpu(), got.t())


@pytest.mark.parametrize("c0,c1,cout,kh,kw,stride", [(128, 0, 192, 3, 3, 1),     # 128 x 192 / 192 x 128 tiles
                                                    (64, 128, 256, 1, 5, 1),    # two sources (the GRU gate shape)
                                                    (64, 0, 64, 3, 3, 2),       # 128 x 64 tiles, strided
                                                    (256, 0, 136, 1, 1, 1),     # 1x1: any weight order
                                                    (192, 0, 128, 5, 1, 1)])    # three 64-channel blocks, vertical taps
def test_conv2d_one_mfma_64_channel_steps(gpu, c0, c1, cout, kh, kw, stride):
    """VFML_CONV_MFMA1 over whole 64-channel blocks: the kernel steps 64 channels of hi halves at a time (weights in
    VFML_KORDER_CBLOCK64 order, or any order for 1x1).  Same numbers as the 32-channel-step kernel would give up to the
    order of the f32 additions: both equal the float64 convolution of the f16-rounded operands."""
    from vfml import hip
    from vfml.weights import pack_conv_weight
    g = torch.Generator().manual_seed(51 + c0 + kh)
    n, H, W = 2, 19, 26
    cin = c0 + c1
    x = torch.randn(n, cin, H, W, generator=g)
    wt = torch.randn(cout, cin, kh, kw, generator=g) / math.sqrt(cin * kh * kw)
    b = torch.randn(cout, generator=g)
    ph, pw = kh // 2, kw // 2
    order = hip.KORDER_TAP if kh * kw == 1 else hip.KORDER_CBLOCK64
    w = as_weight(pack_conv_weight(wt, cblock=64 if order else False), cout, "f16x3", order=order)
    emu = F.relu(F.conv2d(_f16(x), _f16(wt * w.scale) / w.scale, b.double(), stride=stride, padding=(ph, pw))).float()
    ho, wo = emu.shape[-2:]
    P = n * H * W
    LD = cin + 64                                     # both sources as channel slices of one wider buffer
    buf = torch.zeros(P * LD, device=gpu)
    xs = nhwc(x).view(P, cin)
    hip.to_s16(xs[:, :c0].contiguous().reshape(-1), P, c0, c0, buf, LD, dst_off=32)
    if c1:
        hip.to_s16(xs[:, c0:].contiguous().reshape(-1), P, c1, c1, buf, LD, dst_off=32 + c0)
    out = torch.full((n * ho * wo * cout,), float("nan"), device=gpu)
    hip.conv2d(buf, c0, LD, n, H, W, w, b.cuda(), cout, kh, kw, out, cout, stride=stride, pad_h=ph, pad_w=pw, in0_off=32,
               in1=buf if c1 else None, c1=c1, ld1=LD if c1 else 0, in1_off=32 + c0, epilogue=hip.EPI_RELU,
               in_fmt=hip.FMT_S16, mfma=1)
    got = from_nhwc(out, n, ho, wo, cout)
    assert torch.isfinite(got).all()
    assert rel_err(got, emu) < CONV_TOL["f16x3"], rel_err(got, emu)
    if order == hip.KORDER_CBLOCK64 and c1 == 0:
        # the 64-channel-block order belongs to one-MFMA calls more than 32 outputs wide: anything else is refused
        with pytest.raises(RuntimeError, match="CBLOCK64"):
            hip.conv2d(buf, c0, LD, n, H, W, w, b.cuda(), cout, kh, kw, out, cout, stride=stride, pad_h=ph, pad_w=pw,
                       in0_off=32, in_fmt=hip.FMT_S16, mfma=3)
        w4 = as_weight(pack_conv_weight(wt[:4], cblock=64), 4, "f16x3", order=order)
        with pytest.raises(RuntimeError, match="CBLOCK64"):
            hip.conv2d(buf, c0, LD, n, H, W, w4, None, 4, kh, kw, out, 4, stride=stride, pad_h=ph, pad_w=pw,
                       in0_off=32, in_fmt=hip.FMT_S16, mfma=1)


@pytest.mark.parametrize("mfma", [3, 1])
@pytest.mark.parametrize("c0,c1,cout,kh,kw,n,H,W", [
    (64, 128, 256, 1, 5, 3, 19, 26),     # the GRU gate shape: two sources, 1x5, two column tiles of 128
    (128, 0, 192, 3, 3, 2, 17, 23),      # 128 x 192 tiles (cout 192), 3x3: vertical taps masked at the DMA
    (64, 0, 136, 3, 3, 1, 40, 7),        # image rows far narrower than a tile: every fragment crosses several rows
    (64, 64, 128, 2, 4, 2, 9, 14),       # even taps (pad 1 / 2 on an even filter is not "same": refused -> per-tap path)
    (64, 0, 256, 1, 3, 1, 3, 700),       # rows longer than a tile
    (64, 0, 128, 3, 5, 4, 6, 5),         # four tiny images in one tile
    (64, 0, 64, 3, 3, 2, 21, 30),        # 256 x 64 tiles (the encoders' 64-channel layers)
    (128, 0, 96, 3, 3, 1, 33, 17),       # 256 x 96 tiles
    (64, 0, 40, 3, 3, 1, 25, 40),        # ragged columns in a 64-wide tile
])
def test_tap_shared_stage_matches_the_per_tap_kernel(gpu, monkeypatch, mfma, c0, c1, cout, kh, kw, n, H, W):
    """conv_gemm_tapx_kernel (one activation stage per filter row, fragment reads shifted per tap, zero cell for taps
    that leave the image row) against conv_gemm_dma_kernel (VFML_CONV_PER_TAP: one stage per tap): the same products
    in the same order, so BIT-identical outputs - and both against the float64 convolution."""
    from vfml import hip
    from vfml.weights import pack_conv_weight
    g = torch.Generator().manual_seed(7 * c0 + kh + kw + H)
    cin = c0 + c1
    x = torch.randn(n, cin, H, W, generator=g)
    wt = torch.randn(cout, cin, kh, kw, generator=g) / math.sqrt(cin * kh * kw)
    b = torch.randn(cout, generator=g)
    ph, pw = kh // 2, kw // 2
    order = hip.KORDER_CBLOCK64 if mfma == 1 else hip.KORDER_CBLOCK
    w = as_weight(pack_conv_weight(wt, cblock=64 if mfma == 1 else True), cout, "f16x3", order=order)
    if mfma == 1:
        ref = F.conv2d(_f16(x), _f16(wt * w.scale) / w.scale, b.double(), padding=(ph, pw)).float()
    else:
        ref = F.conv2d(x.double(), wt.double(), b.double(), padding=(ph, pw)).float()
    ho, wo = ref.shape[-2:]
    P = n * H * W
    LD = cin + 64
    buf = torch.zeros(P * LD, device=gpu)
    xs = nhwc(x).view(P, cin)
    hip.to_s16(xs[:, :c0].contiguous().reshape(-1), P, c0, c0, buf, LD, dst_off=32)
    if c1:
        hip.to_s16(xs[:, c0:].contiguous().reshape(-1), P, c1, c1, buf, LD, dst_off=32 + c0)
    outs = []
    # (problems this small would get 128 x 64 tiles: force the two shapes the shared-stage kernel is built for)
    monkeypatch.setenv("VFML_DMA_TILE", "2,3,2,2" if cout == 192 else "2,2,4,1" if cout <= 64 else "2,3,4,1" if cout <= 96 else "3,2,2,2")
    for per_tap in (False, True):
        out = torch.full((n * ho * wo * cout,), float("nan"), device=gpu)
        hip.conv2d(buf, c0, LD, n, H, W, w, b.cuda(), cout, kh, kw, out, cout, pad_h=ph, pad_w=pw, in0_off=32,
                   in1=buf if c1 else None, c1=c1, ld1=LD if c1 else 0, in1_off=32 + c0, in_fmt=hip.FMT_S16, mfma=mfma,
                   per_tap=per_tap)
        outs.append(out)
    monkeypatch.delenv("VFML_DMA_TILE")
    got = from_nhwc(outs[0], n, ho, wo, cout)
    assert torch.isfinite(got).all()
    assert rel_err(got, ref) < CONV_TOL["f16x3"], rel_err(got, ref)
    assert torch.equal(outs[0], outs[1])


def test_tap_shared_stage_at_the_1080p_gate_shape(gpu):
    """The 1x5 gate convolution of a 1080p field (3 x 135 x 240 pixels, 128 + 384 channels -> 256) as the dispatcher
    sends it by itself (192 x 128 tiles, shared stage): bit-identical to the per-tap kernel, and a band of output rows
    against the float64 convolution."""
    from vfml import hip
    from vfml.weights import pack_conv_weight
    g = torch.Generator().manual_seed(99)
    n, H, W, c0, c1, cout = 3, 135, 240, 128, 384, 256
    cin = c0 + c1
    x = torch.randn(n, cin, H, W, generator=g)
    wt = torch.randn(cout, cin, 1, 5, generator=g) / math.sqrt(cin * 5)
    b = torch.randn(cout, generator=g)
    w = as_weight(pack_conv_weight(wt, cblock=True), cout, "f16x3", order=hip.KORDER_CBLOCK)
    P = n * H * W
    buf = torch.zeros(P * cin, device=gpu)
    xs = nhwc(x).view(P, cin)
    hip.to_s16(xs[:, :c0].contiguous().reshape(-1), P, c0, c0, buf, cin)
    hip.to_s16(xs[:, c0:].contiguous().reshape(-1), P, c1, c1, buf, cin, dst_off=c0)
    outs = []
    for per_tap in (False, True):
        out = torch.full((P * cout,), float("nan"), device=gpu)
        hip.conv2d(buf, c0, cin, n, H, W, w, b.cuda(), cout, 1, 5, out, cout, pad_w=2, in1=buf, c1=c1, ld1=cin, in1_off=c0,
                   in_fmt=hip.FMT_S16, per_tap=per_tap)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    got = from_nhwc(outs[0], n, H, W, cout)[2:3, :, 130:135]
    ref = F.conv2d(x[2:3, :, 130:135].double(), wt.double(), b.double(), padding=(0, 2)).float()
    assert rel_err(got, ref) < CONV_TOL["f16x3"], rel_err(got, ref)


@pytest.mark.parametrize("n,H,W", [(2, 9, 14), (1, 1, 5), (3, 135, 240)])
def test_tapsum3x3_is_the_3x3_convolution(gpu, n, H, W):
    """vfml_tapsum3x3 over the tap-major 1x1 products == the 3x3 'same' convolution to four channels (zero padding)."""
    from vfml import hip
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(n, 24, H, W, generator=g, dtype=torch.float64)
    wt = torch.randn(4, 24, 3, 3, generator=g, dtype=torch.float64)
    b = torch.randn(4, generator=g, dtype=torch.float64)
    ref = F.conv2d(x, wt, b, padding=1).float()
    w36 = wt.permute(2, 3, 0, 1).reshape(36, 24)                      # row (ky*3+kx)*4 + o
    t = torch.einsum("nchw,kc->nhwk", x, w36).float()                 # the 1x1 convolution's output
    ld = 40
    tp = torch.zeros(n * H * W, ld)
    tp[:, :36] = t.reshape(-1, 36)
    out = torch.full((n * H * W * 4,), float("nan"), device=gpu)
    hip.tapsum3x3(tp.cuda().reshape(-1), ld, b.float().cuda(), n, H, W, out)
    got = out.view(n, H, W, 4).permute(0, 3, 1, 2).cpu()
    assert rel_err(got, ref) < 2e-6
    with pytest.raises(RuntimeError, match="ld_t"):
        hip.tapsum3x3(tp.cuda().reshape(-1), 34, None, n, H, W, out)


@pytest.mark.parametrize("n,H,W", [(1, 540, 960), (2, 6, 64), (1, 9, 32)])
def test_encoder_c64_kernel_is_the_general_convolution(gpu, n, H, W):
    """vfml_conv3x3_c64 (persistent workgroups, weights in registers, one patch per tile) == vfml_conv2d_split on the same
    split-row input, bit for bit, and so are the norm partial sums; ragged tile rows, image borders."""
    from vfml import hip
    from vfml.weights import pack_conv_weight
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(n, 64, H, W, generator=g)
    wt = torch.randn(64, 64, 3, 3, generator=g) / math.sqrt(64 * 9)
    b = torch.randn(64, generator=g) * 0.1
    P = n * H * W
    src = torch.zeros(P * 64, device=gpu)
    hip.to_s16(nhwc(x), P, 64, 64, src, 64)
    Wt = as_weight(pack_conv_weight(wt, cblock=True), 64, "f16x3", order=hip.KORDER_CBLOCK)
    chunks = P // 32
    outs, parts = [], []
    for fused in (False, True):
        out = torch.full((P * 64,), float("nan"), device=gpu)
        part = torch.full((chunks * 64 * 2,), float("nan"), dtype=torch.float64, device=gpu)
        if fused:
            hip.conv3x3_c64(src, 64, n, H, W, Wt, b.cuda(), out, 64, stats_part=part)
        else:
            hip.conv2d(src, 64, 64, n, H, W, Wt, b.cuda(), 64, 3, 3, out, 64, pad_h=1, pad_w=1, stats_part=part, in_fmt=hip.FMT_S16)
        outs.append(out)
        parts.append(part)
    assert torch.isfinite(outs[1]).all() and torch.isfinite(parts[1]).all()
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    assert torch.equal(parts[0], parts[1])
    ref = F.conv2d(x[:1, :, :8].double(), wt.double(), b.double(), padding=1)[:, :, :6].float()
    got = from_nhwc(outs[1], n, H, W, 64)[:1, :, :6]
    assert rel_err(got, ref) < CONV_TOL["f16x3"], rel_err(got, ref)


@pytest.mark.parametrize("n,H,W", [(3, 135, 240), (1, 5, 31), (2, 16, 24)])
def test_flow_half_kernel_is_the_two_convolutions(gpu, n, H, W):
    """vfml_flow_half (7x7 over the flow + ReLU + 3x3 + ReLU, the 128-channel map in LDS) == vfml_flow_rows7 + the two
    one-MFMA convolutions, bit for bit - ragged tiles, image borders (the zero padding of the 128-channel map, not of the
    flow) - and within plain-f16 tolerance of the float64 result."""
    from vfml import hip
    from vfml.weights import pack_conv_weight
    g = torch.Generator().manual_seed(H * W + n)
    flow = (torch.randn(n, 4, H, W, generator=g) * 3.0)
    w1 = torch.randn(128, 4, 7, 7, generator=g) / math.sqrt(4 * 49)
    b1 = torch.randn(128, generator=g) * 0.1
    w2 = torch.randn(64, 128, 3, 3, generator=g) / math.sqrt(128 * 9)
    b2 = torch.randn(64, generator=g) * 0.1
    ref = F.relu(F.conv2d(F.relu(F.conv2d(flow.double(), w1.double(), b1.double(), padding=3)), w2.double(), b2.double(), padding=1)).float()
    P = n * H * W
    # the network's packing of the two layers (vfml/network.py _pack): rows7 layout, 64-channel-block order
    w7 = torch.zeros(128, 32, 7, 1)
    w7[:, :28, :, 0] = w1.permute(0, 3, 1, 2).reshape(128, 28, 7)
    W1 = as_weight(pack_conv_weight(w7, cblock=True), 128, "f16x3", order=hip.KORDER_CBLOCK)
    W2 = as_weight(pack_conv_weight(w2, cblock=64), 64, "f16x3", order=hip.KORDER_CBLOCK64)
    flow4 = nhwc(flow)
    # three launches
    rows = torch.zeros(P * 32, device=gpu)
    hip.flow_rows7(flow4, n, H, W, rows)
    f1 = torch.zeros(P * 128, device=gpu)
    hip.conv2d(rows, 32, 32, n, H, W, W1, b1.cuda(), 128, 7, 1, f1, 128, pad_h=3, epilogue=hip.EPI_RELU, in_fmt=hip.FMT_S16,
               out_fmt=hip.FMT_S16, mfma=1)
    two = torch.full((P * 256,), 3.0, device=gpu)
    hip.conv2d(f1, 128, 128, n, H, W, W2, b2.cuda(), 64, 3, 3, two, 256, out_off=192, pad_h=1, pad_w=1, epilogue=hip.EPI_RELU,
               in_fmt=hip.FMT_S16, out_fmt=hip.FMT_S16, mfma=1)
    # one launch
    one = torch.full((P * 256,), 3.0, device=gpu)
    hip.flow_half(flow4, n, H, W, W1, b1.cuda(), W2, b2.cuda(), one, 256, out_off=192)
    assert torch.equal(one.view(torch.int32), two.view(torch.int32))
    got = s16_decode(one.view(P, 256)[:, 192:].contiguous().reshape(-1), P, 64, 64)
    got = got.view(n, H, W, 64).permute(0, 3, 1, 2).cpu()
    assert rel_err(got, ref) < 4e-3, rel_err(got, ref)


def test_bidirectional_lookup_is_two_lookups(gpu):
    """vfml_corr_lookup_indirect_bidir == the forward and the backward lookup as two launches, bit for bit (fewer query maps
    than the table holds per direction, split-row output, tiled volumes with an f16 level)."""
    from vfml import hip
    g = torch.Generator().manual_seed(21)
    h, w, R, M, nm = 12, 20, 4, 3, 2
    Pn = h * w
    tile = hip.VolTile(3, 2)
    hl = [h >> l for l in range(4)]
    wl = [w >> l for l in range(4)]
    ldl = [((tile.count(hl[l], wl[l]) + 31) // 32) * 32 + 32 for l in range(4)]
    rows = tile.count(h, w)
    pyr = {}
    for d in ("f", "b"):
        pyr[d] = []
        for m in range(M):
            lv = []
            for l in range(4):
                t = torch.randn(rows * ldl[l], generator=g)
                lv.append(t.half().cuda().view(torch.float32) if l == 3 else t.cuda())
            pyr[d].append(lv)
    tabs = {k: torch.zeros(64, dtype=torch.int64, device=gpu) for k in ("f", "b", "fb")}
    hip.ptr_table_set(tabs["f"], [p for m in pyr["f"] for p in m])
    hip.ptr_table_set(tabs["b"], [p for m in pyr["b"] for p in m])
    hip.ptr_table_set(tabs["fb"], [p for d in ("f", "b") for m in pyr[d] for p in m])
    coords = (torch.rand(M * Pn, 4, generator=g) * torch.tensor([w, h, w, h]) * 1.2 - 2.0).reshape(-1).cuda()
    cor_p = 336
    VF = hip.vol_f16_levels(8)
    outs = []
    for fused in (False, True):
        out = torch.zeros(M * Pn * 2 * cor_p, device=gpu)
        if fused:
            hip.corr_lookup(None, hl, wl, ldl, R, Pn, coords, 0, 4, out, 0, 2 * cor_p, out_fmt=hip.FMT_S16, table=tabs["fb"],
                            nmaps=nm, vol_fmt=VF, vol_tile=tile.code, bidir=(2, cor_p, M))
        else:
            hip.corr_lookup(None, hl, wl, ldl, R, Pn, coords, 0, 4, out, 0, 2 * cor_p, out_fmt=hip.FMT_S16, table=tabs["f"],
                            nmaps=nm, vol_fmt=VF, vol_tile=tile.code)
            hip.corr_lookup(None, hl, wl, ldl, R, Pn, coords, 2, 4, out, cor_p, 2 * cor_p, out_fmt=hip.FMT_S16, table=tabs["b"],
                            nmaps=nm, vol_fmt=VF, vol_tile=tile.code)
        outs.append(out)
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    assert outs[0][:nm * Pn * 2 * cor_p].abs().max() > 0 and (outs[0][nm * Pn * 2 * cor_p:] == 0).all()


@pytest.mark.parametrize("parts", [1, 2])
def test_tapsum_update_is_tapsum_then_coords_update(gpu, parts):
    """vfml_tapsum3x3_update == vfml_tapsum3x3 + vfml_coords_update, bit for bit (coords, the f32 flow and the flow quad of a
    split-row unit)."""
    from vfml import hip
    g = torch.Generator().manual_seed(5 + parts)
    n, H, W, ld = 2, 11, 19, 36
    P = n * H * W
    taps = torch.randn(parts * P * ld, generator=g).cuda()
    bias = torch.randn(4, generator=g).cuda()
    res = []
    for fused in (False, True):
        coords = torch.empty(P * 4, device=gpu)
        hip.coords_init(coords, n, H, W)
        flow = torch.zeros(P * 4, device=gpu)
        wide = torch.zeros(P * 16, device=gpu)
        if fused:
            hip.tapsum3x3_update(taps, ld, bias, n, H, W, coords, parts=parts, part_stride=P * ld, flow_a=flow, ld_a=4,
                                 flow_b=wide, ld_b=16, flow_b_off=12, fmt_b=hip.FMT_S16)
        else:
            delta = torch.empty(P * 4, device=gpu)
            hip.tapsum3x3(taps, ld, bias, n, H, W, delta, parts=parts, part_stride=P * ld)
            hip.coords_update(coords, delta, n, H, W, flow_a=flow, ld_a=4, flow_b=wide, ld_b=16, flow_b_off=12,
                              fmt_b=hip.FMT_S16)
        res.append((coords, flow, wide))
    for a, b in zip(*res):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    assert res[0][1].abs().max() > 0


@pytest.mark.parametrize("n,H,W,tile", [(3, 135, 240, None), (1, 17, 23, "3,2,2,2"), (2, 9, 14, "2,2,2,2")])
def test_projection_epilogue_is_the_two_layer_flow_head(gpu, monkeypatch, n, H, W, tile):
    """vfml_conv_desc.proj_out: conv 3x3 (128 -> 256) + ReLU + 1x1 (256 -> 36) in one launch - the partial maps of the two
    128-column tiles add up to what the two launches give (same products, another order of the last additions) and to the
    float64 result; through the tap sum: the 3x3 -> 4-channel flow head.  Ragged sizes on both tile shapes; nothing is
    written outside the partial maps; the route refuses what it is not built for."""
    from vfml import hip
    from vfml.weights import pack_conv_weight
    g = torch.Generator().manual_seed(H * W + n)
    cin, cmid = 128, 256
    x = torch.randn(n, cin, H, W, generator=g)
    w1 = torch.randn(cmid, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b1 = torch.randn(cmid, generator=g) * 0.1
    w2 = torch.randn(4, cmid, 3, 3, generator=g) / math.sqrt(cmid * 9)
    b2 = torch.randn(4, generator=g)
    ref = F.conv2d(F.relu(F.conv2d(x.double(), w1.double(), b1.double(), padding=1)), w2.double(), b2.double(), padding=1).float()
    P = n * H * W
    LD = 768
    buf = torch.zeros(P * LD, device=gpu)
    hip.to_s16(nhwc(x), P, cin, cin, buf, LD, dst_off=256)
    W1 = as_weight(pack_conv_weight(w1, cblock=True), cmid, "f16x3", order=hip.KORDER_CBLOCK)
    W2 = as_weight(w2.permute(2, 3, 0, 1).reshape(36, cmid), 36, "f16x3", order=hip.KORDER_CBLOCK)
    if tile:
        monkeypatch.setenv("VFML_DMA_TILE", tile)
    # two launches
    fh = torch.zeros(P * cmid, device=gpu)
    hip.conv2d(buf, cin, LD, n, H, W, W1, b1.cuda(), cmid, 3, 3, fh, cmid, in0_off=256, pad_h=1, pad_w=1, epilogue=hip.EPI_RELU,
               in_fmt=hip.FMT_S16, out_fmt=hip.FMT_S16)
    taps = torch.zeros(P * 36, device=gpu)
    if tile:
        monkeypatch.delenv("VFML_DMA_TILE")
    hip.conv2d(fh, cmid, cmid, n, H, W, W2, None, 36, 1, 1, taps, 36, in_fmt=hip.FMT_S16)
    two = torch.empty(P * 4, device=gpu)
    hip.tapsum3x3(taps, 36, b2.cuda(), n, H, W, two)
    # one launch
    if tile:
        monkeypatch.setenv("VFML_DMA_TILE", tile)
    parts = torch.full((2 * P * 36 + 64,), float("nan"), device=gpu)
    fh2 = torch.full((P * cmid,), 7.0, device=gpu)
    hip.conv2d(buf, cin, LD, n, H, W, W1, b1.cuda(), cmid, 3, 3, fh2, cmid, in0_off=256, pad_h=1, pad_w=1, epilogue=hip.EPI_RELU,
               in_fmt=hip.FMT_S16, out_fmt=hip.FMT_S16, proj=W2, proj_out=parts, ld_proj=36)
    assert torch.isnan(parts[2 * P * 36:]).all() and torch.isfinite(parts[:2 * P * 36]).all()
    assert (fh2 == 7.0).all()                      # the 256-channel map is not stored
    one = torch.empty(P * 4, device=gpu)
    hip.tapsum3x3(parts, 36, b2.cuda(), n, H, W, one, parts=2, part_stride=P * 36)
    psum = (parts[:P * 36] + parts[P * 36:2 * P * 36]).cpu()
    assert rel_err(psum, taps.cpu()) < 2e-6, rel_err(psum, taps.cpu())
    got1, got2 = from_nhwc(one, n, H, W, 4), from_nhwc(two, n, H, W, 4)
    assert rel_err(got1, got2) < 2e-6
    assert rel_err(got1, ref) < CONV_TOL["f16x3"], rel_err(got1, ref)
    # "2a" (activations as plain f16, in the projection as in the convolution) against its own two launches
    hip.conv2d(buf, cin, LD, n, H, W, W1, b1.cuda(), cmid, 3, 3, fh, cmid, in0_off=256, pad_h=1, pad_w=1, epilogue=hip.EPI_RELU,
               in_fmt=hip.FMT_S16, out_fmt=hip.FMT_S16, mfma="2a")
    if tile:
        monkeypatch.delenv("VFML_DMA_TILE")
    hip.conv2d(fh, cmid, cmid, n, H, W, W2, None, 36, 1, 1, taps, 36, in_fmt=hip.FMT_S16, mfma="2a")
    if tile:
        monkeypatch.setenv("VFML_DMA_TILE", tile)
    hip.conv2d(buf, cin, LD, n, H, W, W1, b1.cuda(), cmid, 3, 3, fh2, cmid, in0_off=256, pad_h=1, pad_w=1,
               epilogue=hip.EPI_RELU, in_fmt=hip.FMT_S16, out_fmt=hip.FMT_S16, proj=W2, proj_out=parts, ld_proj=36, mfma="2a")
    psum = (parts[:P * 36] + parts[P * 36:2 * P * 36]).cpu()
    assert rel_err(psum, taps.cpu()) < 2e-6, rel_err(psum, taps.cpu())
    with pytest.raises(RuntimeError, match="proj_out"):        # the other reduced products are not built with it
        hip.conv2d(buf, cin, LD, n, H, W, W1, b1.cuda(), cmid, 3, 3, fh2, cmid, in0_off=256, pad_h=1, pad_w=1,
                   epilogue=hip.EPI_RELU, in_fmt=hip.FMT_S16, out_fmt=hip.FMT_S16, proj=W2, proj_out=parts, ld_proj=36, mfma=1)
    with pytest.raises(RuntimeError, match="proj_out"):        # nor another epilogue
        hip.conv2d(buf, cin, LD, n, H, W, W1, b1.cuda(), cmid, 3, 3, fh2, cmid, in0_off=256, pad_h=1, pad_w=1,
                   in_fmt=hip.FMT_S16, out_fmt=hip.FMT_S16, proj=W2, proj_out=parts, ld_proj=36)


def _as_f32_storage(half_tensor):
    """A float16 tensor's bytes as the float32 buffer the ctypes wrappers take (even element count)."""
    return half_tensor.contiguous().view(torch.float32)


def test_gemm_form_writes_f16_volumes(gpu):
    """VFML_FMT_F16 out / out_t of the GEMM form: every element is the round-to-nearest f16 of what the f32 form stores,
    ragged tiles in both directions, nothing written outside; narrower than 1024 columns too (the form is forced); the
    other forms refuse the format."""
    from vfml import hip
    g = torch.Generator().manual_seed(77)
    D = 256

    def rows(f):
        t = torch.empty(f.numel(), device=gpu)
        hip.to_s16((f * 16.0).cuda().reshape(-1), f.shape[0], D, D, t, D)
        return t

    def planes(f):
        return hip.SplitWeight(f.shape[0], D, torch.device("cuda")).fill(f.cuda().reshape(-1).contiguous(), scale=16.0)

    for P, S, dual in ((1300, 1412, True), (520, 480, False)):
        f1, f2 = torch.randn(P, D, generator=g), torch.randn(S, D, generator=g)
        ld, ldt = (S + 31) // 32 * 32, (P + 31) // 32 * 32
        out32 = torch.zeros(P * ld, device=gpu)
        hip.conv2d(rows(f1), D, D, 1, 1, P, planes(f2), None, S, 1, 1, out32, ld, out_scale=1.0 / 256.0, in_fmt=hip.FMT_S16)
        if S < 1024:     # (the f32 call of this width went through the convolution form: same sums up to rounding)
            ref = (f1.double() @ f2.double().t() / 16.0).float()
            assert rel_err(out32.view(P, ld)[:, :S].cpu(), ref) < CONV_TOL["f16x3"]
        out16 = torch.full((P * ld,), 7.0, dtype=torch.float16, device=gpu)
        out16_t = torch.full((S * ldt,), 9.0, dtype=torch.float16, device=gpu)
        hip.conv2d(rows(f1), D, D, 1, 1, P, planes(f2), None, S, 1, 1, _as_f32_storage(out16), ld, out_scale=1.0 / 256.0,
                   in_fmt=hip.FMT_S16, out_fmt=hip.FMT_F16, out_t=_as_f32_storage(out16_t) if dual else None,
                   ld_out_t=ldt if dual else 0)
        got = out16.view(P, ld)
        if S >= 1024:
            assert torch.equal(got[:, :S], out32.view(P, ld)[:, :S].half())
        else:
            assert rel_err(got[:, :S].float().cpu(), out32.view(P, ld)[:, :S].cpu()) < 1e-3
        assert (got[:, S:] == 7.0).all()
        if dual:
            gt = out16_t.view(S, ldt)
            assert torch.equal(gt[:, :P], got[:, :S].t())
            assert (gt[:, P:] == 9.0).all()
    with pytest.raises(RuntimeError, match="VFML_FMT_F16"):         # a 3x3 convolution cannot write it
        x = torch.zeros(64 * 64, device=gpu)
        w3 = hip.SplitWeight(64, 9 * 64, torch.device("cuda")).fill(torch.zeros(64 * 9 * 64, device=gpu))
        hip.conv2d(x, 64, 64, 1, 8, 8, w3, None, 64, 3, 3, torch.zeros(64 * 64, device=gpu), 64, pad_h=1, pad_w=1,
                   in_fmt=hip.FMT_S16, out_fmt=hip.FMT_F16)


@pytest.mark.parametrize("radius,levels", [(4, 4), (3, 3)])
def test_corr_lookup_reads_f16_volumes(gpu, radius, levels):
    """vol_fmt VFML_FMT_F16: the lookup over a pyramid of f16 values == the lookup over the same values widened to f32
    (same arithmetic on the same numbers: bit-identical), in both output formats."""
    from vfml import hip
    g = torch.Generator().manual_seed(5 + radius)
    h, w = 24, 40
    P = h * w
    hl = [h >> l for l in range(levels)]
    wl = [w >> l for l in range(levels)]
    ld = [(a * b + 31) // 32 * 32 for a, b in zip(hl, wl)]
    vol16 = [torch.randn(P * l, generator=g).half().cuda() for l in ld]
    vol32 = [v.float() for v in vol16]
    coords = (torch.rand(P, 4, generator=g) * torch.tensor([w + 6.0, h + 6.0, w, h]) - 3.0).cuda().reshape(-1)
    nch = levels * (2 * radius + 1) ** 2
    ldo = (nch + 7) // 8 * 8
    for fmt in (hip.FMT_F32, hip.FMT_S16):
        a = torch.zeros(P * ldo, device=gpu)
        b = torch.zeros(P * ldo, device=gpu)
        hip.corr_lookup(vol32, hl, wl, ld, radius, P, coords, 0, 4, a, 0, ldo, out_fmt=fmt)
        hip.corr_lookup([_as_f32_storage(v) for v in vol16], hl, wl, ld, radius, P, coords, 0, 4, b, 0, ldo, out_fmt=fmt,
                        vol_fmt=hip.FMT_F16)
        assert torch.equal(a, b)
    with pytest.raises(RuntimeError, match="vol_fmt"):
        hip.corr_lookup([_as_f32_storage(v) for v in vol16], hl, wl, ld, 2, P, coords, 0, 4, a, 0, ldo, vol_fmt=hip.FMT_F16)
    # VFML_VOL_F16_LEVELS(m): only the levels of m hold f16 texels (cfg.corr_volume 'f16@k': levels k.. of the pyramid)
    for mask in (14, 12, 8):
        m = mask & ((1 << levels) - 1)
        if m == 0:
            continue
        mixed = [_as_f32_storage(vol16[l]) if (m >> l) & 1 else vol32[l] for l in range(levels)]
        for fmt in (hip.FMT_F32, hip.FMT_S16):
            a = torch.zeros(P * ldo, device=gpu)
            b = torch.zeros(P * ldo, device=gpu)
            hip.corr_lookup(vol32, hl, wl, ld, radius, P, coords, 0, 4, a, 0, ldo, out_fmt=fmt)
            hip.corr_lookup(mixed, hl, wl, ld, radius, P, coords, 0, 4, b, 0, ldo, out_fmt=fmt, vol_fmt=hip.vol_f16_levels(mask))
            assert torch.equal(a, b), (mask, fmt)
    with pytest.raises(RuntimeError, match="vol_fmt"):
        hip.corr_lookup(vol32, hl, wl, ld, radius, P, coords, 0, 4, a, 0, ldo, vol_fmt=hip.vol_f16_levels(5))


@pytest.mark.parametrize("dual", [False, True])
def test_gemm_form_split_k(gpu, dual):
    """vfml_conv_desc.ksplit_ws: a GEMM with few tiles and a long K axis runs as two work items per tile (one per half of
    K, the second half's sums added by a pass of the call): same result as the one-pass call up to the order of one f32
    addition per element, pad columns untouched, with and without the transposed second output; K not a multiple of the
    half (the second half reads past K: zeros)."""
    from vfml import hip
    g = torch.Generator().manual_seed(3)
    P, S, D = 128, 2000, 4160 + 32          # 1 x 16 tiles of 128 x 128, 131 steps of 32 channels (odd)
    f1, f2 = torch.randn(P, D, generator=g) / 8, torch.randn(S, D, generator=g) / 8
    ld, ldt = (S + 31) // 32 * 32, P

    def rows(f):
        t = torch.empty(f.numel(), device=gpu)
        hip.to_s16(f.cuda().reshape(-1), f.shape[0], D, D, t, D)
        return t

    w = hip.SplitWeight(S, D, torch.device("cuda")).fill(f2.cuda().reshape(-1).contiguous(), scale=16.0)
    ref = (f1.double() @ f2.double().t()).float()
    outs = []
    for ws in (None, torch.full((P * ld + S * ldt,), float("nan"), device=gpu)):
        out = torch.full((P * ld,), 7.0, device=gpu)
        out_t = torch.full((S * ldt,), 9.0, device=gpu) if dual else None
        hip.conv2d(rows(f1), D, D, 1, 1, P, w, None, S, 1, 1, out, ld, in_fmt=hip.FMT_S16, out_t=out_t,
                   ld_out_t=ldt if dual else 0, ksplit_ws=ws)
        got = out.view(P, ld).cpu()
        assert rel_err(got[:, :S], ref) < CONV_TOL["f16x3"]
        assert (got[:, S:] == 7.0).all()
        if dual:
            assert torch.equal(out_t.view(S, ldt).cpu(), got[:, :S].t())
        outs.append(got)
    assert not torch.equal(outs[0], outs[1])            # (the split call did take the other route)
    assert rel_err(outs[0][:, :S], outs[1][:, :S]) < 5e-6


@pytest.mark.parametrize("n,chunks,c", [(2, 3000, 64), (1, 16200, 96), (1, 1023, 64), (1, 2048, 12)])
def test_instnorm_finalize_many_partials(gpu, n, chunks, c):
    """vfml_instnorm_finalize over thousands of partials per channel (the split-row convolutions leave one per 32 pixels):
    the slice-wise first pass (c % 8 == 0, >= 1024 chunks) and the direct kernel agree with float64 sums; repeatable."""
    from vfml import hip
    g = torch.Generator().manual_seed(chunks)
    hw = chunks * 32
    x = torch.rand(n, chunks, c, 32, generator=g, dtype=torch.float64) * 2 - 0.5
    part = torch.stack([x.sum(-1), (x * x).sum(-1)], -1).contiguous().cuda()          # [n][chunks][c][2] doubles
    mean = x.permute(0, 2, 1, 3).reshape(n, c, -1).mean(-1)
    var = x.permute(0, 2, 1, 3).reshape(n, c, -1).var(-1, unbiased=False)
    want = torch.stack([mean, 1.0 / torch.sqrt(var + 1e-5)], -1).float()
    st = torch.empty(n * c * 2, device=gpu)
    hip.instnorm_finalize(part, n, chunks, c, hw, st)
    got = st.view(n, c, 2).cpu()
    assert torch.allclose(got, want, rtol=2e-6, atol=1e-7)
    st2 = torch.empty_like(st)
    hip.instnorm_finalize(part, n, chunks, c, hw, st2)
    assert torch.equal(st, st2)


def test_flow_rows7_turns_the_7x7_flow_convolution_into_7x1(gpu):
    """vfml_flow_rows7 + a 7x1 convolution over its 32 channels == the 7x7 'same' convolution over the 4-channel flow map."""
    from vfml import hip
    from vfml.weights import pack_conv_weight
    g = torch.Generator().manual_seed(12)
    n, H, W, cout = 2, 11, 17, 128
    x = torch.randn(n, 4, H, W, generator=g) * 3
    wt = torch.randn(cout, 4, 7, 7, generator=g) / 14
    b = torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=3)).float()
    P = n * H * W
    rows = torch.full((P * 32,), float("nan"), device=gpu)
    hip.flow_rows7(nhwc(x), n, H, W, rows)
    back = torch.empty(P * 32, device=gpu)                      # split rows -> f32 through a 1x1 identity is overkill: check
    w7 = torch.zeros(cout, 32, 7, 1)                            # the rows through the convolution itself
    w7[:, :28, :, 0] = wt.permute(0, 3, 1, 2).reshape(cout, 28, 7)
    w = as_weight(pack_conv_weight(w7, cblock=True), cout, "f16x3", order=hip.KORDER_CBLOCK)
    out = torch.full((P * cout,), float("nan"), device=gpu)
    hip.conv2d(rows, 32, 32, n, H, W, w, b.cuda(), cout, 7, 1, out, cout, pad_h=3, epilogue=hip.EPI_RELU, in_fmt=hip.FMT_S16)
    got = from_nhwc(out, n, H, W, cout)
    assert torch.isfinite(got).all()
    assert rel_err(got, ref) < CONV_TOL["f16x3"], rel_err(got, ref)
    del back


@pytest.mark.parametrize("n,H,W", [(1, 64, 128), (2, 70, 90), (1, 136, 200), (1, 1080, 1920)])
def test_stem_kernel_matches_the_general_convolution(gpu, n, H, W):
    """vfml_stem7x7s2 (csrc/stem.hip: the encoders' 7x7 / 2 stem with the tile's input patch and all weights resident in
    LDS): against a float64 convolution, against the general split-f16 kernel it replaces (same three-term products, other
    grouping of the K axis: equal to f32 rounding), and its per-tile statistics partials - folded by vfml_instnorm_finalize
    - against vfml_instnorm_stats over the stored map.  Exact tile multiples, ragged edges in both directions (odd output
    sizes), two images, and the 1080p frame."""
    from vfml import hip
    from vfml.weights import pack_conv_weight
    g = torch.Generator().manual_seed(H + W)
    x = torch.rand(n, 3, H, W, generator=g) * 2 - 1
    wt = torch.randn(64, 3, 7, 7, generator=g) / (3 * 49) ** 0.5
    b = torch.randn(64, generator=g)
    ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    hw = ho * wo
    x4 = torch.cat([x, torch.zeros(n, 1, H, W)], dim=1)
    xd = nhwc(x4)
    sw = hip.pack_stem_weight(wt.cuda(), torch.device("cuda"))
    out = torch.full((n * hw * 64 + 64,), float("nan"), device=gpu)
    chunks = hip.stem_chunks(H, W)
    part = torch.full((n * chunks * 64 * 2 + 8,), float("nan"), device=gpu, dtype=torch.float64)
    hip.stem7x7s2(xd, n, H, W, sw, b.to(gpu), out, stats_part=part)
    torch.cuda.synchronize()
    assert torch.isnan(out[n * hw * 64:]).all() and not torch.isnan(out[:n * hw * 64]).any()
    assert torch.isnan(part[n * chunks * 64 * 2:]).all() and not torch.isnan(part[:n * chunks * 64 * 2]).any()
    got = from_nhwc(out[:n * hw * 64], n, ho, wo, 64)
    ref = F.conv2d(x.double(), wt.double(), b.double(), stride=2, padding=3)
    assert rel_err(got.double(), ref) < CONV_TOL["f16x3"], rel_err(got.double(), ref)
    # the general kernel on the same operands (tap-major weights over the padded 4 channels)
    wobj = as_weight(pack_conv_weight(wt, cin_pad=4), 64, "f16x3")
    old = torch.empty(n * hw * 64, device=gpu)
    hip.conv2d(xd, 4, 4, n, H, W, wobj, b.to(gpu), 64, 7, 7, old, 64, stride=2, pad_h=3, pad_w=3)
    assert rel_err(out[:n * hw * 64].cpu(), old.cpu()) < 2e-6
    # statistics: partials folded == the stand-alone pass over the stored map; means against float64
    st_a = torch.empty(n * 64 * 2, device=gpu)
    st_b = torch.empty_like(st_a)
    hip.instnorm_finalize(part, n, chunks, 64, hw, st_a)
    ws = torch.empty(hip.instnorm_workspace_bytes(n, hw, 64) // 8 + 1, device=gpu, dtype=torch.float64)
    hip.instnorm_stats(out[:n * hw * 64].clone(), n, hw, 64, st_b, ws)
    assert torch.allclose(st_a, st_b, rtol=2e-7, atol=1e-9), (st_a - st_b).abs().max().item()
    assert torch.allclose(st_a.view(n, 64, 2)[..., 0].cpu().double(), got.double().mean(dim=(2, 3)), rtol=1e-5, atol=1e-6)
    # without the partials: same map
    out2 = torch.empty(n * hw * 64, device=gpu)
    hip.stem7x7s2(xd, n, H, W, sw, None, out2)
    assert torch.allclose(out2.view(-1, 64) + b.to(gpu), out[:n * hw * 64].view(-1, 64), rtol=0, atol=1e-5)
    if (H, W) == (1080, 1920):
        for fn, name in ((lambda: hip.stem7x7s2(xd, n, H, W, sw, b.to(gpu), out, stats_part=part), "stem kernel"),
                         (lambda: hip.conv2d(xd, 4, 4, n, H, W, wobj, b.to(gpu), 64, 7, 7, old, 64, stride=2, pad_h=3, pad_w=3,
                                             stats_part=torch.empty(n * ((hw + 127) // 128) * 64 * 2, device=gpu, dtype=torch.float64)),
                          "general kernel")):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            print(f"stem 1080p: {name} {us:.1f} us per launch = {2.0 * hw * 196 * 64 / us / 1e6:.1f} TFLOP/s algorithmic")
