"""MemFlow pair path (BASELINE config C4) against its CPU oracle, and the new
entry points it needs (row softmax into split rows, transposed split planes, residual-add epilogue)."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
EPE_TOL = 1e-3


def _pair(seed=0, **over):
    from oracle import memflow_oracle as mm
    from vfml.memflow_net import build_memflow_network, memflow_cfg, seeded_memflow_state_dict
    cfg, ocfg = memflow_cfg(), mm.get_cfg()
    for k, v in over.items():
        setattr(cfg, k, v)
        setattr(ocfg, k, v)
    sd = seeded_memflow_state_dict(cfg, seed)
    net = build_memflow_network(cfg)
    net.load_state_dict(sd)
    net.cuda().eval()
    ora = mm.build_network(ocfg)
    ora.load_state_dict(sd)
    ora.eval()
    return net, ora


@pytest.mark.parametrize("precision", ["f16x3"])
def test_memflow_forward_matches_oracle(gpu, precision):
    net, ora = _pair(precision=precision)
    x = torch.rand(1, 2, 3, 128, 192, generator=torch.Generator().manual_seed(3)) * 2 - 1
    low_ref, ref = ora(x)
    low, got = net(x.cuda())
    assert got.shape == ref.shape == (1, 2, 128, 192) and low.shape == low_ref.shape
    epe = (got.cpu() - ref).pow(2).sum(1).sqrt()
    print(f"[memflow {precision}] mean EPE {epe.mean().item():.3e} px, max {epe.max().item():.3e}, |flow| {ref.abs().mean().item():.2f}")
    assert epe.mean().item() < EPE_TOL
    assert (low.cpu() - low_ref).abs().max().item() < 1e-3


def test_memflow_plain_attention_plane_at_a_small_size(gpu, monkeypatch):
    """256 x 320: 1280 keys, the smallest size class that takes the one-f16-per-probability read-out; both read-out
    paths against the oracle, and against each other."""
    from vfml import memflow_net
    net, ora = _pair()
    x = torch.rand(1, 2, 3, 256, 320, generator=torch.Generator().manual_seed(11)) * 2 - 1
    _, ref = ora(x)
    _, plain = net(x.cuda())
    plain = plain.clone()
    assert len(net._att_planes) == 1                      # the plane path ran
    monkeypatch.setattr(memflow_net, "ATT_PLAIN", False)
    _, rows = net(x.cuda())
    for got in (plain, rows):
        assert (got.cpu() - ref).pow(2).sum(1).sqrt().mean().item() < 2e-5
    assert (plain - rows).pow(2).sum(1).sqrt().mean().item() < 2e-5


def test_memflow_1080p_matches_oracle(gpu):
    """One pair at the size of BASELINE config C4: 32400 keys per attention row - the probabilities average 3e-5,
    which is where their storage scale (memflow_net.ATT_SCALE) matters."""
    import time
    net, ora = _pair()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    x = torch.rand(1, 2, 3, 1080, 1920, generator=torch.Generator().manual_seed(8)) * 2 - 1
    x = torch.nn.functional.avg_pool2d(x[0], 9, 1, 4)[None]          # some structure for the matcher
    t0 = time.time()
    with torch.no_grad():
        _, ref = ora(x)
    dt = time.time() - t0
    _, got = net(x.cuda())
    epe = (got.cpu() - ref).pow(2).sum(1).sqrt()
    print(f"[memflow 1080p] mean EPE {epe.mean().item():.3e} px, max {epe.max().item():.3e}, |flow| {ref.abs().mean().item():.2f}, "
          f"oracle {dt:.0f} s")
    assert epe.mean().item() < EPE_TOL


def test_readout_changes_the_flow(gpu):
    """gamma = 0 switches the memory read-out off: the result must differ from gamma = 0.5 (the read-out
    path is live) and still match its own oracle."""
    net, ora = _pair()
    x = torch.rand(1, 2, 3, 128, 128, generator=torch.Generator().manual_seed(4)) * 2 - 1
    _, a = net(x.cuda())
    a = a.clone()
    with torch.no_grad():
        net.update_block.gamma.zero_()
        ora.update_block.gamma.zero_()
    _, b = net(x.cuda())
    assert (a - b).abs().max().item() > 1e-3
    assert (b.cpu() - ora(x)[1]).pow(2).sum(1).sqrt().mean().item() < EPE_TOL


def test_softmax_rows_and_transposed_planes(gpu):
    from tests_support import s16_decode
    from vfml import hip
    g = torch.Generator().manual_seed(5)
    rows, cols, ld = 37, 203, 224
    x = torch.randn(rows, ld, generator=g) * 4
    out = torch.full((rows * 208,), 9.0, device=gpu)
    hip.softmax_rows_s16(x.cuda().reshape(-1), rows, cols, ld, out, 208)
    got = s16_decode(out, rows, 208, 208)
    ref = torch.softmax(x[:, :cols].double(), dim=-1).float()
    assert (got[:, cols:] == 0).all()
    assert (got[:, :cols] - ref).abs().max().item() < 5e-7   # 22-bit split rows
    # the 1080p attention row (register-resident kernel, 16-byte loads), an input stride that forbids them, and a
    # row too long for the registers (three-sweep kernel)
    for rows, cols, ld, ldo in ((5, 32400, 32416, 32416), (7, 1001, 1003, 1008), (3, 40000, 40000, 40000)):
        x = torch.randn(rows, ld, generator=g) * 3
        ref = torch.softmax(x[:, :cols].double(), dim=-1)
        for scale, rel, floor in ((16384.0, 2e-6, 1e-11), (1.0, 2e-6, 7e-8)):    # unscaled: f16 subnormal steps of 6e-8
            out = torch.full((rows * ldo,), 9.0, device=gpu)
            hip.softmax_rows_s16(x.cuda().reshape(-1), rows, cols, ld, out, ldo, scale=scale)
            got = s16_decode(out, rows, ldo, ldo).double() / scale
            assert (got[:, cols:] == 0).all()
            assert ((got[:, :cols] - ref).abs() <= rel * ref + floor).all(), (rows, cols, scale)
        assert (got[:, :cols].sum(dim=1) - 1).abs().max().item() < 2e-3    # unscaled rows lose mass to truncation
    v = torch.randn(50, 24, generator=g)
    sw = hip.SplitWeight(24, 50, gpu).fill_transposed(v.cuda().reshape(-1), 50, ld=24, scale=4.0)
    rec = (sw.hi.view(24, sw.kp).float() + sw.lo.view(24, sw.kp).float()).cpu() / 4.0
    assert (rec[:, 50:] == 0).all()
    assert ((rec[:, :50] - v.t()).abs() <= 2.0 ** -20 * v.t().abs() + 2.0 ** -22).all()


@pytest.mark.parametrize("P", [1100, 2500])
def test_readout_with_the_attention_as_a_plain_f16_plane(gpu, P):
    """The shipped read-out: scores -> vfml_softmax_rows_f16 (one f16 per probability, times 2^14) as the weight plane,
    V^T as split rows, one GEMM  out_t[P][128] = A . V, then m + gamma * out_t into split rows."""
    from tests_support import s16_decode
    from vfml import hip
    g = torch.Generator().manual_seed(P)
    d, scale, gamma = 128, 16384.0, 0.37
    kp = (P + 31) // 32 * 32
    scores = torch.randn(P, kp, generator=g) * 2
    V = torch.randn(P, d, generator=g)
    m = torch.randn(P, d, generator=g)
    A = torch.softmax(scores[:, :P].double(), dim=-1)
    w = hip.PlainWeight(P, kp, gpu, scale=scale)
    w.hi.fill_(7.0)
    hip.softmax_rows_f16(scores.cuda().reshape(-1), P, P, kp, w)
    plane = w.hi.view(P, kp).float().cpu() / scale
    assert (plane[:, P:] == 0).all()
    assert ((plane[:, :P].double() - A).abs() <= 2.0 ** -11 * A + 1e-9).all()          # round to nearest: half an ulp
    vt = torch.full((d * kp,), 5.0, device=gpu)
    hip.transpose_to_s16(V.cuda().reshape(-1), P, d, d, vt, kp, scale=16.0)
    vdec = s16_decode(vt, d, kp, kp) / 16.0
    assert (vdec[:, P:] == 0).all()
    assert ((vdec[:, :P] - V.t()).abs() <= 2.0 ** -20 * V.t().abs() + 2.0 ** -22).all()
    out = torch.empty(d * kp, device=gpu)
    out_t = torch.empty(P * d, device=gpu)
    hip.conv2d(vt, kp, kp, d, 1, 1, w, None, P, 1, 1, out, kp, out_scale=1.0 / 16.0, in_fmt=hip.FMT_S16, out_t=out_t, ld_out_t=d)
    ref = plane[:, :P].double() @ V.double()               # with the plane's own rounding: the GEMM itself is fp32-grade
    got_t = out_t.view(P, d).cpu().double()
    assert (got_t - ref).abs().max().item() < 2e-6 * ref.abs().max().item() + 1e-6
    assert (out.view(d, kp)[:, :P].cpu().double() - ref.t()).abs().max().item() < 2e-6 * ref.abs().max().item() + 1e-6
    # against the exact read-out: at most half an f16 ulp per probability, |sum p eps v| <= 2^-12 sum p |v|
    assert ((got_t - A @ V.double()).abs() <= 2.0 ** -11 * (A @ V.double().abs()) + 1e-6).all()
    m16 = torch.empty(P * d, device=gpu)
    hip.to_s16(m.cuda().reshape(-1), P, d, d, m16, d)
    res = torch.empty(P * d, device=gpu)
    hip.add_to_s16(out_t, d, m16, d, res, d, P, d, scale=gamma)
    want = m.double() + gamma * got_t
    assert (s16_decode(res, P, d, d).double() - want).abs().max().item() < 5e-6        # split rows: 22 bits of |m| <= 5


def test_add_aux_epilogue_is_attention_readout(gpu):
    """out = m + gamma * (A @ V) with A in split rows and V as transposed planes: the per-iteration GEMM."""
    from tests_support import s16_decode
    from vfml import hip
    g = torch.Generator().manual_seed(6)
    P, d = 264, 128
    A = torch.softmax(torch.randn(P, P, generator=g), dim=-1)
    V = torch.randn(P, d, generator=g)
    m = torch.randn(P, d, generator=g)
    ld = 288
    A16 = torch.zeros(P * ld, device=gpu)
    Apad = torch.zeros(P, ld)
    Apad[:, :P] = A
    hip.to_s16(Apad.cuda().reshape(-1), P, ld, ld, A16, ld)
    m16 = torch.empty(P * d, device=gpu)
    hip.to_s16(m.cuda().reshape(-1), P, d, d, m16, d)
    vt = hip.SplitWeight(d, P, gpu).fill_transposed(V.cuda().reshape(-1), P, ld=d, scale=16.0)
    out = torch.empty(P * d, device=gpu)
    hip.conv2d(A16, P, ld, 1, 1, P, vt, None, d, 1, 1, out, d, out_scale=0.5, epilogue=hip.EPI_ADD_AUX,
               aux0=m16, ld_aux0=d, in_fmt=hip.FMT_S16, out_fmt=hip.FMT_S16, aux_fmt=hip.FMT_S16)
    got = s16_decode(out, P, d, d)
    ref = (m.double() + 0.5 * (A.double() @ V.double())).float()
    assert ((got - ref).abs().max() / ref.abs().max()).item() < 5e-6


def test_memflow_host_path_matches_oracle(gpu, tmp_path, monkeypatch):
    """MemFlowInference.compute_optical_flow (window ending at the frame, 0..255 floats, range heuristic,
    pad to /8, LAST TWO frames, unpad, CPU numpy) vs the oracle's restatement of the same script."""
    import contextlib
    import io
    import os
    import numpy as np
    from oracle import memflow_oracle as mm
    from processing.memflow_inference import MemFlowInference
    from vfml.memflow_net import memflow_cfg, seeded_memflow_state_dict
    from vfml.synth import synthetic_clip
    os.makedirs(tmp_path / "MemFlow_ckpt")
    sd = seeded_memflow_state_dict(memflow_cfg(), 0)
    torch.save(sd, tmp_path / "MemFlow_ckpt" / "MemFlowNet_sintel.pth")
    monkeypatch.chdir(tmp_path)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = MemFlowInference("cuda", sequence_length=3)
        eng.load_model()
    frames = synthetic_clip(4, 132, 200)                      # not multiples of 8: the padder is live
    got = eng.compute_optical_flow(frames, 2)
    assert got.shape == (132, 200, 2) and got.dtype == np.float32
    ora = mm.build_network(mm.get_cfg()).eval()
    ora.load_state_dict(sd)
    x = torch.from_numpy(np.stack(frames[0:3])).permute(0, 3, 1, 2).float()[None]
    ref = mm.compute_flow(ora, x).permute(1, 2, 0).numpy()
    epe = np.sqrt(((got - ref) ** 2).sum(-1))
    assert epe.mean() < EPE_TOL, epe.mean()
    assert np.array_equal(eng.compute_optical_flow_tiled(frames, 2), got)      # MemFlow never tiles


def test_memflow_resident_loop_reuses_encoder_outputs_exactly(gpu, tmp_path, monkeypatch):
    """The resident job loop (uint8 clip in HBM, frame ids as cache keys: the "current" frame of one field is
    the "previous" frame of the next and is encoded once) gives bit-identical fields to the host-array call,
    which encodes both frames of every pair; padded size, so the pad is part of the cached input."""
    import contextlib
    import io
    import os
    import numpy as np
    from processing.memflow_inference import MemFlowInference
    from vfml.memflow_net import memflow_cfg, seeded_memflow_state_dict
    from vfml.synth import synthetic_clip
    os.makedirs(tmp_path / "MemFlow_ckpt")
    torch.save(seeded_memflow_state_dict(memflow_cfg(), 0), tmp_path / "MemFlow_ckpt" / "MemFlowNet_sintel.pth")
    monkeypatch.chdir(tmp_path)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = MemFlowInference("cuda", sequence_length=3)
        eng.load_model()
    proc = eng.get_processor()
    frames = synthetic_clip(6, 132, 200)
    clip = proc.upload_clip(frames)
    net = proc.core_engine.model
    for i in range(6):
        a = proc.compute_optical_flow_resident(clip, i).cpu().numpy()
        assert np.array_equal(a, eng.compute_optical_flow(frames, i)), i
    assert sum(1 for k in net._feat_cache if k[0] == "f") >= 2          # the cache is in use


def test_memflow_fields_batched_equal_fields_one_by_one(gpu, tmp_path, monkeypatch):
    """Job loops pass three consecutive pairs per call of the engine (forward_pairs): every pair keeps its own
    attention operator and fresh memory, the fields are the same bits as one call per pair - padded size,
    frame 0 (the pair (0, 0)) on the single path."""
    import contextlib
    import io
    import os
    import numpy as np
    from processing.memflow_inference import MemFlowInference
    from vfml.memflow_net import memflow_cfg, seeded_memflow_state_dict
    from vfml.runner import run_sharded
    from vfml.synth import synthetic_clip
    os.makedirs(tmp_path / "MemFlow_ckpt")
    torch.save(seeded_memflow_state_dict(memflow_cfg(), 0), tmp_path / "MemFlow_ckpt" / "MemFlowNet_sintel.pth")
    monkeypatch.chdir(tmp_path)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = MemFlowInference("cuda", sequence_length=3)
        eng.load_model()
    proc = eng.get_processor()
    frames = synthetic_clip(8, 132, 200)
    clip = proc.upload_clip(frames)
    batched = [f.clone() for f in proc.compute_optical_flow_resident_batch(clip, list(range(8)))]
    proc.core_engine.model.clear_feature_cache()
    for i in range(8):
        assert torch.equal(batched[i], proc.compute_optical_flow_resident(clip, i)), i
    job = run_sharded(proc, clip, range(8))
    assert np.array_equal(job[6], batched[6].cpu().numpy()) and np.array_equal(job[0], batched[0].cpu().numpy())
