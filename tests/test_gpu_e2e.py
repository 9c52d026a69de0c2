"""End-to-end parity of the HIP engine against the CPU oracle: same seeded weights, same inputs,
mean end-point error < 1e-3 px (the tolerance BASELINE.json's north_star states)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

EPE_TOL = 1e-3  # px, mean over pixels of ||flow_engine - flow_oracle||_2


def _pair(seed=0, **over):
    from oracle import mof_oracle as mo
    from vfml import build_network, get_cfg
    from vfml.weights import seeded_state_dict
    cfg, ocfg = get_cfg(), mo.get_cfg()
    for k, v in over.items():
        setattr(cfg, k, v)
        setattr(ocfg, k, v)
    sd = seeded_state_dict(cfg, seed)
    net = build_network(cfg)
    net.load_state_dict(sd)
    net.cuda().eval()
    ora = mo.build_network(ocfg)
    ora.load_state_dict(sd)
    ora.eval()
    return net, ora


def _epe(a, b):
    return (a - b).pow(2).sum(2).sqrt()


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
@pytest.mark.parametrize("T,H,W", [(3, 128, 128), (5, 128, 192), (4, 136, 160)])
def test_model_forward_matches_oracle(gpu, T, H, W, precision):
    net, ora = _pair(precision=precision)
    g = torch.Generator().manual_seed(T * 1000 + H)
    x = torch.rand(1, T, 3, H, W, generator=g)
    ref, _ = ora(x, {})
    got, _ = net(x.cuda(), {})
    got = got.cpu()
    assert got.shape == ref.shape == (1, 2 * (T - 2), 2, H, W)
    e = _epe(got, ref)
    assert torch.isfinite(got).all()
    print(f"[{precision}] T={T} {H}x{W}: mean EPE {e.mean().item():.3e} px, max {e.max().item():.3e} px")
    assert e.mean().item() < EPE_TOL, f"mean EPE {e.mean().item():.3e} px (max {e.max().item():.3e})"


def test_fast_mode_config_matches_oracle(gpu):
    """--fast overrides (reference processing/videoflow_core.py:91-94): depth 6, 3 levels, radius 3."""
    net, ora = _pair(decoder_depth=6, corr_levels=3, corr_radius=3)
    x = torch.rand(1, 3, 3, 128, 160, generator=torch.Generator().manual_seed(5))
    ref, _ = ora(x, {})
    got, _ = net(x.cuda(), {})
    assert _epe(got.cpu(), ref).mean().item() < EPE_TOL


def test_bof_tri_frame_variant_matches_oracle(gpu):
    """--vf-architecture bof (BASELINE config 5 shape: seq_len 9): the tri-frame network on the centre
    triple of the window; output [1,2,2,H,W], the reference's index pick lands on the backward flow."""
    net, ora = _pair(network="BOFNet")
    x = torch.rand(1, 9, 3, 128, 160, generator=torch.Generator().manual_seed(7))
    ref, _ = ora(x, {})
    got, _ = net(x.cuda(), {})
    assert got.shape == ref.shape == (1, 2, 2, 128, 160)
    assert _epe(got.cpu(), ref).mean().item() < EPE_TOL
    centre, _ = net(x[:, 3:6].cuda(), {})
    assert torch.equal(centre, got)


def test_uint8_frames_equal_float_frames(gpu):
    """Handing the engine u8 HWC frames (device-side /255) gives the same field as the reference's
    host-side float conversion (processing/videoflow_processor.py:152-157)."""
    net, _ = _pair()
    u8 = torch.randint(0, 256, (3, 128, 128, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(6))
    f = (u8.float() / 255.0).permute(0, 3, 1, 2)[None]
    a, _ = net(f.cuda(), {})
    a = a.clone()
    b, _ = net.forward_u8(u8.cuda())
    assert torch.equal(a, b)


def test_sliding_window_feature_cache_is_exact(gpu):
    """Per-frame encoder outputs reused across overlapping windows (frame_keys) give bit-identical
    fields to encoding every window from scratch; so does the resident-clip processor path."""
    import contextlib
    import io
    import numpy as np
    from processing.videoflow_processor import VideoFlowProcessor
    from vfml.synth import synthetic_clip
    net, _ = _pair()
    frames = synthetic_clip(8, 128, 160)
    clip = torch.from_numpy(np.stack(frames)).cuda()
    with contextlib.redirect_stdout(io.StringIO()):
        proc = VideoFlowProcessor("cuda", sequence_length=5)
    proc.core.model = net
    for i in range(8):
        a = proc.compute_optical_flow_resident(clip, i).clone()           # cached encoders
        net.clear_feature_cache()
        idx = proc.window_indices(8, i)
        b, _ = net.forward_u8(clip[idx])                                   # everything recomputed
        assert torch.equal(a, b[0, 3].permute(1, 2, 0)), i
        c = proc.compute_optical_flow(frames, i)                           # the reference API path (host floats)
        assert np.array_equal(a.cpu().numpy(), c), i
    # tiles: a different crop of the same frame is a different cache entry
    tile = {'x': 32, 'y': 0, 'width': 128, 'height': 128}
    t1 = proc.compute_optical_flow_resident(clip, 3, tile=tile).clone()
    t2, _ = net.forward_u8(clip[proc.window_indices(8, 3)][:, 0:128, 32:160].contiguous())
    assert torch.equal(t1, t2[0, 3].permute(1, 2, 0))


def test_context_store_ring_wraps_and_falls_back(gpu, monkeypatch):
    """The gate convolutions' per-frame context parts live in a ring of frame slots that the launches reach through a device
    cell (vfml_conv_desc.addend_ind): a sliding job longer than the ring (it wraps through the mirrored first slots), the same
    frames in random order (centres not in arrival order: gathered) and VFML_CTX_GATHER=1 (every window gathered) all give
    the bits of windows computed from scratch."""
    import contextlib
    import io
    import numpy as np
    from processing.videoflow_processor import VideoFlowProcessor
    from vfml.synth import synthetic_clip
    net, _ = _pair()
    nfr = 2 * net.CTX_RING + 3
    frames = synthetic_clip(nfr, 128, 160)
    clip = torch.from_numpy(np.stack(frames)).cuda()
    with contextlib.redirect_stdout(io.StringIO()):
        proc = VideoFlowProcessor("cuda", sequence_length=5)
    proc.core.model = net
    ref = []
    for i in range(nfr):                                   # from scratch (no keys: nothing cached)
        b, _ = net.forward_u8(clip[proc.window_indices(nfr, i)])
        ref.append(b[0, 3].permute(1, 2, 0).clone())
    net.clear_feature_cache()
    wrapped = 0
    for i in range(nfr):                                   # the sliding job
        a = proc.compute_optical_flow_resident(clip, i)
        assert torch.equal(a, ref[i]), i
        live = sorted(net._ctx_store["live"])
        wrapped += live != list(range(live[0], live[0] + len(live)))
    assert wrapped >= 2                                    # (windows whose centres straddle the end of the ring)
    order = np.random.default_rng(0).permutation(nfr)
    for i in order[:12]:                                   # random access over the same (partly cached) frames
        assert torch.equal(proc.compute_optical_flow_resident(clip, int(i)), ref[int(i)]), int(i)
    monkeypatch.setenv("VFML_CTX_GATHER", "1")
    net.clear_feature_cache()
    for i in range(6):
        assert torch.equal(proc.compute_optical_flow_resident(clip, i), ref[i]), i


def test_engine_refuses_cpu_tensors():
    from vfml import build_network, get_cfg
    net = build_network(get_cfg())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 3, 3, 128, 128), {})


@pytest.mark.parametrize("model", ["videoflow", "memflow"])
def test_cli_on_gpu_writes_the_cache_the_api_would(gpu, tmp_path, monkeypatch, model):
    """flow_processor.py end to end on the GPU (resident clip, sharded runner with world=1): every cached
    field equals the field the reference API path computes for that frame."""
    import contextlib
    import io
    import os
    import numpy as np
    import flow_processor
    from processing.flow_inference import VideoFlowInference
    from processing.memflow_inference import MemFlowInference
    from vfml import get_cfg
    from vfml.memflow_net import memflow_cfg, seeded_memflow_state_dict
    from vfml.synth import synthetic_clip
    from vfml.weights import write_seeded_checkpoint
    write_seeded_checkpoint(str(tmp_path), get_cfg(), seed=0)
    os.makedirs(tmp_path / "MemFlow_ckpt")
    torch.save(seeded_memflow_state_dict(memflow_cfg(), 0), tmp_path / "MemFlow_ckpt" / "MemFlowNet_sintel.pth")
    monkeypatch.chdir(tmp_path)
    argv = ["--input", "synthetic:160x128x5", "--output", str(tmp_path / "out"), "--device", "cuda", "--model", model,
            "--sequence-length", "3", "--interactive", "--skip-lods"]
    with contextlib.redirect_stdout(io.StringIO()):
        assert flow_processor.main(argv) == 0
        eng = (MemFlowInference("cuda", sequence_length=3) if model == "memflow"
               else VideoFlowInference("cuda", sequence_length=3))
        eng.load_model()
    tag = "memflow_sintel" if model == "memflow" else "videoflow_mof_sintel_standard"
    cache = tmp_path / "out" / f"synthetic_160x128x5_flow_cache_{tag}_seq3_start0_frames5"
    frames = synthetic_clip(5, 128, 160)
    for i in (0, 2, 4):
        got = np.load(cache / f"flow_frame_{i:06d}.npz")["flow"]
        assert np.array_equal(got, eng.compute_optical_flow(frames, i)), i


def test_longest_window_and_reference_api_with_padding(gpu, tmp_path, monkeypatch):
    """seq_len 10 (the reference's maximum, processing/videoflow_processor.py:357-361: 8 centre frames =
    the lookup kernel's map limit) on a frame whose sides are not multiples of 8, through the reference
    API path (host floats, InputPadder, index pick) with --fast."""
    import contextlib
    import io
    import numpy as np
    from oracle import mof_oracle as mo
    from processing.flow_inference import VideoFlowInference
    from vfml import get_cfg
    from vfml.synth import synthetic_clip
    from vfml.weights import seeded_state_dict, write_seeded_checkpoint
    write_seeded_checkpoint(str(tmp_path), get_cfg(), seed=0)
    monkeypatch.chdir(tmp_path)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = VideoFlowInference("cuda", fast_mode=True, sequence_length=10)
        eng.load_model()
    assert eng.get_model_info()["config"] == {"decoder_depth": 6, "corr_levels": 3, "corr_radius": 3}
    frames = synthetic_clip(10, 100, 132)
    got = eng.compute_optical_flow(frames, 5)
    assert got.shape == (100, 132, 2)
    ocfg = mo.get_cfg()
    ocfg.decoder_depth, ocfg.corr_levels, ocfg.corr_radius = 6, 3, 3
    ora = mo.build_network(ocfg).eval()
    ora.load_state_dict(seeded_state_dict(get_cfg(), 0))
    x = eng.prepare_frame_sequence(frames, 5).cpu()
    pad = mo.InputPadder(x.shape[-2:])
    ref = pad.unpad(ora(pad.pad(x), {})[0])[0, 8].permute(1, 2, 0).numpy()      # shape[1]//2 of 16 flows
    assert np.sqrt(((got - ref) ** 2).sum(-1)).mean() < EPE_TOL


def test_host_array_api_fast_path_is_exact_and_sees_edits(gpu, tmp_path, monkeypatch):
    """The reference's call shape (list of uint8 numpy frames in, numpy field out) keyed on frame CONTENT:
    bit-identical to the float32 path it replaces, along a sliding loop, for tiles, and after a frame is
    edited in place (its hash, hence its cache entries, change)."""
    import contextlib
    import io
    import numpy as np
    from processing.flow_inference import VideoFlowInference
    from vfml import get_cfg
    from vfml.synth import synthetic_clip
    from vfml.weights import write_seeded_checkpoint
    write_seeded_checkpoint(str(tmp_path), get_cfg(), seed=0)
    monkeypatch.chdir(tmp_path)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = VideoFlowInference("cuda", sequence_length=5)
        eng.load_model()
    proc = eng.get_processor()
    frames = synthetic_clip(7, 128, 160)

    def float_path(fr, i):       # what compute_optical_flow did before the fast path (and still does for float frames)
        return proc.core.compute_flow_from_tensor(proc.prepare_frame_sequence(fr, i)).permute(1, 2, 0).cpu().numpy()

    for i in (0, 1, 3, 4, 6):
        assert proc._flow_from_host_u8(frames, i) is not None
        assert np.array_equal(eng.compute_optical_flow(frames, i), float_path(frames, i))
    frames[4][10:40, 20:60] = 255 - frames[4][10:40, 20:60]          # in-place edit of a cached frame
    assert np.array_equal(eng.compute_optical_flow(frames, 4), float_path(frames, 4))
    assert np.array_equal(eng.compute_optical_flow(frames, 3), float_path(frames, 3))
    # float frames and sizes that need padding take the reference's own path
    assert proc._flow_from_host_u8([f.astype(np.float32) for f in frames], 3) is None
    assert proc._flow_from_host_u8(synthetic_clip(5, 132, 164), 2) is None


def test_full_size_properties_1080p_and_4k_tiles(gpu):
    """BASELINE's full sizes, through properties that need no oracle run: at 1080p (config 2) the sliding
    caches give bit-identical fields to encoding the window from scratch, a second run is bit-identical
    (no atomics, no order dependence), and the host-array API equals the resident path; at 4K --tile
    (config 3) the runner's pasted frame equals the six tiles computed one by one, ragged edge tiles included.
    (The oracle comparison at 1080p itself is part of every bench.py run: `cpu_baseline.epe_mean_px`.)"""
    import contextlib
    import io
    import numpy as np
    from processing.videoflow_processor import VideoFlowProcessor
    from vfml.runner import run_sharded
    from vfml.synth import synthetic_clip
    net, _ = _pair()
    with contextlib.redirect_stdout(io.StringIO()):
        proc = VideoFlowProcessor("cuda", sequence_length=5)
    proc.core.model = net
    frames = synthetic_clip(7, 1080, 1920)
    clip = proc.upload_clip(frames)
    a = [proc.compute_optical_flow_resident(clip, i).clone() for i in (2, 3, 4)]      # sliding, cached
    net.clear_feature_cache()
    b, _ = net.forward_u8(clip[proc.window_indices(7, 4)])                             # from scratch
    assert torch.equal(a[2], b[0, 3].permute(1, 2, 0))
    assert torch.isfinite(a[2]).all() and float(a[2].abs().max()) < 2000.0
    again = proc.compute_optical_flow_resident(clip, 4)
    assert torch.equal(a[2], again)
    assert np.array_equal(proc.compute_optical_flow(frames, 3), a[1].cpu().numpy())   # host-array API
    del a, b, again
    net.clear_feature_cache()
    torch.cuda.empty_cache()

    with contextlib.redirect_stdout(io.StringIO()):
        tproc = VideoFlowProcessor("cuda", tile_mode=True, sequence_length=5)
    tproc.core.model = net
    big = synthetic_clip(5, 2160, 3840)
    bclip = tproc.upload_clip(big)
    pasted = run_sharded(tproc, bclip, [2], tile_mode=True)[0]
    tiles = tproc.calculate_tile_grid(3840, 2160)[4]
    assert [(t['width'], t['height']) for t in tiles] == [(1280, 1280)] * 3 + [(1280, 880)] * 3
    for t in tiles:
        net.clear_feature_cache()
        crop = bclip[tproc.window_indices(5, 2)][:, t['y']:t['y'] + t['height'], t['x']:t['x'] + t['width']].contiguous()
        one, _ = net.forward_u8(crop)
        ref = one[0, 3].permute(1, 2, 0).cpu().numpy()
        assert np.array_equal(pasted[t['y']:t['y'] + t['height'], t['x']:t['x'] + t['width']], ref)
    net.clear_feature_cache()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("precision", ["f16x3", "mixed"])
def test_prefetched_encoders_give_the_same_fields(gpu, monkeypatch, precision):
    """(both arithmetic plans: the mixed plan - what bench.py and the CLI run by default - dispatches other kernel
    instantiations, the one-MFMA 64-channel-step and "2a" forms, than the fp32-grade plan.)
    The encoders of the NEXT window run on a side stream beside this field's update iterations (network.py
    prefetch_frames): the same kernels on the same inputs, so every field of a streamed 1080p clip is bit-identical to the
    one computed without the prefetch - also the field whose iterations ran beside it.  (Round 2 found the fixed-radius
    lookup returning different samples with an MFMA kernel on a second stream: profiles/r02_kernel_anatomy.md section 7.)"""
    import contextlib
    import io
    from processing.videoflow_processor import VideoFlowProcessor
    from vfml.synth import synthetic_clip
    net, _ = _pair(precision=precision)
    with contextlib.redirect_stdout(io.StringIO()):
        proc = VideoFlowProcessor("cuda", sequence_length=5)
    proc.core.model = net
    frames = synthetic_clip(10, 1080, 1920)
    clip = proc.upload_clip(frames)
    order = [2, 3, 4, 5, 6, 7]
    monkeypatch.setenv("VFML_PREFETCH", "0")
    net.clear_feature_cache()
    serial = [proc.compute_optical_flow_resident(clip, i).clone() for i in order]
    for rep in range(3):
        monkeypatch.setenv("VFML_PREFETCH", "1")
        net.clear_feature_cache()
        pre = [proc.compute_optical_flow_resident(clip, i).clone() for i in order]
        torch.cuda.synchronize()
        for i, a, b in zip(order, serial, pre):
            assert torch.equal(a, b), f"field {i} (pass {rep}): {int((a != b).sum())} values differ, max {float((a - b).abs().max()):.3g} px"
    net.clear_feature_cache()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("graph", ["1", "0"])
def test_flow_half_of_the_motion_encoder_on_a_second_stream(gpu, monkeypatch, graph):
    """VFML_FLOW_BRANCH=1: the flow half of the motion encoder (flow -> convf1 -> convf2) runs on a second stream beside the
    lookups of the same iteration, joined before the convolution that reads both halves - eagerly and as two branches of the
    captured graph.  Same kernels, same inputs: 1080p fields bit-identical to the one-stream engine."""
    import contextlib
    import io
    from processing.videoflow_processor import VideoFlowProcessor
    from vfml.synth import synthetic_clip
    net, _ = _pair()
    with contextlib.redirect_stdout(io.StringIO()):
        proc = VideoFlowProcessor("cuda", sequence_length=5)
    proc.core.model = net
    clip = proc.upload_clip(synthetic_clip(9, 1080, 1920))
    order = [2, 3, 4, 5, 6]
    monkeypatch.setenv("VFML_GRAPH", graph)
    monkeypatch.setenv("VFML_FLOW_BRANCH", "0")
    net.clear_feature_cache()
    one = [proc.compute_optical_flow_resident(clip, i).clone() for i in order]
    monkeypatch.setenv("VFML_FLOW_BRANCH", "1")
    for rep in range(2):
        net.clear_feature_cache()
        two = [proc.compute_optical_flow_resident(clip, i).clone() for i in order]
        torch.cuda.synchronize()
        for i, a, b in zip(order, one, two):
            assert torch.equal(a, b), f"field {i} (pass {rep}): {int((a != b).sum())} values differ, max {float((a - b).abs().max()):.3g} px"
    net.clear_feature_cache()
    torch.cuda.empty_cache()


def test_bof_fields_batched_equal_fields_one_by_one(gpu):
    """`--vf-architecture bof`, seq 9: consecutive interior fields go through the tri-frame network four at a
    time (tri_batch); clip-edge fields, whose centre triples repeat frames, one by one.  Both orders of
    evaluation give the same bits, and the runner's job equals the per-frame API."""
    import contextlib
    import io
    import numpy as np
    from processing.videoflow_processor import VideoFlowProcessor
    from vfml.runner import run_sharded
    from vfml.synth import synthetic_clip
    net, _ = _pair(network="BOFNet")
    with contextlib.redirect_stdout(io.StringIO()):
        proc = VideoFlowProcessor("cuda", sequence_length=9, architecture="bof", dataset="things")
    proc.core.model = net
    frames = synthetic_clip(12, 128, 160)
    clip = proc.upload_clip(frames)
    batched = [f.clone() for f in proc.compute_optical_flow_resident_batch(clip, list(range(12)))]
    net.clear_feature_cache()
    for i in range(12):
        one = proc.compute_optical_flow_resident(clip, i)
        assert torch.equal(batched[i], one), i
    net.clear_feature_cache()
    job = run_sharded(proc, clip, range(12))
    assert np.array_equal(job[5], batched[5].cpu().numpy()) and np.array_equal(job[0], batched[0].cpu().numpy())
    assert np.array_equal(job[7], proc.compute_optical_flow(frames, 7))


def test_two_clips_of_one_shape_through_one_engine(gpu):
    """A clip's frames are cached under the identity of the UPLOAD, not of its device address: clip A is freed
    before clip B (same shape, different pixels) is uploaded - torch's caching allocator hands B the address A had
    and a fresh tensor starts at version 0 again - and B's fields still equal a from-scratch forward of B
    (with (data_ptr, _version) keys they were A's encoder maps and pyramids).  MOF and MemFlow processors."""
    import contextlib
    import io
    import numpy as np
    from processing.memflow_processor import MemFlowProcessor
    from processing.videoflow_processor import VideoFlowProcessor
    from vfml.memflow_net import build_memflow_network, memflow_cfg, seeded_memflow_state_dict
    from vfml.synth import synthetic_clip
    net, _ = _pair()
    with contextlib.redirect_stdout(io.StringIO()):
        proc = VideoFlowProcessor("cuda", sequence_length=5)
    proc.core.model = net
    fa = synthetic_clip(7, 128, 160)
    fb = [np.ascontiguousarray(255 - f[::-1]) for f in synthetic_clip(7, 128, 160)]     # same shape, other pixels
    clip = proc.upload_clip(fa)
    addr = clip.data_ptr()
    got_a = [proc.compute_optical_flow_resident(clip, i).clone() for i in (2, 3, 4)]
    del clip
    clip = proc.upload_clip(fb)
    same_addr = clip.data_ptr() == addr
    got_b = [proc.compute_optical_flow_resident(clip, i).clone() for i in (2, 3, 4)]
    tri = [f.clone() for f in proc.compute_optical_flow_resident_batch(clip, [3])]
    net.clear_feature_cache()
    for k, i in enumerate((2, 3, 4)):
        ref, _ = net.forward_u8(clip[proc.window_indices(7, i)])
        ref = ref[0, 3].permute(1, 2, 0)
        assert torch.equal(got_b[k], ref), (i, same_addr)
        assert not torch.equal(got_a[k], ref)
    assert torch.equal(tri[0], got_b[1])
    print(f"second upload reused the first clip's address: {same_addr}")

    cfg = memflow_cfg()
    mnet = build_memflow_network(cfg)
    mnet.load_state_dict(seeded_memflow_state_dict(cfg, 0))
    mnet.cuda().eval()
    with contextlib.redirect_stdout(io.StringIO()):
        mproc = MemFlowProcessor("cuda", sequence_length=3)
    mproc.core_engine.model = mnet
    clip = mproc.upload_clip(fa)
    ma = [f.clone() for f in mproc.compute_optical_flow_resident_batch(clip, [2, 3, 4])]
    del clip
    clip = mproc.upload_clip(fb)
    mb = [f.clone() for f in mproc.compute_optical_flow_resident_batch(clip, [2, 3, 4])]
    mnet.clear_feature_cache()
    for k, i in enumerate((2, 3, 4)):
        ref = mproc.compute_optical_flow_resident(clip, i)
        assert torch.equal(mb[k], ref), i
        assert not torch.equal(ma[k], ref)


def test_bof_720p_seq9_fp16_config(gpu):
    """BASELINE config 5 at its full size: BOF_things, seq_len 9, 1280x720, fp16.  The tri-frame network on the
    centre triple of the 9-frame window, (a) in the fp32-grade arithmetic (f16x3) within the 1e-3 px tolerance of the
    CPU oracle, (b) in the config's fp16-GRADE plan (vfml/cfg.py BOF_F16_PLAN: one MFMA per product except the per-frame
    encoders / context parts at three and the linear flow path at "2a") INSIDE that tolerance with margin - < 5e-4 px, about
    twice what it measures (1.5e-4 .. 2.1e-4) - on three weight seeds (seed 0 against the CPU oracle, seeds 1-2 against the f16x3
    engine, itself ~6e-6 px from the oracle), (c) in 'f16' (plain f16 operands everywhere): 2.1e-3 px, OUTSIDE the
    contract by design - reported and bounded at twice its measured value so that a broken kernel, not a rounding, fails,
    (d) the job loop's batched evaluation (eight fields per pass, tri_batch) bit-identical to one field per call at this
    size, in the f16 arithmetic."""
    import contextlib
    import io
    import numpy as np
    from oracle import mof_oracle as mo
    from processing.videoflow_processor import VideoFlowProcessor
    from vfml import build_network, get_cfg
    from vfml.cfg import BOF_F16_PLAN
    from vfml.synth import synthetic_clip
    from vfml.weights import seeded_state_dict
    H, W, T = 720, 1280, 9
    frames = synthetic_clip(14, H, W)
    cfg = get_cfg()
    cfg.network = "BOFNet"
    sd = seeded_state_dict(cfg, 0)
    ocfg = mo.get_cfg()
    ocfg.network = "BOFNet"
    ora = mo.build_network(ocfg).eval()
    ora.load_state_dict(sd)
    i = 6                                                   # a field with a full window: frames 2..10, triple 5,6,7
    win = np.stack(frames[i - T // 2:i + T // 2 + 1])
    x = torch.from_numpy(win.astype(np.float32) / 255.0).permute(0, 3, 1, 2)[None]
    ref, _ = ora(x, {})
    ref = ref[0, ref.shape[1] // 2].permute(1, 2, 0)       # the reference's pick: the backward flow
    got = {}
    fields = {}
    for prec in ("f16x3", "mixed", "f16"):
        c = get_cfg()
        c.network, c.precision = "BOFNet", prec
        if prec == "mixed":
            c.mfma_plan = dict(BOF_F16_PLAN)
        net = build_network(c)
        net.load_state_dict(sd)
        net.cuda().eval()
        with contextlib.redirect_stdout(io.StringIO()):
            proc = VideoFlowProcessor("cuda", sequence_length=T, architecture="bof", dataset="things")
        proc.core.model = net
        clip = proc.upload_clip(frames)
        f = proc.compute_optical_flow_resident(clip, i).clone()
        fields[prec] = f
        e = (f.cpu() - ref).pow(2).sum(-1).sqrt()
        got[prec] = (float(e.mean()), float(e.max()))
        print(f"BOF 720p seq9 [{prec}]: mean EPE {got[prec][0]:.3e} px, max {got[prec][1]:.3e} px "
              f"(|flow| mean {float(ref.abs().mean()):.3f} px)")
        assert torch.isfinite(f).all()
        if prec == "f16":
            batched = [t.clone() for t in proc.compute_optical_flow_resident_batch(clip, list(range(14)))]
            net.clear_feature_cache()
            for k in (0, 3, 4, 6, 9, 10, 13):              # clip edges (repeated frames, one by one) and interior (batched)
                assert torch.equal(batched[k], proc.compute_optical_flow_resident(clip, k)), k
            assert torch.equal(batched[i], f)
        del net, proc, clip
        torch.cuda.empty_cache()
    assert got["f16x3"][0] < EPE_TOL, got
    assert got["mixed"][0] < 5e-4, got                      # BOF_F16_PLAN: inside the 1e-3 contract with margin
    # plain f16 operands through ~100 dependent layers and 12 iterations: outside the fp32 tolerance by design (2.1e-3 px
    # measured); bounded at about twice that so that a broken kernel (not a rounding) fails
    assert got["f16"][0] < 4.5e-3, got
    # the plan on two more weight seeds, against the fp32-grade engine
    for seed in (1, 2):
        sd2 = seeded_state_dict(cfg, seed)
        out = {}
        for prec in ("f16x3", "mixed"):
            c = get_cfg()
            c.network, c.precision = "BOFNet", prec
            if prec == "mixed":
                c.mfma_plan = dict(BOF_F16_PLAN)
            net = build_network(c)
            net.load_state_dict(sd2)
            net.cuda().eval()
            win_u8 = torch.from_numpy(win).cuda()
            o = net.forward_u8(win_u8, return_lowres=False)[0]
            out[prec] = o[0, o.shape[1] // 2].permute(1, 2, 0).cpu()
            net.release_workspace()
            del net
            torch.cuda.empty_cache()
        e = (out["mixed"] - out["f16x3"]).pow(2).sum(-1).sqrt()
        print(f"BOF 720p seq9 [BOF_F16_PLAN] seed {seed}: mean EPE {float(e.mean()):.3e} px vs the f16x3 engine, max {float(e.max()):.3e}")
        assert float(e.mean()) < 5e-4, (seed, float(e.mean()))


def test_graph_replay_of_the_iteration_body_is_bit_identical(gpu):
    """The update iterations of a field run as one replayed HIP graph from the third field of a configuration on
    (eager, capture + replay, replay ...): every field equals the eager engine's, along a sliding job, across a change
    of geometry and back (a workspace reallocation drops the graphs), and after new weights are loaded."""
    import contextlib
    import io
    import numpy as np
    from processing.videoflow_processor import VideoFlowProcessor
    from vfml import build_network, get_cfg
    from vfml.synth import synthetic_clip
    from vfml.weights import seeded_state_dict
    nets = {}
    for use in (True, False):
        cfg = get_cfg()
        cfg.use_graph = use
        net = build_network(cfg)
        net.load_state_dict(seeded_state_dict(cfg, 0))
        nets[use] = net.cuda().eval()
    procs = {}
    for use, net in nets.items():
        with contextlib.redirect_stdout(io.StringIO()):
            procs[use] = VideoFlowProcessor("cuda", sequence_length=5)
        procs[use].core.model = net
    frames = synthetic_clip(9, 128, 160)
    small = [np.ascontiguousarray(f[:128, :128]) for f in frames]
    clips = {use: (p.upload_clip(frames), p.upload_clip(small)) for use, p in procs.items()}
    replays = 0
    for which, idxs in ((0, range(9)), (1, range(2, 6)), (0, range(3, 7))):
        for i in idxs:
            a = procs[True].compute_optical_flow_resident(clips[True][which], i)
            b = procs[False].compute_optical_flow_resident(clips[False][which], i)
            assert torch.equal(a, b), (which, i)
        replays += sum(1 for g in nets[True]._graphs.values() if not isinstance(g, str))
    assert replays >= 3 and not nets[False]._graphs          # graphs were captured and used on one side only
    # the full-output call (another launch sequence) and new weights
    for seed in (0, 1):
        for use, net in nets.items():
            net.load_state_dict(seeded_state_dict(get_cfg(), seed))
        outs = {use: [net.forward_u8(clips[use][0][2:7], return_lowres=False)[0].clone() for _ in range(3)]
                for use, net in nets.items()}
        assert all(torch.equal(outs[True][k], outs[False][0]) for k in range(3)), seed


MIXED_TOL = 1e-4    # px: the mixed plan's budget at 1080p, a tenth of north_star's tolerance


def test_mixed_plan_stays_within_its_budget_at_1080p(gpu):
    """cfg.precision='mixed' with the shipped plan (vfml/cfg.py DEFAULT_MIXED_PLAN + DEFAULT_MIXED_CORR_VOLUME: the coarsest
    pyramid level as f16 - what VideoFlowCore runs by default): mean EPE <= 1e-4 px at
    1920x1080 on three weight seeds and for T in {3, 5}.  Reference: the CPU oracle for seed 0 (both T); for seeds 1
    and 2 the engine's own exact-f32 arithmetic (cfg.precision='f32', v_mfma_f32_32x32x2_f32 - itself within 3e-6 px
    of the oracle: test_model_forward_matches_oracle and bench.py's cpu_baseline) - a full-size oracle field costs
    40 s of CPU time."""
    import numpy as np
    from oracle import mof_oracle as mo
    from vfml import build_network, get_cfg
    from vfml.cfg import DEFAULT_MIXED_CORR_VOLUME, DEFAULT_MIXED_PLAN
    from vfml.synth import synthetic_clip
    from vfml.weights import seeded_state_dict
    H, W = 1080, 1920
    frames = synthetic_clip(5, H, W)
    clip = torch.from_numpy(np.stack(frames)).cuda()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    worst = 0.0
    for seed in (0, 1, 2):
        sd = seeded_state_dict(get_cfg(), seed)
        nets = {}
        for prec in ("mixed", "f32"):
            c = get_cfg()
            c.precision = prec
            if prec == "mixed":
                c.mfma_plan = dict(DEFAULT_MIXED_PLAN)
                c.corr_volume = DEFAULT_MIXED_CORR_VOLUME
            n = build_network(c)
            n.load_state_dict(sd)
            nets[prec] = n.cuda().eval()
        for T in (3, 5):
            win = clip[1:4] if T == 3 else clip
            got = nets["mixed"].forward_u8(win, return_lowres=False)[0]
            got = got[0, got.shape[1] // 2].permute(1, 2, 0).cpu()
            if seed == 0:
                ora = mo.build_network(mo.get_cfg()).eval()
                ora.load_state_dict(sd)
                x = (win.cpu().float() / 255.0).permute(0, 3, 1, 2)[None]
                ref = ora(x, {})[0]
                ref = ref[0, ref.shape[1] // 2].permute(1, 2, 0)
                tag = "CPU oracle"
            else:
                ref = nets["f32"].forward_u8(win, return_lowres=False)[0]
                ref = ref[0, ref.shape[1] // 2].permute(1, 2, 0).cpu()
                tag = "exact-f32 engine"
            e = (got - ref).pow(2).sum(-1).sqrt()
            worst = max(worst, float(e.mean()))
            print(f"mixed plan 1080p seed {seed} T={T}: mean EPE {float(e.mean()):.3e} px, max {float(e.max()):.3e} px vs {tag}")
            assert float(e.mean()) <= MIXED_TOL, (seed, T, float(e.mean()))
        for n in nets.values():
            n.release_workspace()
        del nets
        torch.cuda.empty_cache()
    print(f"mixed plan: worst mean EPE {worst:.3e} px (budget {MIXED_TOL:.0e})")


@pytest.mark.parametrize("T,H,W", [(5, 128, 256), (3, 256, 256)])
def test_f16_correlation_volume_stays_inside_the_contract(gpu, T, H, W):
    """cfg.corr_volume = 'f16' (opt-in): the pyramids hold one f16 per correlation value (half the bytes written and
    gathered).  Against the fp32 oracle the field stays inside the 1e-3 px contract; against the engine's own f32-volume
    field it differs (the switch does something), and a sliding job reproduces from-scratch fields bit for bit."""
    net, ora = _pair(precision="f16x3")
    g = torch.Generator().manual_seed(T + H)
    x = torch.rand(1, T, 3, H, W, generator=g)
    ref, _ = ora(x, {})
    base, _ = net(x.cuda(), {})
    net.cfg.corr_volume = "f16"
    net.clear_feature_cache()
    got, _ = net(x.cuda(), {})
    e = _epe(got.cpu(), ref)
    print(f"[corr_volume f16] T={T} {H}x{W}: mean EPE {e.mean().item():.3e} px, max {e.max().item():.3e} px "
          f"(f32 volume: {_epe(base.cpu(), ref).mean().item():.3e})")
    assert torch.isfinite(got).all()
    assert e.mean().item() < EPE_TOL
    assert not torch.equal(got, base)
    # the sliding path (pyramids cached per frame pair, a pair's volume and its transposed twin from one GEMM pass)
    # reproduces from-scratch fields bit for bit with f16 volumes too
    frames = (torch.rand(T + 1, H, W, 3, generator=g) * 255).to(torch.uint8).cuda()
    keys = [("k", i) for i in range(T + 1)]
    net.clear_feature_cache()
    net.forward_u8(frames[:T], return_lowres=False, frame_keys=keys[:T])
    slid, _ = net.forward_u8(frames[1:], return_lowres=False, frame_keys=keys[1:])
    net.clear_feature_cache()
    scratch, _ = net.forward_u8(frames[1:], return_lowres=False)
    assert torch.equal(slid, scratch)


def test_f16_correlation_volume_at_1080p(gpu):
    """The opt-in f16 volume at 1920x1080, T = 5, against the engine's exact-f32 arithmetic (itself within 3e-6 px of the
    oracle): with the fp32-grade plan and with the mixed plan.  Inside the 1e-3 px contract; NOT inside the mixed plan's own
    1e-4 budget - which is why it is not a default."""
    import numpy as np
    from vfml import build_network, get_cfg
    from vfml.cfg import DEFAULT_MIXED_PLAN
    from vfml.synth import synthetic_clip
    from vfml.weights import seeded_state_dict
    clip = torch.from_numpy(np.stack(synthetic_clip(5, 1080, 1920))).cuda()
    sd = seeded_state_dict(get_cfg(), 1)

    def field(prec, vol):
        c = get_cfg()
        c.precision, c.corr_volume = prec, vol
        if prec == "mixed":
            c.mfma_plan = dict(DEFAULT_MIXED_PLAN)
        n = build_network(c)
        n.load_state_dict(sd)
        n.cuda().eval()
        f = n.forward_u8(clip, return_lowres=False)[0]
        f = f[0, f.shape[1] // 2].permute(1, 2, 0).cpu()
        n.release_workspace()
        del n
        torch.cuda.empty_cache()
        return f

    ref = field("f32", "f32")
    for prec in ("f16x3", "mixed"):
        e = (field(prec, "f16") - ref).pow(2).sum(-1).sqrt()
        print(f"corr_volume f16, {prec}, 1080p seed 1 T=5: mean EPE {float(e.mean()):.3e} px, max {float(e.max()):.3e} px")
        assert float(e.mean()) < EPE_TOL
