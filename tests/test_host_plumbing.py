"""Host-side mirror vs golden fixtures cut from the reference's own modules
(tests/golden/make_fixtures.py).  Bit-exact: windows, tiles, names, file bytes."""
import io
import json
import os
import contextlib

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
J = json.load(open(os.path.join(HERE, "golden", "host_plumbing.json")))
A = np.load(os.path.join(HERE, "golden", "host_plumbing.npz"))


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def make_proc(**kw):
    from processing.videoflow_processor import VideoFlowProcessor
    with quiet():
        return VideoFlowProcessor("cpu", **kw)


class FakeModel(torch.nn.Module):
    """Same stand-in model the fixture script used: flow k = (k+1) * mean over frames of channels 0..1."""

    def forward(self, x, _):
        B, T, C, H, W = x.shape
        base = x[:, :, :2].mean(dim=1, keepdim=True)
        flows = torch.cat([base * (k + 1) for k in range(2 * (T - 2))], dim=1)
        return flows.view(B, 2 * (T - 2), 2, H, W), None


def test_frame_windows_match_reference():
    for w in J["windows"]:
        p = make_proc(sequence_length=w["T"])
        assert p.window_indices(w["n"], w["i"]) == w["idx"], w
        frames = [np.full((2, 2, 3), i, np.uint8) for i in range(w["n"])]
        t = p.prepare_frame_sequence(frames, w["i"])
        assert list(t.shape) == w["shape"] and str(t.dtype) == w["dtype"]
        assert (t[0, :, 0, 0, 0] * 255.0).round().long().tolist() == w["idx"]
        assert len(frames) == w["n"]  # caller's list is not mutated


def test_frame_tensor_values_bit_exact():
    p = make_proc(sequence_length=3)
    got = p.prepare_frame_sequence(list(A["seq_u8_in"]), 1).numpy()
    assert got.dtype == np.float32 and np.array_equal(got, A["seq_u8_out"])
    got = p.prepare_frame_sequence(list(A["seq_f32_in"]), 2).numpy()
    assert np.array_equal(got, A["seq_f32_out"])      # float frames pass through unscaled


def test_tile_grids_match_reference():
    p = make_proc()
    for g in J["tile_grids"]:
        kw = {"tile_size": g["tile_size"]} if "tile_size" in g else {}
        tw, th, cols, rows, info = p.calculate_tile_grid(g["w"], g["h"], **kw)
        assert [tw, th] == g["tile"] and cols == g["cols"] and rows == g["rows"]
        assert [[t["x"], t["y"], t["width"], t["height"], t["col"], t["row"]] for t in info] == g["tiles"]
        frame = np.arange(g["h"] * g["w"]).reshape(g["h"], g["w"])
        t = info[-1]
        assert p.extract_tile(frame, t).shape == (t["height"], t["width"])


def test_tiled_assembly_matches_reference():
    p = make_proc(tile_mode=True, sequence_length=5)
    p.core.model = FakeModel()
    grid = p.calculate_tile_grid
    p.calculate_tile_grid = lambda w, h, tile_size=1280: grid(w, h, 16)
    frames = list(A["tiled_in"])
    assert np.array_equal(p.compute_optical_flow_tiled(frames, 2), A["tiled_out_i2"])
    assert np.array_equal(p.compute_optical_flow_tiled(frames, 0), A["tiled_out_i0"])
    p.set_tile_mode(False)
    out = p.compute_optical_flow_tiled(frames, 5)
    assert out.dtype == np.float32 and out.shape == (40, 50, 2)
    assert np.array_equal(out, A["untiled_out_i5"])


def test_core_errors_and_middle_pick_match_reference():
    from processing.videoflow_core import VideoFlowCore
    core = VideoFlowCore("cpu")
    assert core.get_model_info() == J["model_info_unloaded"]
    args = {"not_loaded": torch.zeros(1, 5, 3, 8, 8), "not_tensor": np.zeros((1, 5, 3, 8, 8)),
            "ndim": torch.zeros(5, 3, 8, 8), "batch": torch.zeros(2, 5, 3, 8, 8), "channels": torch.zeros(1, 5, 4, 8, 8)}
    for key in ("not_loaded", "not_tensor", "ndim", "batch", "channels"):
        if key == "not_tensor":
            core.model = FakeModel()
        kind, msg = J["core_errors"][key]
        with pytest.raises({"RuntimeError": RuntimeError, "ValueError": ValueError}[kind]) as e:
            core.compute_flow_from_tensor(args[key])
        assert str(e.value) == msg
    for p in J["middle_pick"]:
        out = core.compute_flow_from_tensor(torch.ones(1, p["T"], 3, 8, 8))
        assert list(out.shape) == p["shape"]
        assert int(round(out[0, 0, 0].item())) - 1 == p["picked"]


def test_device_mismatch_rule():
    from processing.videoflow_core import VideoFlowCore
    core = VideoFlowCore("cuda")
    core.model = FakeModel()
    with pytest.raises(ValueError, match="doesn't match model device"):
        core.compute_flow_from_tensor(torch.zeros(1, 3, 3, 8, 8))     # cpu tensor, cuda engine


def test_missing_weights_message(tmp_path, monkeypatch):
    from processing.videoflow_core import VideoFlowCore
    monkeypatch.chdir(tmp_path)
    kind, msg = J["missing_weights"]
    with pytest.raises(FileNotFoundError) as e:
        VideoFlowCore("cpu", dataset="things", variant="noise", architecture="BOF").load_model()
    assert str(e.value) == msg


def test_compat_layer_surface():
    from processing.flow_inference import VideoFlowInference
    with quiet():
        v = VideoFlowInference("cpu", tile_mode=True, sequence_length=3)
    assert v.model is None and v.cfg is None and v.sequence_length == 3
    assert v.calculate_tile_grid(1920, 1080)[2:4] == (2, 1)
    v.set_sequence_length(7)
    assert v.get_processor().sequence_length == 7 and v.sequence_length == 7
    with pytest.raises(ValueError):
        v.set_sequence_length(11)
    v.set_tile_mode(False)
    assert v.get_processor().tile_mode is False
    assert v.get_core_engine() is v.get_processor().core
    assert v.get_model_info() == {"status": "not_loaded"}
    with pytest.raises(RuntimeError, match="Model not loaded"):
        v.compute_optical_flow([np.zeros((8, 8, 3), np.uint8)] * 3, 0)
    with pytest.raises(ValueError, match="cannot be empty"):
        v.validate_frames([], 0)
    with pytest.raises(AttributeError):
        v.no_such_method


def test_validate_frames_rules():
    p = make_proc()
    ok = [np.zeros((4, 4, 3), np.uint8)] * 2
    p.validate_frames(ok, 1)
    for frames, idx, pat in (("x", 0, "must be a list"), (ok, 2, "out of range"), ([1, 2], 0, "numpy arrays"),
                             ([np.zeros((4, 4))], 0, "3D arrays"), ([np.zeros((4, 4, 4), np.uint8)], 0, "3 color"),
                             ([np.zeros((4, 4, 3), np.int32)], 0, "Unsupported frame dtype"),
                             ([np.full((4, 4, 3), 300.0, np.float32)], 0, "Float frames")):
        with pytest.raises(ValueError, match=pat):
            p.validate_frames(frames, idx)
    p.validate_frames([np.full((4, 4, 3), 200.0, np.float32)], 0)     # 0..255 floats tolerated


def test_names_match_reference():
    from storage.filename_generator import generate_cache_directory, generate_output_filename
    for c in J["cache_dirs"]:
        assert generate_cache_directory("/data/some clip.v2.mp4", **c["kw"]) == c["dir"]
    for o in J["output_names"]:
        assert generate_output_filename("/x/clip.mov", **o["kw"]) == o["name"]
    from storage import FlowCacheManager
    assert FlowCacheManager().generate_cache_path("/data/clip.mp4", 0, 300, 5, False, True, 'videoflow', 'sintel',
                                                  'mof', 'standard') == \
        "/data/clip_flow_cache_videoflow_mof_sintel_standard_seq5_start0_frames300_tile"


def _members(path):
    z = np.load(path)
    return {k: {"dtype": str(z[k].dtype), "shape": list(z[k].shape),
                "value": z[k].tolist() if z[k].size <= 4 else None} for k in z.files}


def test_cache_files_match_reference(tmp_path):
    from storage import FlowCacheManager
    mgr = FlowCacheManager()
    flow = A["flow_small"]
    d = str(tmp_path / "c")
    assert list(mgr.check_cache_exists(os.path.join(d, "nope"), 3)) == J["cache_check_empty"]
    mgr.save_flow_to_cache(flow, d, 3, "both")
    assert sorted(os.listdir(d)) == J["cache_files"]
    raw = np.frombuffer(open(os.path.join(d, "flow_frame_000003.flo"), "rb").read(), dtype=np.uint8)
    assert np.array_equal(raw, A["flo_bytes"])
    assert _members(os.path.join(d, "flow_frame_000003.npz")) == J["npz_members"]
    assert np.array_equal(np.load(os.path.join(d, "flow_frame_000003.npz"))["flow"], flow)
    assert list(mgr.check_cache_exists(d, 4)) == J["cache_check_partial"]
    for i in (0, 1, 2):
        mgr.save_flow_to_cache(flow + i, d, i, "npz")
    assert list(mgr.check_cache_exists(d, 4)) == J["cache_check_complete"]
    assert np.array_equal(mgr.load_cached_flow(d, 2), A["cache_loaded_2"])
    d2 = str(tmp_path / "f")
    mgr.save_flow_to_cache(torch.from_numpy(flow), d2, 0, "flo")        # tensors accepted too
    assert list(mgr.check_cache_exists(d2, 1)) == J["cache_check_flo"]
    assert np.array_equal(mgr.load_cached_flow(d2, 0), A["cache_loaded_flo"])
    with pytest.raises(FileNotFoundError):
        mgr.load_cached_flow(d2, 9)
    with pytest.raises(ValueError, match="Invalid format_type"):
        mgr.load_cached_flow(d2, 0, "bmp")
    mgr.save_optical_flow_files(flow, os.path.join(d2, "base"), 5, "npz")
    assert _members(os.path.join(d2, "base_frame_000005.npz")) == J["flowfile_members"]
    open(os.path.join(d2, "bad.flo"), "wb").write(b"XXXX" + b"\0" * 8)
    with pytest.raises(ValueError, match="magic"):
        mgr.file_handler.load_flow_flo(os.path.join(d2, "bad.flo"))


def test_lod_pyramids_match_reference(tmp_path):
    from storage import FlowCacheManager, LODGenerator
    for tag in "abcd":
        lods = LODGenerator.generate_lods(A[f"lod_{tag}_in"], 4)
        for k, l in enumerate(lods):
            ref = A[f"lod_{tag}_{k}"]
            assert l.shape == ref.shape and l.dtype == ref.dtype
            assert np.array_equal(l, ref), (tag, k, np.abs(l - ref).max())
    mgr = FlowCacheManager()
    d = str(tmp_path)
    mgr.save_flow_lods(LODGenerator.generate_lods(A["flow_small"], 3), d, 1)
    assert sorted(n for n in os.listdir(d) if "lod" in n) == J["lod_files"]
    z = np.load(os.path.join(d, "flow_frame_000001_lod2.npz"))
    assert {k: {"dtype": str(z[k].dtype), "shape": list(z[k].shape)} for k in z.files} == J["lod_members"]
    assert mgr.check_flow_lods_exist(d, 1, 3) is False and mgr.load_flow_lod(d, 1, 2).shape == (2, 3, 2)


def test_device_manager_contract(monkeypatch):
    from config import DeviceManager
    dm = DeviceManager()
    assert dm.get_device("cpu") == "cpu"
    assert dm.get_device("cuda") == "cpu"          # cached (reference config/device_manager.py:26-27)
    dm.reset()
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    assert dm.get_device("cuda") == "cpu" and dm.get_device_info() == {"device": "cpu", "cuda_available": False}
    dm.reset()
    assert dm.get_device("auto") == "cpu"


def test_cli_flags_match_reference():
    """Same flag names, defaults, choices and kinds as the reference parser (flow_processor.py:1272-1332)."""
    import flow_processor
    parser = flow_processor.build_parser()
    ours = {}
    for a in parser._actions:
        if not a.option_strings or a.option_strings[0] == "-h":
            continue
        ent = {}
        if a.default is not None or a.option_strings[0] in ("--start-time", "--duration", "--flow-input", "--save-flow",
                                                            "--use-flow-cache", "--model-path"):
            ent["default"] = a.default
        if a.choices is not None:
            ent["choices"] = list(a.choices)
        if a.nargs == 0:
            ent["action"] = "store_true"
            ent.pop("default", None)
        if a.type is not None:
            ent["type"] = a.type.__name__
        ours[a.option_strings[0]] = ent
    assert ours == J["cli_flags"]


def test_cli_fills_cache_with_injected_cpu_model(tmp_path, monkeypatch):
    """flow_processor.py end to end on CPU (synthetic input, stand-in model): cache directory name,
    one .npz per frame, LOD files, second run is a cache hit."""
    import flow_processor
    import processing.videoflow_core as core_mod
    from vfml import get_cfg
    from vfml.weights import write_seeded_checkpoint
    write_seeded_checkpoint(str(tmp_path), get_cfg(), seed=0)
    monkeypatch.chdir(tmp_path)

    class Net(FakeModel):
        def load_state_dict(self, sd, strict=True):
            return None

    monkeypatch.setattr(core_mod, "build_network", lambda cfg: Net())
    monkeypatch.setattr(core_mod, "load_checked", lambda model, state, what: model.load_state_dict(state))
    argv = ["--input", "synthetic:64x48x5", "--output", str(tmp_path / "out"), "--device", "cpu", "--sequence-length",
            "3", "--interactive"]
    with quiet():
        assert flow_processor.main(argv) == 0
    cache = tmp_path / "out" / "synthetic_64x48x5_flow_cache_videoflow_mof_sintel_standard_seq3_start0_frames5"
    names = sorted(os.listdir(cache))
    assert [n for n in names if "lod" not in n] == [f"flow_frame_{i:06d}.npz" for i in range(5)]
    assert len([n for n in names if "lod" in n]) == 25
    z = np.load(cache / "flow_frame_000002.npz")
    assert z["flow"].shape == (48, 64, 2) and int(z["frame_idx"]) == 2
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        assert flow_processor.main(argv) == 0
    assert "nothing to compute" in out.getvalue()
    # --start-time / --duration select the frame range the reference's way (int(seconds * fps), synthetic clips run
    # at 30 fps) and name the cache directory after the RESOLVED range (reference flow_processor.py:667-677)
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        assert flow_processor.main(argv[:-1] + ["--start-time", "0.07", "--duration", "0.1", "--skip-lods", "--interactive"]) == 0
    assert "Start time: 0.07s -> frame 2" in out.getvalue() and "Duration: 0.1s -> 3 frames" in out.getvalue()
    cache2 = tmp_path / "out" / "synthetic_64x48x5_flow_cache_videoflow_mof_sintel_standard_seq3_start2_frames3"
    assert sorted(os.listdir(cache2)) == [f"flow_frame_{i:06d}.npz" for i in range(3)]
    # cached frame 0 of that job is frame 2 of the clip, computed on the 3-frame sub-clip
    from vfml.synth import synthetic_clip
    sub = synthetic_clip(5, 48, 64)[2:5]
    with quiet():
        p = make_proc(sequence_length=3)
    p.core.model = FakeModel()
    assert np.array_equal(np.load(cache2 / "flow_frame_000001.npz")["flow"], p.compute_optical_flow(sub, 1))
    with quiet():
        assert flow_processor.main(argv[:-1] + ["--start-time", "9", "--interactive"]) == 1    # past the end


def test_time_to_frame_and_range_clamp_match_reference():
    """--start-time / --duration / --start-frame / --frames resolve exactly as the reference's VideoInfo does
    (fixtures cut from it: tests/golden/make_time_fixtures.py)."""
    import flow_processor as fp
    cases = json.load(open(os.path.join(HERE, "golden", "time_ranges.json")))["cases"]
    assert len(cases) > 200
    for c in cases:
        s, n = c["start_frame"], c["frames"]
        if c["start_time"] is not None:
            s = fp.time_to_frame(c["start_time"], c["fps"])
        if c["duration"] is not None:
            n = fp.time_to_frame(c["duration"], c["fps"])
        assert [s, n] == c["resolved"], c
        if "error" in c:
            with pytest.raises(ValueError) as e:
                fp.validate_frame_range(s, n, c["total"])
            assert str(e.value) == c["error"]
        else:
            assert list(fp.validate_frame_range(s, n, c["total"])) == c["range"], c


def test_memflow_windows_and_errors_match_reference(tmp_path, monkeypatch):
    """MemFlow half of the processing package vs the reference's own modules (fixtures)."""
    from processing.memflow_core import MemFlowCore
    from processing.memflow_inference import MemFlowInference
    from processing.memflow_processor import MemFlowProcessor
    for w in J["memflow_windows"]:
        with quiet():
            p = MemFlowProcessor("cpu", sequence_length=w["T"])
        frames = [np.full((64, 64, 3), i, np.uint8) for i in range(6)]
        t = p.prepare_frame_sequence(frames, w["i"])
        assert t[0, :, 0, 0, 0].long().tolist() == w["idx"] and list(t.shape) == w["shape"]
        assert str(t.dtype) == w["dtype"] and str(t.device) == w["device"] and len(frames) == 6
    assert list(p.calculate_tile_grid(1920, 1080)) == J["memflow_tile_grid"]
    assert p.extract_tile(frames[0], {}) is frames[0]
    E = J["memflow_errors"]
    kinds = {"ValueError": ValueError, "TypeError": TypeError, "RuntimeError": RuntimeError,
             "FileNotFoundError": FileNotFoundError}
    monkeypatch.chdir(tmp_path)
    with quiet():
        mc = MemFlowCore("cpu")
    args = {"type": np.zeros((1, 2, 3, 64, 64)), "ndim": torch.zeros(2, 3, 64, 64), "batch": torch.zeros(2, 2, 3, 64, 64),
            "frames": torch.zeros(1, 1, 3, 64, 64), "channels": torch.zeros(1, 2, 4, 64, 64),
            "small": torch.zeros(1, 2, 3, 32, 64)}
    for key, arg in args.items():
        with pytest.raises(kinds[E[key][0]]) as e:
            mc.validate_input_tensor(arg)
        assert str(e.value) == E[key][1]
    with pytest.raises(RuntimeError) as e:
        mc.compute_flow_from_tensor(torch.zeros(1, 2, 3, 64, 64))
    assert str(e.value) == E["not_loaded"][1]
    with pytest.raises(FileNotFoundError) as e:
        mc.load_model()
    assert str(e.value) == E["missing"][1]
    seqs = {"empty": [], "one": [np.zeros((64, 64, 3), np.uint8)], "small": [np.zeros((32, 64, 3), np.uint8)] * 2,
            "mismatch": [np.zeros((64, 64, 3), np.uint8), np.zeros((64, 72, 3), np.uint8)]}
    for key, fr in seqs.items():
        with pytest.raises(ValueError) as e:
            p.validate_frame_sequence(fr)
        assert str(e.value) == E["seq_" + key][1]
    with quiet():
        inf = MemFlowInference("cpu", sequence_length=3)
    assert inf.model is None and inf.get_processor().sequence_length == 3
    assert inf.calculate_tile_grid(64, 64)[2:4] == (1, 1)
    assert inf.get_memory_usage() == {'device': 'cpu', 'note': 'CPU memory tracking not available'}
    # the value-range heuristic of the reference's inference script (memflow_inference_isolated.py:81-85)
    x = torch.tensor([0.0, 127.5, 255.0])
    assert torch.equal(MemFlowCore.normalise(x), torch.tensor([-1.0, 0.0, 1.0]))
    assert torch.equal(MemFlowCore.normalise(x / 255.0 * 1.5), 2 * (x / 255.0 * 1.5) - 1)
    assert torch.equal(MemFlowCore.normalise(x / 255.0), x / 255.0)


def test_async_cache_writer_writes_what_the_sync_path_writes(tmp_path):
    from storage import AsyncFlowCacheWriter, FlowCacheManager
    rng = np.random.default_rng(5)
    flows = [rng.standard_normal((33, 47, 2)).astype(np.float32) for _ in range(9)]
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    mgr = FlowCacheManager()
    for i, f in enumerate(flows):
        mgr.save_flow_to_cache(f, a, i, "both")
        mgr.save_flow_lods(mgr.lod_generator.generate_lods(f, 3), a, i)
    with AsyncFlowCacheWriter(b, "both", workers=4, num_lods=3) as w:
        for i, f in enumerate(flows):
            w.submit(f, i)
    assert sorted(os.listdir(a)) == sorted(os.listdir(b)) and len(os.listdir(b)) == 9 * (2 + 3)
    for name in os.listdir(a):
        if name.endswith(".flo"):
            assert open(os.path.join(a, name), "rb").read() == open(os.path.join(b, name), "rb").read()
        else:
            x, y = np.load(os.path.join(a, name)), np.load(os.path.join(b, name))
            assert x.files == y.files and all(np.array_equal(x[k], y[k]) and x[k].dtype == y[k].dtype for k in x.files)
    assert mgr.check_cache_exists(b, 9) == (True, "npz", []) and mgr.check_flow_lods_exist(b, 9, 3)
    bad = AsyncFlowCacheWriter(str(tmp_path / "file_not_dir"), "npz")
    open(tmp_path / "file_not_dir", "w").close()
    bad.submit(flows[0], 0)
    with pytest.raises(Exception):
        bad.close()


@pytest.mark.parametrize("mode", ["huffman", "stored", "zlib"])
def test_npz_writer_modes_read_back_like_savez_compressed(tmp_path, mode):
    """storage.cache_manager.write_npz (VFML_NPZ_DEFLATE): whatever produces the deflate stream, np.load - the reference's
    reader, storage/cache_manager.py:67-70 - sees the members np.savez_compressed (:47, :262) would have stored: same
    names in the same order, dtypes, shapes (0-d scalars stay 0-d) and values; the archive passes zipfile's CRC check."""
    import zipfile
    from storage.cache_manager import write_npz
    rng = np.random.default_rng(11)
    flow = rng.standard_normal((37, 53, 2)).astype(np.float32)
    members = {'flow': flow, 'frame_idx': 7, 'shape': flow.shape, 'dtype': str(flow.dtype), 'lod_level': 2,
               'min_flow': float(flow.min()), 'strided': flow[:, ::2], 'empty': np.zeros((0, 2), np.float32)}
    ref, got = str(tmp_path / "ref.npz"), str(tmp_path / "got.npz")
    np.savez_compressed(ref, **members)
    write_npz(got, members, mode)
    a, b = np.load(ref), np.load(got)
    assert a.files == b.files
    for k in a.files:
        assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k
    with zipfile.ZipFile(got) as z:
        assert z.testzip() is None
        assert [i.filename for i in z.infolist()] == [k + ".npy" for k in members]
    if mode == "huffman":       # an entropy-coded stream, not a stored one: the cache keeps its compressed size
        big = np.zeros((64, 64, 2), np.float32)
        write_npz(str(tmp_path / "z.npz"), {'flow': big}, mode)
        assert os.path.getsize(tmp_path / "z.npz") < big.nbytes // 4
    with pytest.raises(ValueError):
        write_npz(got, members, "lzma")
