"""Cut golden fixtures from the REFERENCE's own host code (run in the build container only).

    python tests/golden/make_fixtures.py [/root/reference]

The reference's model arithmetic is absent (empty VideoFlow/ submodule), but its host plumbing is
importable: storage.* and config.* directly, processing.videoflow_{processor,core} once three stub
modules stand in for the submodule imports (core.Networks, utils.utils,
configs.multiframes_sintel_submission).  This script drives those modules on small deterministic
inputs and records inputs + outputs as data (host_plumbing.json, host_plumbing.npz).  Nothing of
the reference's source text is stored.  The GPU box never runs this file.
"""
import io
import json
import os
import sys
import tempfile
import types
import contextlib

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub_submodule():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class IdentityPadder:
        def __init__(self, dims):
            self.dims = tuple(dims)

        def pad(self, x):
            return x

        def unpad(self, x):
            return x

    class Bag:
        pass

    def get_cfg():
        b = Bag()
        b.model, b.decoder_depth, b.corr_levels, b.corr_radius = "", 12, 4, 4
        return b

    mod("core")
    mod("core.Networks", build_network=lambda cfg: None)
    mod("utils")
    mod("utils.utils", InputPadder=IdentityPadder)
    mod("configs")
    mod("configs.multiframes_sintel_submission", get_cfg=get_cfg)
    # processing/__init__ also pulls the MemFlow half, which needs nothing at import time
    return get_cfg


def main():
    sys.path.insert(0, REF)
    _stub_submodule()
    quiet = contextlib.redirect_stdout(io.StringIO())
    with quiet:
        from storage import cache_manager as cm
        from storage import filename_generator as fg
        from processing.videoflow_processor import VideoFlowProcessor
        from processing.videoflow_core import VideoFlowCore

    J, A = {}, {}

    # ---- frame windows (processing/videoflow_processor.py:122-162) -------------------------------
    wins = []
    for T in (1, 2, 3, 4, 5, 7, 9):
        for n in (1, 2, 3, 5, 8, 12):
            with quiet:
                p = VideoFlowProcessor("cpu", sequence_length=T)
            frames = [np.full((2, 2, 3), i, np.uint8) for i in range(n)]
            for i in range(n):
                t = p.prepare_frame_sequence(list(frames), i)
                idx = (t[0, :, 0, 0, 0] * 255.0).round().long().tolist()
                wins.append({"T": T, "n": n, "i": i, "idx": idx, "shape": list(t.shape), "dtype": str(t.dtype)})
    J["windows"] = wins
    rng = np.random.default_rng(1)
    u8 = [rng.integers(0, 256, (6, 5, 3), dtype=np.uint8) for _ in range(4)]
    f32 = [rng.random((6, 5, 3), dtype=np.float32) for _ in range(4)]
    with quiet:
        p = VideoFlowProcessor("cpu", sequence_length=3)
    A["seq_u8_in"] = np.stack(u8)
    A["seq_u8_out"] = p.prepare_frame_sequence(list(u8), 1).numpy()
    A["seq_f32_in"] = np.stack(f32)
    A["seq_f32_out"] = p.prepare_frame_sequence(list(f32), 2).numpy()

    # ---- tile grids (:73-120) ---------------------------------------------------------------------
    tiles = []
    for (w, h) in ((1920, 1080), (3840, 2160), (256, 256), (1280, 1280), (1281, 1279), (2560, 720), (640, 3000)):
        tw, th, cols, rows, info = p.calculate_tile_grid(w, h)
        tiles.append({"w": w, "h": h, "tile": [tw, th], "cols": cols, "rows": rows,
                      "tiles": [[t["x"], t["y"], t["width"], t["height"], t["col"], t["row"]] for t in info]})
    tw, th, cols, rows, info = p.calculate_tile_grid(1000, 700, tile_size=256)
    tiles.append({"w": 1000, "h": 700, "tile_size": 256, "tile": [tw, th], "cols": cols, "rows": rows,
                  "tiles": [[t["x"], t["y"], t["width"], t["height"], t["col"], t["row"]] for t in info]})
    J["tile_grids"] = tiles

    # ---- tiled flow assembly with a fake core (:231-283) -----------------------------------------
    class FakeModel(torch.nn.Module):
        def forward(self, x, _):
            B, T, C, H, W = x.shape
            base = x[:, :, :2].mean(dim=1, keepdim=True)           # depends on every frame of the crop
            flows = torch.cat([base * (k + 1) for k in range(2 * (T - 2))], dim=1)
            return flows.view(B, 2 * (T - 2), 2, H, W), None

    with quiet:
        pt = VideoFlowProcessor("cpu", tile_mode=True, sequence_length=5)
    pt.core.model = FakeModel()
    orig_grid = pt.calculate_tile_grid
    pt.calculate_tile_grid = lambda w, h, tile_size=1280: orig_grid(w, h, 16)   # small tiles, same code path
    fr = [rng.integers(0, 256, (40, 50, 3), dtype=np.uint8) for _ in range(6)]
    A["tiled_in"] = np.stack(fr)
    A["tiled_out_i2"] = pt.compute_optical_flow_tiled(list(fr), 2)
    A["tiled_out_i0"] = pt.compute_optical_flow_tiled(list(fr), 0)
    pt.set_tile_mode(False)
    A["untiled_out_i5"] = pt.compute_optical_flow_tiled(list(fr), 5)

    # ---- core: index pick + validation messages (processing/videoflow_core.py:130-198) ------------
    with quiet:
        core = VideoFlowCore("cpu")
    errs = {}
    try:
        core.compute_flow_from_tensor(torch.zeros(1, 5, 3, 8, 8))
    except Exception as e:
        errs["not_loaded"] = [type(e).__name__, str(e)]
    core.model = FakeModel()
    for key, arg in (("not_tensor", np.zeros((1, 5, 3, 8, 8))), ("ndim", torch.zeros(5, 3, 8, 8)),
                     ("batch", torch.zeros(2, 5, 3, 8, 8)), ("channels", torch.zeros(1, 5, 4, 8, 8))):
        try:
            core.compute_flow_from_tensor(arg)
        except Exception as e:
            errs[key] = [type(e).__name__, str(e)]
    J["core_errors"] = errs
    picks = []
    for T in (3, 4, 5, 7, 9):
        x = torch.ones(1, T, 3, 8, 8)
        out = core.compute_flow_from_tensor(x)
        picks.append({"T": T, "picked": int(round(out[0, 0, 0].item())) - 1, "shape": list(out.shape)})
    J["middle_pick"] = picks
    J["model_info_unloaded"] = VideoFlowCore("cpu").get_model_info()
    try:
        os.chdir(tempfile.mkdtemp())
        VideoFlowCore("cpu", dataset="things", variant="noise", architecture="BOF").load_model()
    except Exception as e:
        J["missing_weights"] = [type(e).__name__, str(e)]

    # ---- names (storage/filename_generator.py) ----------------------------------------------------
    names = []
    for kw in ({}, {"fast_mode": True}, {"tile_mode": True, "fast_mode": True}, {"model": "memflow", "dataset": "sintel"},
               {"model": "videoflow", "dataset": "sintel", "architecture": "mof", "variant": "standard",
                "sequence_length": 5, "max_frames": 300}, {"start_frame": 17, "max_frames": 64, "sequence_length": 9,
                                                             "architecture": "bof", "dataset": "things"}):
        names.append({"kw": kw, "dir": fg.generate_cache_directory("/data/some clip.v2.mp4", **kw)})
    J["cache_dirs"] = names
    outs = []
    for kw in ({}, {"max_frames": 300, "flow_only": True, "flow_format": "motion-vectors-rg8", "fps": 30},
               {"start_time": 1.5, "duration": 2.0, "taa": True, "fps": 23.976}, {"start_frame": 10, "fast_mode": True,
                                                                                     "tile_mode": True, "uncompressed": True},
               {"flow_only": True, "flow_format": "hsv"}, {"flow_only": True, "flow_format": "torchvision", "fps": 59.94}):
        outs.append({"kw": kw, "name": fg.generate_output_filename("/x/clip.mov", **kw)})
    J["output_names"] = outs

    # ---- cache files (storage/cache_manager.py) ---------------------------------------------------
    flow = rng.standard_normal((7, 9, 2)).astype(np.float32) * 5
    A["flow_small"] = flow
    d = tempfile.mkdtemp()
    mgr = cm.FlowCacheManager()
    J["cache_check_empty"] = list(mgr.check_cache_exists(os.path.join(d, "nope"), 3))
    mgr.save_flow_to_cache(flow, d, 3, "both")
    A["flo_bytes"] = np.frombuffer(open(os.path.join(d, "flow_frame_000003.flo"), "rb").read(), dtype=np.uint8)
    z = np.load(os.path.join(d, "flow_frame_000003.npz"))
    J["npz_members"] = {k: {"dtype": str(z[k].dtype), "shape": list(z[k].shape),
                            "value": z[k].tolist() if z[k].size <= 4 else None} for k in z.files}
    J["cache_files"] = sorted(os.listdir(d))
    J["cache_check_partial"] = list(mgr.check_cache_exists(d, 4))
    for i in (0, 1, 2):
        mgr.save_flow_to_cache(flow + i, d, i, "npz")
    J["cache_check_complete"] = list(mgr.check_cache_exists(d, 4))
    A["cache_loaded_2"] = mgr.load_cached_flow(d, 2)
    d2 = tempfile.mkdtemp()
    mgr.save_flow_to_cache(flow, d2, 0, "flo")
    J["cache_check_flo"] = list(mgr.check_cache_exists(d2, 1))
    A["cache_loaded_flo"] = mgr.load_cached_flow(d2, 0)
    mgr.save_optical_flow_files(flow, os.path.join(d2, "base"), 5, "npz")
    z = np.load(os.path.join(d2, "base_frame_000005.npz"))
    J["flowfile_members"] = {k: {"dtype": str(z[k].dtype), "shape": list(z[k].shape),
                                 "value": z[k].tolist() if z[k].size <= 4 else None} for k in z.files}
    # LOD pyramids (:77-161)
    for tag, shp in (("a", (5, 7, 2)), ("b", (8, 8, 2)), ("c", (9, 4, 2)), ("d", (1, 6, 2))):
        f = rng.standard_normal(shp).astype(np.float32) * 3
        lods = cm.LODGenerator.generate_lods(f, 4)
        A[f"lod_{tag}_in"] = f
        for k, l in enumerate(lods):
            A[f"lod_{tag}_{k}"] = l
    with quiet:
        mgr.save_flow_lods(cm.LODGenerator.generate_lods(flow, 3), d, 1)
    J["lod_files"] = sorted(n for n in os.listdir(d) if "lod" in n)
    z = np.load(os.path.join(d, "flow_frame_000001_lod2.npz"))
    J["lod_members"] = {k: {"dtype": str(z[k].dtype), "shape": list(z[k].shape)} for k in z.files}

    # ---- MemFlow windows / tensors (processing/memflow_processor.py:97-139) and core checks -----------
    with quiet:
        from processing.memflow_processor import MemFlowProcessor
        from processing.memflow_core import MemFlowCore
    mw = []
    for T in (1, 2, 3, 5):
        with quiet:
            mp_ = MemFlowProcessor("cpu", sequence_length=T)
        fr = [np.full((64, 64, 3), i, np.uint8) for i in range(6)]
        for i in range(6):
            t = mp_.prepare_frame_sequence(list(fr), i)
            mw.append({"T": T, "i": i, "idx": t[0, :, 0, 0, 0].long().tolist(), "shape": list(t.shape),
                       "dtype": str(t.dtype), "device": str(t.device)})
    J["memflow_windows"] = mw
    J["memflow_tile_grid"] = list(mp_.calculate_tile_grid(1920, 1080))
    merr = {}
    with quiet:
        mc = MemFlowCore("cpu")
    for key, arg in (("type", np.zeros((1, 2, 3, 64, 64))), ("ndim", torch.zeros(2, 3, 64, 64)),
                     ("batch", torch.zeros(2, 2, 3, 64, 64)), ("frames", torch.zeros(1, 1, 3, 64, 64)),
                     ("channels", torch.zeros(1, 2, 4, 64, 64)), ("small", torch.zeros(1, 2, 3, 32, 64))):
        try:
            mc.validate_input_tensor(arg)
        except Exception as e:
            merr[key] = [type(e).__name__, str(e)]
    try:
        mc.compute_flow_from_tensor(torch.zeros(1, 2, 3, 64, 64))
    except Exception as e:
        merr["not_loaded"] = [type(e).__name__, str(e)]
    try:
        mc.load_model()
    except Exception as e:
        merr["missing"] = [type(e).__name__, str(e)]
    for key, fr_ in (("empty", []), ("one", [np.zeros((64, 64, 3), np.uint8)]),
                     ("small", [np.zeros((32, 64, 3), np.uint8)] * 2),
                     ("mismatch", [np.zeros((64, 64, 3), np.uint8), np.zeros((64, 72, 3), np.uint8)])):
        try:
            mp_.validate_frame_sequence(fr_)
        except Exception as e:
            merr["seq_" + key] = [type(e).__name__, str(e)]
    J["memflow_errors"] = merr

    # ---- CLI flag surface (flow_processor.py:1272-1332): names, defaults, choices — read from the
    # parser definition with the ast module (the file itself needs cv2 to import)
    import ast
    tree = ast.parse(open(os.path.join(REF, "flow_processor.py")).read())
    flags = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument" and node.args:
            name = ast.literal_eval(node.args[0])
            kw = {k.arg: k.value for k in node.keywords}
            ent = {}
            if "default" in kw:
                ent["default"] = ast.literal_eval(kw["default"])
            if "choices" in kw:
                ent["choices"] = ast.literal_eval(kw["choices"])
            if "action" in kw:
                ent["action"] = ast.literal_eval(kw["action"])
            if "type" in kw:
                ent["type"] = kw["type"].id
            flags[name] = ent
    J["cli_flags"] = flags

    with open(os.path.join(HERE, "host_plumbing.json"), "w") as f:
        json.dump(J, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "host_plumbing.npz"), **A)
    print("wrote", len(J), "json sections and", len(A), "arrays")


if __name__ == "__main__":
    main()
