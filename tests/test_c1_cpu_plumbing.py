"""BASELINE.json configs[0]: MOF_sintel seq_len=3 on 8 x 256x256 synthetic frames, device=cpu via
DeviceManager — the reference's whole host path (device pick, checkpoint naming/loading, windows,
index pick, numpy output, cache files, completeness check, LODs) with no GPU.

The shipped engine has no CPU arithmetic (MOFNetHIP.forward raises on CPU tensors), so the model
arithmetic here comes from the CPU oracle, injected through the one seam the reference itself has:
the `build_network` name imported by processing/videoflow_core.py."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch


@pytest.fixture()
def workdir(tmp_path, monkeypatch):
    from vfml import get_cfg
    from vfml.weights import write_seeded_checkpoint
    write_seeded_checkpoint(str(tmp_path), get_cfg(), seed=0, dataparallel_prefix=True)   # 'module.' keys, :106-108
    monkeypatch.chdir(tmp_path)
    return tmp_path


def test_product_engine_refuses_cpu(workdir):
    from config import DeviceManager
    from processing.flow_inference import VideoFlowInference
    with contextlib.redirect_stdout(io.StringIO()):
        eng = VideoFlowInference(DeviceManager().get_device("cpu"), sequence_length=3)
        eng.load_model()                                   # building + strict loading works anywhere
    assert eng.get_model_info()["config"] == {"decoder_depth": 12, "corr_levels": 4, "corr_radius": 4}
    frames = [np.zeros((128, 128, 3), np.uint8)] * 3
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        eng.compute_optical_flow(frames, 1)


def test_c1_cpu_plumbing_end_to_end(workdir, monkeypatch):
    import processing.videoflow_core as core_mod
    from config import DeviceManager
    from oracle import mof_oracle as mo
    from processing.flow_inference import VideoFlowInference
    from storage import FlowCacheManager
    from vfml.synth import synthetic_clip

    def oracle_network(cfg):
        ocfg = mo.get_cfg()
        ocfg.decoder_depth, ocfg.corr_levels, ocfg.corr_radius = cfg.decoder_depth, cfg.corr_levels, cfg.corr_radius
        ocfg.decoder_depth = 2          # keep the CPU suite fast; plumbing is what is under test
        return mo.build_network(ocfg)

    monkeypatch.setattr(core_mod, "build_network", oracle_network)
    dm = DeviceManager()
    device = dm.get_device("cpu")
    assert device == "cpu" and dm.get_device_info()["device"] == "cpu"
    with contextlib.redirect_stdout(io.StringIO()):
        eng = VideoFlowInference(device, fast_mode=False, tile_mode=False, sequence_length=3,
                                 dataset="sintel", architecture="mof", variant="standard")
        eng.load_model()
    assert eng.model is not None and eng.cfg.model == "VideoFlow_ckpt/MOF_sintel.pth"
    info = eng.get_model_info()
    assert info["status"] == "loaded" and info["architecture"] == "MOF" and info["sequence_length"] == 3
    assert info["compatibility_layer"] == "VideoFlowInference" and info["device"] == "cpu"
    assert eng.get_memory_usage() == {"message": "Memory tracking only available for CUDA devices"}

    frames = synthetic_clip(8, 256, 256)
    mgr = FlowCacheManager()
    cache = mgr.generate_cache_path(str(workdir / "clip.mp4"), 0, len(frames), 3, False, False, "videoflow",
                                    "sintel", "mof", "standard")
    assert os.path.basename(cache) == "clip_flow_cache_videoflow_mof_sintel_standard_seq3_start0_frames8"
    flows = []
    for i in range(len(frames)):                      # the loop of flow_processor.py:1460-1470
        eng.validate_frames(frames, i)
        f = eng.compute_optical_flow_tiled(frames, i)
        assert f.shape == (256, 256, 2) and f.dtype == np.float32 and np.isfinite(f).all()
        mgr.save_flow_to_cache(f, cache, i, "npz")
        flows.append(f)
    assert mgr.check_cache_exists(cache, len(frames)) == (True, "npz", [])
    assert np.array_equal(mgr.load_cached_flow(cache, 5), flows[5])
    # the field the reference indexes for T=3 is flow[0, 1]: the centre frame's backward flow
    x = eng.prepare_frame_sequence(frames, 4)
    ref, _ = eng.model(x, {})
    assert np.array_equal(flows[4], ref[0, 1].permute(1, 2, 0).numpy())
    # clip borders reuse frames (front padding at i=0, back padding at the end)
    assert eng.get_processor().window_indices(8, 0) == [0, 0, 1] and eng.get_processor().window_indices(8, 7) == [6, 7, 7]
    lods = mgr.lod_generator.generate_lods(flows[0], 5)
    assert [l.shape for l in lods] == [(256, 256, 2), (128, 128, 2), (64, 64, 2), (32, 32, 2), (16, 16, 2)]
    mgr.save_flow_lods(lods, cache, 0)
    assert mgr.load_flow_lod(cache, 0, 4).shape == (16, 16, 2)
