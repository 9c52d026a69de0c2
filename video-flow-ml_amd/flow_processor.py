#!/usr/bin/env python3
"""flow_processor.py — command line of the drop-in, flow-field part.

Same flags as the reference CLI (flow_processor.py:1272-1332) and the same flow-cache contract
(cache directory name, flow_frame_%06d.npz members, completeness check, optional LODs), with the
frame loop of its cache-filling paths (:959-976 normal mode, :1460-1470 --interactive) running on
the MI355X engine and, under `python -m torch.distributed.run --nproc-per-node N flow_processor.py …`,
sharded over N GPUs (vfml.runner): whole-frame jobs with no collective at all - every rank writes the
cache files of its own fields - tiled jobs with the tiles streaming to rank 0 in chunked RCCL gathers.

Out of scope here: decoding/encoding video with OpenCV, flow visualisation encoders,
TAA, the side-by-side composer and the Tk/Qt tools; flags that only concern those are accepted and
reported as skipped.  Inputs: a `.npy` file holding uint8 frames [F,H,W,3]; `synthetic:WxHxF`
(vfml.synth); or any video file when OpenCV is importable.
"""
import argparse
import os
import sys
import time

script_dir = os.path.dirname(os.path.abspath(__file__))
if script_dir not in sys.path:
    sys.path.insert(0, script_dir)

import numpy as np
import torch

from config import DeviceManager
from processing.flow_inference import VideoFlowInference
from processing.memflow_inference import MemFlowInference
from storage import AsyncFlowCacheWriter, FlowCacheManager
from vfml import dist as vdist
from vfml.runner import ClipFeeder, run_sharded


def build_parser():
    p = argparse.ArgumentParser(description='VideoFlow Optical Flow Processor (MI355X engine)')
    p.add_argument('--input', default='big_buck_bunny_720p_h264.mov', help='Input video (.npy frames, synthetic:WxHxF, or a video file)')
    p.add_argument('--output', default='results', help='Output directory')
    p.add_argument('--device', default='auto', choices=['auto', 'cuda', 'cpu'])
    p.add_argument('--frames', type=int, default=1000, help='Maximum number of frames to process')
    p.add_argument('--start-frame', type=int, default=0)
    p.add_argument('--start-time', type=float, default=None)
    p.add_argument('--duration', type=float, default=None)
    p.add_argument('--fast', action='store_true', help='Fast mode (depth 6, 3 levels, radius 3)')
    p.add_argument('--flow-only', action='store_true')
    p.add_argument('--taa', action='store_true')
    p.add_argument('--flow-input', type=str, default=None)
    p.add_argument('--flow-format', choices=['gamedev', 'hsv', 'torchvision', 'motion-vectors-rg8', 'motion-vectors-rgb8'],
                   default='gamedev')
    p.add_argument('--motion-vectors-clamp-range', type=float, default=32.0)
    p.add_argument('--tile', action='store_true', help='1280x1280 tile mode')
    p.add_argument('--sequence-length', type=int, default=5)
    p.add_argument('--save-flow', choices=['flo', 'npz', 'both'], default=None)
    p.add_argument('--force-recompute', action='store_true')
    p.add_argument('--use-flow-cache', type=str, default=None)
    p.add_argument('--interactive', action='store_true')
    p.add_argument('--show-tiles', action='store_true')
    p.add_argument('--no-autoplay', action='store_true')
    p.add_argument('--skip-lods', action='store_true')
    p.add_argument('--uncompressed', action='store_true')
    p.add_argument('--model', choices=['videoflow', 'memflow'], default='videoflow')
    p.add_argument('--model-path', type=str, default=None)
    p.add_argument('--stage', choices=['sintel', 'things', 'kitti'], default='sintel')
    p.add_argument('--vf-dataset', choices=['sintel', 'things', 'kitti'], default='sintel')
    p.add_argument('--vf-architecture', choices=['mof', 'bof'], default='mof')
    p.add_argument('--vf-variant', choices=['standard', 'noise'], default='standard')
    return p


SYNTHETIC_FPS = 30.0     # frame rate of `synthetic:` clips and .npy frame stacks (they carry none)


def time_to_frame(time_seconds, fps):
    """Seconds -> frame number, the reference's rule (flow_processor.py:137-139, video/video_info.py:80-93)."""
    if fps <= 0:
        raise ValueError("Cannot convert time to frame: invalid FPS")
    return int(time_seconds * fps)


def validate_frame_range(start_frame, frame_count, total_frames):
    """The reference's clamp (video/video_info.py:110-132): negative starts become 0, a start past the end is an
    error, the count is cut to what the clip holds."""
    if start_frame < 0:
        start_frame = 0
    elif start_frame >= total_frames:
        raise ValueError(f"Start frame {start_frame} exceeds total frames {total_frames}")
    return start_frame, min(frame_count, total_frames - start_frame)


def probe_input(spec):
    """-> (fps, total_frames) of an input without decoding it."""
    if spec.startswith('synthetic:'):
        w, h, n = (int(v) for v in spec.split(':', 1)[1].lower().split('x'))
        return SYNTHETIC_FPS, n
    if spec.endswith('.npy'):
        return SYNTHETIC_FPS, int(np.load(spec, mmap_mode='r').shape[0])
    try:
        import cv2
    except ImportError:
        raise SystemExit(f"Cannot decode {spec}: OpenCV is not installed. Use a .npy frame stack or synthetic:WxHxF.")
    cap = cv2.VideoCapture(spec)
    fps, n = cap.get(cv2.CAP_PROP_FPS), int(cap.get(cv2.CAP_PROP_FRAME_COUNT))
    cap.release()
    return fps, n


def resolve_frame_range(spec, start_frame, max_frames, start_time=None, duration=None, log=print):
    """--start-time / --duration -> (start_frame, max_frames) exactly as the reference converts them
    (flow_processor.py:667-677, :1403-1420; video/frame_extractor.py:88-98): `int(seconds * fps)` replaces the frame
    arguments, then the range is clamped to the clip.  The resolved pair is what the cache directory is named after
    (storage/filename_generator.py: `_start{s}_frames{n}`)."""
    fps, total = probe_input(spec)
    if start_time is not None or duration is not None:
        log(f"Video FPS: {fps:.2f}")
        if start_time is not None:
            start_frame = time_to_frame(start_time, fps)
            log(f"Start time: {start_time}s -> frame {start_frame}")
        if duration is not None:
            max_frames = time_to_frame(duration, fps)
            log(f"Duration: {duration}s -> {max_frames} frames")
    return validate_frame_range(start_frame, max_frames, total)


def load_frames(spec, start_frame, max_frames, fps_default=SYNTHETIC_FPS):
    """-> (frames list of uint8 [H,W,3], fps, width, height, start_frame): the 5-tuple shape of the
    reference's FrameExtractor.extract_frames (video/frame_extractor.py:139)."""
    if spec.startswith('synthetic:'):
        from vfml.synth import synthetic_clip
        w, h, n = (int(v) for v in spec.split(':', 1)[1].lower().split('x'))
        frames = synthetic_clip(n, h, w)[start_frame:start_frame + max_frames]
    elif spec.endswith('.npy'):
        arr = np.load(spec, mmap_mode='r')
        if arr.ndim != 4 or arr.shape[3] != 3 or arr.dtype != np.uint8:
            raise ValueError(f"{spec}: expected uint8 [F,H,W,3], got {arr.dtype} {arr.shape}")
        frames = [np.ascontiguousarray(f) for f in arr[start_frame:start_frame + max_frames]]
    else:
        try:
            import cv2
        except ImportError:
            raise SystemExit(f"Cannot decode {spec}: OpenCV is not installed. Use a .npy frame stack or synthetic:WxHxF.")
        cap = cv2.VideoCapture(spec)
        fps_default = cap.get(cv2.CAP_PROP_FPS) or fps_default
        cap.set(cv2.CAP_PROP_POS_FRAMES, start_frame)
        frames = []
        while len(frames) < max_frames:
            ok, bgr = cap.read()
            if not ok:
                break
            frames.append(cv2.cvtColor(bgr, cv2.COLOR_BGR2RGB))
        cap.release()
    if not frames:
        raise SystemExit(f"No frames read from {spec}")
    h, w = frames[0].shape[:2]
    return frames, fps_default, w, h, start_frame


def main(argv=None):
    args = build_parser().parse_args(argv)
    rank, local_rank, world = vdist.init_distributed()
    log = print if rank == 0 else (lambda *a, **k: None)

    for flag, on in (("--taa", args.taa), ("--show-tiles", args.show_tiles), ("--flow-input", args.flow_input)):
        if on:
            log(f"note: {flag} concerns video composition / visualisation, which this build does not do; ignored")
    if not (args.input.startswith('synthetic:') or os.path.exists(args.input)):
        log(f"Error: Input video not found: {args.input}")
        return 1

    device = DeviceManager().get_device(args.device)
    if device == 'cuda' and world > 1:
        torch.cuda.set_device(local_rank)
        device = f"cuda:{local_rank}"
    try:
        start_frame, max_frames = resolve_frame_range(args.input, args.start_frame, args.frames, args.start_time,
                                                      args.duration, log)
    except ValueError as e:
        log(f"Error: {e}")
        return 1
    if max_frames <= 0:
        log(f"Error: empty frame range (start {start_frame}, {max_frames} frames)")
        return 1
    frames, fps, width, height, start = load_frames(args.input, start_frame, max_frames)
    n = len(frames)
    mgr = FlowCacheManager()
    cache_src = args.input if not args.input.startswith('synthetic:') else os.path.join(args.output, args.input.replace(':', '_') + ".npy")
    memflow = args.model == 'memflow'
    cache_dir = args.use_flow_cache or mgr.generate_cache_path(
        cache_src, start, n, args.sequence_length, args.fast, args.tile, args.model,
        args.stage if memflow else args.vf_dataset, args.vf_architecture, args.vf_variant)
    complete, fmt, missing = mgr.check_cache_exists(cache_dir, n)
    if complete and not args.force_recompute:
        log(f"Flow cache complete ({fmt}), nothing to compute: {cache_dir}")
        return 0

    if memflow:     # reference flow_processor.py:64-75: model path defaults to MemFlow_ckpt/MemFlowNet_{stage}.pth
        eng = MemFlowInference(device, args.model_path or f"MemFlow_ckpt/MemFlowNet_{args.stage}.pth", args.stage,
                               args.sequence_length)
    else:
        eng = VideoFlowInference(device, args.fast, args.tile, args.sequence_length, args.vf_dataset,
                                 args.vf_architecture, args.vf_variant)
    eng.load_model()
    proc = eng.get_processor()
    feeder = ClipFeeder(frames, device)     # frames go up through a pinned ring while earlier fields compute
    save_format = args.save_flow or 'npz'
    tiled = bool(args.tile and not memflow)
    num_lods = 0 if args.skip_lods else 5
    gpu_lods = 0 if (args.skip_lods or save_format == 'flo') else 5
    # Whole-frame jobs: EVERY rank writes the cache files of its own fields (one node = one filesystem; the reference's
    # cache is a directory of per-frame files, storage/cache_manager.py:247-262) - no gather, and the writers' compression
    # threads scale with the GPUs instead of funnelling every field into rank 0's.  Tiled jobs: tiles of one frame come
    # from several ranks, so they stream to rank 0 (chunked gathers), which pastes and writes.
    per_rank = not tiled
    cpus = max(1, vdist.host_cpu_share() // (world if per_rank else 1))
    writer = AsyncFlowCacheWriter(cache_dir, save_format, workers=min(16, cpus), num_lods=num_lods,
                                  manager=mgr) if (per_rank or rank == 0) else None
    t0 = time.time()
    # LOD levels are reduced on the GPU that computed the field (bit-identical to the reference's loop) and travel
    # with it; tiled frames are assembled on rank 0 and reduced by the writer's threads
    sink = (lambda k, field, lods: writer.submit(field, k, lods)) if writer is not None else None
    if per_rank:
        run_sharded(proc, None, range(n), tile_mode=False, rank=rank, world=world, feeder=feeder, local_sink=sink,
                    collect=False, num_lods=gpu_lods)
    else:
        run_sharded(proc, None, range(n), tile_mode=True, rank=rank, world=world, feeder=feeder, on_field=sink,
                    collect=False, num_lods=gpu_lods)
    if str(device).startswith('cuda'):
        torch.cuda.synchronize()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()                 # every rank's fields are computed and handed to a writer
    dt = time.time() - t0
    if writer is not None:
        writer.close()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()                 # every rank's files are on disk
    if rank == 0:
        dt_all = time.time() - t0
        log(f"{n} flow fields ({width}x{height}, seq {args.sequence_length}) in {dt:.2f} s = {n / dt:.2f} fields/s "
            f"on {world} GPU(s)")
        log(f"Flow cache written: {cache_dir} ({n / max(dt_all, 1e-9):.1f} fields/s end to end, incl. "
            f"{'no ' if args.skip_lods else ''}LODs)")
        complete, _, missing = mgr.check_cache_exists(cache_dir, n)
        if not complete:
            log(f"Error: flow cache incomplete after the job, missing frames {missing[:8]}{'...' if len(missing) > 8 else ''}")
            return 1
        if not args.interactive:
            log("note: video encoding / composition is out of scope for this build; the flow cache is the output")
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
