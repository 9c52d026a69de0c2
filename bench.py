#!/usr/bin/env python3
"""bench.py — flow-fields/sec of the MI355X engine on BASELINE.json's headline workload.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step = one flow field: MOF_sintel, seq_len=5, 1920x1080, synthetic clip (BASELINE.json configs[1]),
computed through the drop-in API (processing.VideoFlowProcessor -> VideoFlowCore -> HIP engine) from a
uint8 clip already resident in HBM, result left in HBM as [H,W,2] float32.  With N>1 every rank
runs K steps on its own frame range (weak scaling, no data-path collective) and one RCCL gather
brings the finished fields to rank 0 inside the timed region.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      dominant kernel (the split-f16 LDS-DMA implicit-GEMM conv), achieved algorithmic TFLOP/s from HIP
                events recorded around its launches inside the timed region vs 2500 / 3 TFLOP/s (three f16 MFMAs
                per product); `traffic`: HBM-side bytes per launch from the PMC passes recorded in
                profiles/r01_j_hbm_traffic.json beside `algorithmic_bytes_per_launch`; `hbm_kernels`: the lookup
  cpu_baseline  the CPU oracle (oracle/mof_oracle.py, "port") timed on this box's host cores on a
                bounded sample of the same workload (N=1 only), plus the engine-vs-oracle EPE on it
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# MI355X_MICROARCH.md, dense matrix peaks.  The default arithmetic ("f16x3") spends three
# v_mfma_f32_32x32x16_f16 per fp32-grade product, so its ceiling in ALGORITHMIC FLOP/s is a third of
# the f16 MFMA peak; "f32" runs v_mfma_f32_32x32x2_f32 and is priced against the f32 matrix peak.
F16_MATRIX_PEAK_TFLOPS = 2500.0
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E, ~8 TB/s
F32_MATRIX_PEAK_TFLOPS = 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--seq", type=int, default=5)
    ap.add_argument("--workload", default="mof1080p", choices=["mof1080p", "mof4k-tile", "memflow1080p", "bof720p"],
                    help="mof1080p = BASELINE.json configs[1] (the headline, default); the others are the remaining "
                         "GPU configs, measured with the same protocol for DESIGN.md (not the driver's line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-output", action="store_true",
                    help="compute all 2(T-2) flows of the model output per field instead of the one the reference "
                         "path keeps ([0, shape[1]//2]); the default drops, in the last two iterations, the centre "
                         "frames that cannot influence that flow (bit-identical for it)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads for the CPU baseline (0 = this process's CPU share: affinity / cgroup quota, "
                         "capped at 16 - the per-GPU share of the pool's boxes)")
    ap.add_argument("--cpu-sample-height", type=int, default=0,
                    help="CPU-baseline sample: 0 = one full-size field (minutes of CPU time), else one field on a "
                         "centre crop of this height (16:9), scaled to full size by the analytic FLOP ratio")
    args = ap.parse_args()

    if args.workload == "mof4k-tile":
        args.height, args.width = 2160, 3840
    elif args.workload == "bof720p":
        args.height, args.width, args.seq = 720, 1280, 9
    elif args.workload == "memflow1080p":
        args.seq = 3
    if args.workload != "mof1080p":
        args.no_cpu_baseline = True
    if args.full_output:
        os.environ["VFML_FULL_OUTPUT"] = "1"      # read by processing/videoflow_processor.py at import
    from vfml import dist as vdist, get_cfg, hip
    from vfml.synth import synthetic_clip
    from vfml.weights import write_seeded_checkpoint

    rank, local_rank, world = vdist.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    hip.lib()  # fail loudly if the HIP extension is missing

    # -- model through the reference's API path ------------------------------------------------
    work = tempfile.mkdtemp(prefix=f"vfml_bench_r{rank}_")
    write_seeded_checkpoint(work, get_cfg(), seed=0)
    cwd = os.getcwd()
    os.chdir(work)          # the reference resolves VideoFlow_ckpt/ relative to the CWD
    import contextlib
    import io
    from config import DeviceManager
    from processing.memflow_processor import MemFlowProcessor
    from processing.videoflow_processor import VideoFlowProcessor
    with contextlib.redirect_stdout(io.StringIO()):
        device_name = DeviceManager().get_device("cuda")
        device_name = device_name if world == 1 else f"cuda:{local_rank}"
        if args.workload == "memflow1080p":
            from vfml.memflow_net import memflow_cfg, seeded_memflow_state_dict
            os.makedirs("MemFlow_ckpt")
            torch.save(seeded_memflow_state_dict(memflow_cfg(), 0), "MemFlow_ckpt/MemFlowNet_sintel.pth")
            proc = MemFlowProcessor(device_name, sequence_length=args.seq)
        else:
            arch = "bof" if args.workload == "bof720p" else "mof"
            if arch == "bof":
                write_seeded_checkpoint(work, get_cfg(), seed=0, architecture="bof", dataset="things")
            proc = VideoFlowProcessor(device_name, tile_mode=args.workload == "mof4k-tile", sequence_length=args.seq,
                                      dataset="things" if arch == "bof" else "sintel", architecture=arch)
        proc.load_model()
    os.chdir(cwd)
    tiles = proc.calculate_tile_grid(args.width, args.height)[4] if args.workload == "mof4k-tile" else [None]

    def run_fields(clip, idxs, dst):
        """Flow fields of frames `idxs` into dst[k] ([H,W,2] each, on the device); in --tile mode
        tile-major (all fields of tile 0, then tile 1, ...: the order vfml.runner uses), so that
        consecutive items share their crop's cached frames."""
        if tiles == [None] and hasattr(proc, "compute_optical_flow_resident_batch"):
            step = getattr(proc, "TRI_BATCH", None) or getattr(proc, "PAIR_BATCH", 1)   # (several fields per call where it can)
            for k0 in range(0, len(idxs), step):
                for j, f in enumerate(proc.compute_optical_flow_resident_batch(clip, idxs[k0:k0 + step])):
                    dst[k0 + j].copy_(f)
            return
        for t in tiles:
            for k, i in enumerate(idxs):
                f = proc.compute_optical_flow_resident(clip, i, tile=t)
                if t is None:
                    dst[k].copy_(f)
                else:
                    dst[k, t['y']:t['y'] + t['height'], t['x']:t['x'] + t['width']].copy_(f)

    # -- synthetic clip, uploaded once ---------------------------------------------------------
    K, Wm, T = args.steps, args.warmup, args.seq
    per_rank = K + Wm
    nframes = per_rank + T - 1
    # every rank generates the same clip and works on its own window range (frame-range sharding
    # with a T//2 halo; here the "global" clip is world * per_rank fields long)
    clip_np = synthetic_clip(world * per_rank + T - 1, args.height, args.width)
    lo = rank * per_rank
    clip = proc.upload_clip(clip_np[lo:lo + nframes])
    half = T // 2
    fields = [half + i for i in range(per_rank)]          # local indices with a full window
    torch.cuda.synchronize()

    out = torch.empty(max(K, Wm), args.height, args.width, 2, device=dev)
    run_fields(clip, fields[:Wm], out)
    torch.cuda.synchronize()
    vdist.barrier(dev)

    out = out[:K]
    hip.profile_begin()
    t0 = time.perf_counter()
    run_fields(clip, fields[Wm:], out)
    torch.cuda.synchronize()
    t_compute = time.perf_counter() - t0
    tg = time.perf_counter()
    gathered = vdist.gather_to_rank0(out.view(-1), [out.numel()] * world)
    torch.cuda.synchronize()
    t_gather = time.perf_counter() - tg
    vdist.barrier(dev)
    elapsed = time.perf_counter() - t0
    prof_hbm = hip.profile_end_hbm()
    prof = hip.profile_end()
    elapsed = vdist.max_over_ranks(elapsed, dev)
    if rank != 0:
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return
    assert gathered is not None and len(gathered) == world

    total_fields = K * world
    core = proc.core if hasattr(proc, "core") else proc.core_engine
    depth = core.cfg.decoder_depth
    result = {
        "metric": "flow-fields/sec @1080p seq5 MOF_sintel" if args.workload == "mof1080p"
                  else f"flow-fields/sec {args.workload}",
        "value": total_fields / elapsed,
        "unit": "flow-fields/s",
        "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": 1000.0 * elapsed / K,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": {"f16x3": "f16x3 (split-f16 MFMA, f32 accumulate, fp32-grade)", "f32": "f32"}[
            getattr(core.cfg, "precision", "f16x3")],
        "data": "synthetic",
        "config": {"workload": {"mof1080p": "MOF_sintel", "mof4k-tile": "MOF_sintel --tile (6 tiles/frame)",
                                "memflow1080p": "MemFlowNet_sintel (pair path)", "bof720p": "BOF_things"}[args.workload]
                               + f" seq_len={T} {args.width}x{args.height} synthetic clip, decoder_depth={depth}, "
                                 f"seeded weights",
                   "fields_per_gpu": K, "clip_frames_per_gpu": nframes, "parallelism": f"frames-dp{world}",
                   "inputs": "uint8 clip resident in HBM", "outputs": "[H,W,2] f32 in HBM, gathered to rank 0",
                   "model_output": ("all 2(T-2) flows per field (--full-output)" if args.full_output else
                                    "the flow the reference path keeps, [0, shape[1]//2]; the last two iterations "
                                    "skip the centre frames outside its dependency cone (bit-identical for it; "
                                    "--full-output computes all 2(T-2))")},
        "compute_ms_per_step": 1000.0 * t_compute / K,
        "gather_ms": 1000.0 * t_gather,
    }

    # -- roofline of the dominant kernel -------------------------------------------------------
    if prof:
        name, d = max(prof.items(), key=lambda kv: kv[1]["ms"])
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        split = "split" in name or "dma" in name      # both are split-f16 kernels (3 f16 MFMAs per product)
        peak = F16_MATRIX_PEAK_TFLOPS / 3.0 if split else F32_MATRIX_PEAK_TFLOPS
        traffic, traffic_note = hbm_traffic_from_profiles(name, args.workload)
        result["roofline"] = {
            "kernel": name, "bound": "mfma", "achieved": achieved, "peak": peak,
            "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic, "traffic_note": traffic_note,
            "peak_note": ("algorithmic FLOP/s; peak = 2500 TFLOP/s dense f16 MFMA / 3 MFMAs per product"
                          if split else "f32 matrix peak (v_mfma_f32_32x32x2_f32)"),
            "mfma_executed_tflops": achieved * (3.0 if split else 1.0),
            "launches": d["launches"], "avg_launch_us": 1000.0 * d["ms"] / d["launches"],
            "gflop_per_launch": d["flops"] / d["launches"] / 1e9,
            "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
            "share_of_step": d["ms"] / (1000.0 * t_compute),
            "all_variants": {k: {"ms": v["ms"], "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12,
                                 "launches": v["launches"]} for k, v in prof.items()},
            # the HBM-bound stages of the path beside it: algorithmic bytes / HIP-event time
            "hbm_kernels": {k: {"bound": "hbm", "achieved": v["bytes"] / (v["ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": v["bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "launches": v["launches"], "avg_launch_us": 1000.0 * v["ms"] / v["launches"],
                                "share_of_step": v["ms"] / (1000.0 * t_compute)} for k, v in prof_hbm.items()},
        }

    # -- CPU baseline (oracle) on a bounded sample of the same workload -------------------------
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, proc, clip_np, fields[Wm], out[0], T)
    print(json.dumps(result))


def hbm_traffic_from_profiles(kernel, workload):
    """HBM bytes per launch of `kernel` from the PMC passes over this very command (a live run cannot collect them:
    rocprofv3 has to wrap the process).  profiles/r01_j_hbm_traffic.json holds the per-kernel means of the default
    workload; None for any other workload or when the file / the kernel is missing."""
    path = os.path.join(ROOT, "profiles", "r01_j_hbm_traffic.json")
    if workload != "mof1080p" or not os.path.exists(path):
        return None, "no PMC passes recorded for this workload"
    try:
        with open(path) as f:
            rec = json.load(f)["kernels"].get(kernel)
    except (OSError, ValueError, KeyError):
        rec = None
    if not rec:
        return None, "kernel not in profiles/r01_j_hbm_traffic.json"
    return rec["hbm_bytes_per_launch"], ("bytes per launch, mean over %d launches: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 from separate "
                                         "rocprofv3 --pmc passes over `bench.py --steps 2 --warmup 1` (profiles/r01_j_hbm_traffic.json)"
                                         % rec["launches"])


def host_cpu_share(cap=16):
    """CPUs this process may actually use: scheduler affinity, cgroup-v2 quota, capped (os.cpu_count()
    reports the whole host, and oversubscribing a 16-CPU share with 100+ threads is far slower)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def cpu_baseline(args, proc, clip_np, field_idx, engine_field, T):
    """The CPU oracle (fp32 PyTorch restatement) on the host cores, on a bounded sample of the
    workload: one field of the same clip/window/weights, by default on a 960x544 centre crop, scaled
    to a full-size field by the analytic FLOP ratio (vfml/flops.py; correlation is quadratic in area,
    convolutions linear).  Also returns the engine-vs-oracle end-point error on that sample."""
    from oracle import mof_oracle as mo
    from vfml.flops import field_work
    cores = args.cpu_threads or host_cpu_share()
    torch.set_num_threads(cores)
    depth = proc.core.cfg.decoder_depth
    ocfg = mo.get_cfg()
    ocfg.decoder_depth = depth
    ora = mo.build_network(ocfg).eval()
    ora.load_state_dict({k: v.cpu() for k, v in proc.core.model.state_dict().items()})
    idx = proc.window_indices(len(clip_np), field_idx)
    win = np.stack([clip_np[i] for i in idx])
    H, W = win.shape[1:3]
    full = not (args.cpu_sample_height and args.cpu_sample_height < H)
    if not full:
        h = args.cpu_sample_height // 8 * 8
        w = min(W, (h * 16 // 9) // 8 * 8)
        y0, x0 = (H - h) // 2 // 8 * 8, (W - w) // 2 // 8 * 8
        win = np.ascontiguousarray(win[:, y0:y0 + h, x0:x0 + w])
    x = torch.from_numpy(win.astype(np.float32) / 255.0).permute(0, 3, 1, 2)[None]
    t0 = time.perf_counter()
    flows, _ = ora(x, {})
    dt = time.perf_counter() - t0
    ref = flows[0, flows.shape[1] // 2].permute(1, 2, 0)
    if full:
        got = engine_field.cpu()
        ratio = 1.0
        sample = f"1 full {W}x{H} seq{T} field (field {field_idx} of the clip)"
    else:
        eng, _ = proc.core.model.forward_u8(torch.from_numpy(win).to(engine_field.device), return_lowres=False)
        got = eng[0, eng.shape[1] // 2].permute(1, 2, 0).cpu()
        ratio = field_work(H, W, T, depth)["total_flops"] / field_work(win.shape[1], win.shape[2], T, depth)["total_flops"]
        sample = (f"1 field on the {win.shape[2]}x{win.shape[1]} centre crop of the window of field {field_idx} "
                  f"({dt:.1f} s), scaled to {W}x{H} by the analytic FLOP ratio {ratio:.2f}")
    epe = (got - ref).pow(2).sum(-1).sqrt()
    return {"value": 1.0 / (dt * ratio), "unit": "flow-fields/s", "cores": cores, "kind": "port",
            "sample": sample + ", fp32 PyTorch CPU oracle (oracle/mof_oracle.py), all host threads",
            "sample_seconds": dt, "flop_ratio_to_full": ratio, "threads": torch.get_num_threads(),
            "epe_mean_px": float(epe.mean()), "epe_max_px": float(epe.max()),
            "epe_note": "engine vs oracle on the same sample, same seeded weights; tolerance 1e-3 px mean"}


if __name__ == "__main__":
    main()
