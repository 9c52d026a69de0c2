#!/usr/bin/env python3
"""bench.py — flow-fields/sec of the MI355X engine on BASELINE.json's headline workload.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step = one flow field: MOF_sintel, seq_len=5, 1920x1080, synthetic clip (BASELINE.json configs[1]), computed
through the drop-in API (processing.VideoFlowProcessor -> VideoFlowCore -> HIP engine) by the job loop of the CLI
(vfml.runner.run_sharded).

Timed region: uint8 frames resident in HOST memory + model loaded  ->  last field resident in
rank-0 HOST memory.  Inside it: every new frame goes up through the pinned ring (ClipFeeder), K fields are computed
per rank, each field is copied to pinned host memory while the next one computes, and with N > 1 the ranks' fields
stream to rank 0 in chunked RCCL gathers beside the computation (weak scaling: K fields per rank, no data-path
collective).  W warm-up fields of the same sliding job run before it, so the timed fields are the steady state of a
long job (one new frame per field).  `engine_ms_per_step` is the same loop with inputs and outputs left in HBM.

Extra objects on the line:
  roofline      dominant kernel, from a SEPARATE pass after the timed region (per-launch HIP events on the launch
                stream perturb the timing, so they never run inside `value`; the encoder prefetch that otherwise runs
                on a side stream beside the iterations is queued behind the field in this pass): achieved algorithmic TFLOP/s vs the dense
                f16 MFMA peak / MFMAs per product; `traffic`: HBM-side bytes per launch from the PMC passes recorded
                under profiles/ (a stored constant of that profile run, labelled so); `hbm_kernels`: the lookup
  cpu_baseline  the CPU oracle (oracle/mof_oracle.py, "port") timed on this box's host cores BEFORE the timed region on a
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

# MI355X_MICROARCH.md, dense matrix peaks.  'f16x3' spends three v_mfma_f32_32x32x16_f16 per fp32-grade product, so
# its ceiling in ALGORITHMIC FLOP/s is a third of the f16 MFMA peak (a half / the whole of it for layers that run with
# two / one MFMA per product); 'f32' runs v_mfma_f32_32x32x2_f32 and is priced against the f32 matrix peak.
F16_MATRIX_PEAK_TFLOPS = 2500.0
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E, ~8 TB/s
F32_MATRIX_PEAK_TFLOPS = 157.3

DTYPE_NOTE = {
    "f16x3": "f16x3 (split-f16 MFMA: 3 f16 MFMAs per product, f32 accumulate, fp32-grade)",
    "f16x2": "f16x2 (weights as plain f16: 2 f16 MFMAs per product, f32 accumulate)",
    "f16": "f16 (plain f16 MFMA operands, f32 accumulate; at 720p 2.1e-3 px mean EPE vs the fp32 oracle: outside the 1e-3 contract)",
    "f32": "f32",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--seq", type=int, default=5)
    ap.add_argument("--workload", default="mof1080p", choices=["mof1080p", "mof4k-tile", "memflow1080p", "bof720p"],
                    help="mof1080p = BASELINE.json configs[1] (the headline, default); the others are the remaining "
                         "GPU configs, measured with the same protocol for DESIGN.md (not the driver's line)")
    ap.add_argument("--precision", default=None, choices=["f16x3", "f16x2", "f16", "mixed", "f32"],
                    help="arithmetic of the engine (vfml/cfg.py); default: the mixed plan (mof1080p: DEFAULT_MIXED_PLAN, "
                         "1e-4-grade; bof720p: BOF_F16_PLAN, BASELINE config 5's fp16 inside the 1e-3 px contract - the "
                         "line then carries plain 'f16' everywhere beside it)")
    ap.add_argument("--corr-volume", default=None, choices=["f32", "f16", "f16@1", "f16@2", "f16@3"],
                    help="storage of the correlation pyramids (vfml/cfg.py corr_volume; default f32; f16 is the opt-in "
                         "half-size volume, ~1e-4 px against the oracle at 1080p)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the separate per-launch timing pass")
    ap.add_argument("--full-output", action="store_true",
                    help="compute all 2(T-2) flows of the model output per field instead of the one the reference "
                         "path keeps ([0, shape[1]//2]); the default drops, in the last two iterations, the centre "
                         "frames that cannot influence that flow (bit-identical for it)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads for the CPU baseline (0 = this process's CPU share: affinity / cgroup quota, "
                         "capped at 16 - the per-GPU share of the pool's boxes)")
    ap.add_argument("--cpu-sample-height", type=int, default=0,
                    help="CPU-baseline sample: 0 (default) = one full-size field of the workload (about 40 s of CPU work "
                         "at 1080p on 16 threads; the line then carries an engine-vs-oracle check at the headline size); "
                         "N = one field on a 16:9 centre crop of that height, scaled to full size by the analytic FLOP ratio")
    ap.add_argument("--no-jobs", action="store_true",
                    help="skip the whole-job figures (job_300, cli_e2e: fresh child processes, about a minute more)")
    args = ap.parse_args()

    if args.workload == "mof4k-tile":
        args.height, args.width = 2160, 3840
    elif args.workload == "bof720p":
        args.height, args.width, args.seq = 720, 1280, 9
        # (the tri-frame network computes eight fields per pass of the engine: whole passes only, or the last, partial one
        # - a third of the chip - sets the average: 20 steps 134 fields/s, 24 or 32 steps 174)
        args.steps = -(-args.steps // 8) * 8
    elif args.workload == "memflow1080p":
        args.seq = 3
    if args.workload != "mof1080p":
        args.no_cpu_baseline = True
    if args.full_output:
        os.environ["VFML_FULL_OUTPUT"] = "1"      # read by processing/videoflow_processor.py at import
    # mof1080p runs the EPE-budgeted mixed plan by default (vfml/cfg.py DEFAULT_MIXED_PLAN: mean EPE <= 1e-4 px at
    # 1080p on three weight seeds and T in {3, 5}, tests/test_gpu_e2e.py) and reports the fp32-grade all-3 arithmetic
    # ('f16x3') beside it as `plans`; --precision f16x3 makes that the headline instead
    precision = args.precision or {"bof720p": "mixed", "mof1080p": "mixed"}.get(args.workload)
    if precision and args.workload != "memflow1080p":
        os.environ["VFML_PRECISION"] = precision  # read by VideoFlowCore
        if args.workload == "bof720p" and precision == "mixed":
            os.environ.setdefault("VFML_MFMA_PLAN", "bof-f16")
    if args.corr_volume:
        os.environ["VFML_CORR_VOLUME"] = args.corr_volume
    from vfml import dist as vdist, get_cfg, hip
    from vfml.runner import ClipFeeder, run_sharded
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
    tile_mode = args.workload == "mof4k-tile"
    core = proc.core if hasattr(proc, "core") else proc.core_engine

    # -- synthetic clip in host memory; one sliding job per rank: W warm-up fields, K timed, P profiled, E engine-only --
    K, Wm, T = args.steps, args.warmup, args.seq
    Pn = 0 if args.no_roofline else min(K, 4)
    En = min(K, 6)
    alt_precision = ({"mof1080p": "f16x3", "bof720p": "f16"}.get(args.workload)
                     if (precision == "mixed" and world == 1) else None)
    An = (Wm + K + Wm) if alt_precision else 0           # second plan: warm-up + timed, then the first plan's warm-up again
    per_rank = Wm + K + En + An + Pn
    # the job is one clip of world * per_rank fields; every rank holds it in host memory (same synthetic generator on
    # every rank - no input exchange) and feeds its own frame range to its GPU as its fields come up
    # (only ITS stretch of it: frame t of the synthetic clip depends on t alone, and a rank never reads another rank's
    # frames - the other entries stay None and are never uploaded)
    total_frames = world * per_rank + T - 1
    half = T // 2
    mine = [rank * per_rank + half + i for i in range(per_rank)]       # this rank's fields (full windows)
    lo_f, hi_f = mine[0] - half, mine[-1] + half + 1
    clip_np = [None] * total_frames
    clip_np[lo_f:hi_f] = synthetic_clip(hi_f - lo_f, args.height, args.width, start=lo_f)

    # -- CPU baseline (oracle) on a bounded sample of the same workload: BEFORE anything is timed (and before the
    #    engine's buffers are sized for the full frame) ----------------------------------------------------------------
    result_cpu = None
    if world == 1 and not args.no_cpu_baseline:
        result_cpu = cpu_baseline(args, proc, clip_np, mine[Wm], T)
        core.model.release_workspace()
        torch.cuda.empty_cache()

    feeder = ClipFeeder(clip_np, dev)

    def job(idxs, collect=True):
        """The CLI's job loop on this rank alone (flow_processor.py -> vfml.runner.run_sharded, world 1)."""
        return run_sharded(proc, None, idxs, tile_mode=tile_mode, rank=0, world=1, feeder=feeder, collect=collect)

    if Wm:
        job(mine[:Wm], collect=False)
    torch.cuda.synchronize()

    # -- timed region: host frames -> fields in rank-0 host memory -----------------------------------------------
    # the global item list is laid out so that the runner's contiguous shards are the ranks' own K timed fields
    timed = [r * per_rank + half + Wm + j for r in range(world) for j in range(K)]
    run_sharded(proc, None, timed, tile_mode=tile_mode, rank=rank, world=world, feeder=feeder, prepare_only=True)
    # rank 0's result array, touched before the clock starts (the job fills it at world x 41 x 16.6 MB per second)
    result_buf = np.zeros((len(timed), args.height, args.width, 2), dtype=np.float32) if rank == 0 else None
    if result_buf is not None:
        result_buf.fill(0.0)
    vdist.barrier(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run_sharded(proc, None, timed, tile_mode=tile_mode, rank=rank, world=world, feeder=feeder, out=result_buf)
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    vdist.barrier(dev)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = vdist.max_over_ranks(elapsed, dev)
    if rank == 0:
        assert out is not None and out.shape[0] == K * world and np.isfinite(out[-1]).all() and np.isfinite(out[0]).all()
    del out, result_buf

    # -- engine only: the next fields with inputs and outputs left in HBM ----------------------------------------
    eng_ms = None
    if En and not tile_mode:
        idxs = mine[Wm + K:Wm + K + En]
        feeder.ensure(max(idxs) + T)
        torch.cuda.synchronize()
        e0 = time.perf_counter()
        if hasattr(proc, "compute_optical_flow_resident_batch"):
            step = getattr(proc, "TRI_BATCH", None) or getattr(proc, "PAIR_BATCH", 1)
            for k0 in range(0, len(idxs), step):
                proc.compute_optical_flow_resident_batch(feeder.clip, idxs[k0:k0 + step])
        else:
            for i in idxs:
                proc.compute_optical_flow_resident(feeder.clip, i)
        torch.cuda.synchronize()
        eng_ms = 1000.0 * (time.perf_counter() - e0) / len(idxs)
    # -- the other arithmetic plan on the next fields of the same job (N = 1 only): its own warm-up, then K timed fields
    alt = None
    if alt_precision:
        first_plan, first_volume = dict(core.cfg.mfma_plan or {}), getattr(core.cfg, "corr_volume", "f32")
        core.cfg.precision, core.cfg.mfma_plan = alt_precision, None
        if not args.corr_volume:
            core.cfg.corr_volume = "f32"      # (the fp32-grade / plain-f16 comparison plans run on f32 volumes)
        core.model.clear_feature_cache()      # nothing of the first plan's cached frames / pyramids is of use to this one
        torch.cuda.synchronize()
        base = Wm + K + En
        job(mine[base:base + Wm], collect=False)
        torch.cuda.synchronize()
        e0 = time.perf_counter()
        out = job(mine[base + Wm:base + Wm + K])
        torch.cuda.synchronize()
        alt = time.perf_counter() - e0
        assert np.isfinite(out[-1]).all()
        del out
        core.cfg.precision = precision
        core.cfg.corr_volume = first_volume
        if precision == "mixed":
            core.cfg.mfma_plan = dict(first_plan)
        core.model.clear_feature_cache()
        job(mine[base + Wm + K:base + Wm + K + Wm], collect=False)     # back in the headline plan's steady state
        torch.cuda.synchronize()
    # -- roofline: separate pass with per-launch HIP events (never inside `value`; LAST: whatever ran after a profiled
    #    pass in one process measured 1.5 ms per field slow - the second plan used to) ------------------------------
    prof, prof_hbm, t_prof = {}, {}, None
    if Pn:
        torch.cuda.synchronize()
        e0 = time.perf_counter()
        if rank == 0:
            hip.profile_begin()
        # (the next window's encoders normally run on a side stream BESIDE the update iterations, network.py
        # prefetch_frames; here they queue behind the field so that every event-bracketed launch has the GPU to itself -
        # a launch that shares the chip with another stream's kernels says nothing about the kernel)
        dbg_before = os.environ.get("VFML_PREFETCH_DBG")
        os.environ["VFML_PREFETCH_DBG"] = "serial"
        job(mine[per_rank - Pn:], collect=False)           # (every rank does the same work; only rank 0 records)
        torch.cuda.synchronize()
        if dbg_before is None:
            del os.environ["VFML_PREFETCH_DBG"]
        else:
            os.environ["VFML_PREFETCH_DBG"] = dbg_before
        t_prof = time.perf_counter() - e0
        if rank == 0:
            prof_hbm = hip.profile_end_hbm()
            prof = hip.profile_end()

    vdist.barrier(dev)
    if rank != 0:
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return

    total_fields = K * world
    depth = core.cfg.decoder_depth
    prec = getattr(core.cfg, "precision", "f16x3")
    dtype = DTYPE_NOTE.get(prec) or ("mixed (per-layer MFMA count, f32 accumulate): " +
                                     json.dumps(getattr(core.cfg, "mfma_plan", None) or {}, sort_keys=True))
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
        "dtype": dtype,
        "data": "synthetic",
        "config": {"workload": {"mof1080p": "MOF_sintel", "mof4k-tile": "MOF_sintel --tile (6 tiles/frame)",
                                "memflow1080p": "MemFlowNet_sintel (pair path)", "bof720p": "BOF_things"}[args.workload]
                               + f" seq_len={T} {args.width}x{args.height} synthetic clip, decoder_depth={depth}, "
                                 f"seeded weights",
                   "fields_per_gpu": K, "clip_frames": len(clip_np), "parallelism": f"frames-dp{world}",
                   "cli_default_precision": "mixed (processing/videoflow_core.py: what flow_processor.py and the "
                                            "VideoFlowProcessor API run unless VFML_PRECISION says otherwise)",
                   "corr_volume": getattr(core.cfg, "corr_volume", "f32"),
                   "inputs": "uint8 frames in host memory (uploaded inside the timed region through a pinned ring)",
                   "outputs": "[H,W,2] f32 fields in rank-0 host memory (pinned D2H of field i under field i+1"
                              + ("; ranks' fields streamed to rank 0 in chunked RCCL gathers)" if world > 1 else ")"),
                   "steady_state": f"{Wm} warm-up fields of the same sliding job precede the timed fields",
                   "model_output": ("all 2(T-2) flows per field (--full-output)" if args.full_output else
                                    "the flow the reference path keeps, [0, shape[1]//2]; the last two iterations "
                                    "skip the centre frames outside its dependency cone (bit-identical for it; "
                                    "--full-output computes all 2(T-2))")},
        "rank0_local_ms_per_step": 1000.0 * t_local / K,
        "engine_ms_per_step": eng_ms,       # same job, frames already in HBM, fields left in HBM (a separate pass)
    }
    if alt is not None:
        result["plans"] = {
            prec: {"value": total_fields / elapsed, "ms_per_step": 1000.0 * elapsed / K, "dtype": dtype,
                   "epe_budget": ("mean EPE <= 1e-4 px vs the fp32 oracle at 1080p (3 weight seeds, T in {3, 5}): "
                                  "tests/test_gpu_e2e.py::test_mixed_plan_stays_within_its_budget_at_1080p"
                                  if args.workload == "mof1080p" else
                                  "mean EPE < 5e-4 px vs the fp32 oracle at 720p seq 9 (vfml/cfg.py BOF_F16_PLAN: "
                                  "tests/test_gpu_e2e.py::test_bof_720p_seq9_fp16_config)")},
            alt_precision: {"value": K / alt, "ms_per_step": 1000.0 * alt / K, "dtype": DTYPE_NOTE[alt_precision],
                            "note": "same job and protocol, the next fields of the clip, after its own warm-up"},
        }

    # -- roofline of the dominant kernel -------------------------------------------------------
    if prof:
        name, d = max(prof.items(), key=lambda kv: kv[1]["ms"])
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        split = "split" in name or "dma" in name or "tapx" in name      # the split-f16 kernels
        targs = [t.strip() for t in name[name.index("<") + 1:name.rindex(">")].split(",")]
        # the NM template argument (terms of the split product): its position per kernel family
        nm_arg = int(targs[7]) if "dma" in name else int(targs[4]) if "tapx" in name else (int(targs[5]) if split else 1)
        nm = {4: 2, 5: 1}.get(nm_arg, nm_arg)
        peak = F16_MATRIX_PEAK_TFLOPS / nm if split else F32_MATRIX_PEAK_TFLOPS
        traffic, traffic_note = hbm_traffic_from_profiles(name, args.workload)
        result["roofline"] = {
            "kernel": name, "bound": "mfma", "achieved": achieved, "peak": peak,
            "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic, "traffic_note": traffic_note,
            "peak_note": (f"algorithmic FLOP/s; peak = 2500 TFLOP/s dense f16 MFMA / {nm} MFMAs per product"
                          if split else "f32 matrix peak (v_mfma_f32_32x32x2_f32)"),
            "mfma_executed_tflops": achieved * nm,
            "launches": d["launches"], "avg_launch_us": 1000.0 * d["ms"] / d["launches"],
            "gflop_per_launch": d["flops"] / d["launches"] / 1e9,
            "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
            "pass": f"separate pass over {Pn} further fields of the same job after the timed region, per-launch HIP "
                    f"events on the launch stream, the next window's encoders queued behind the field instead of beside it "
                    f"({1000.0 * t_prof / Pn:.2f} ms per field that way)",
            "share_of_pass": d["ms"] / (1000.0 * t_prof),
            # the layers behind that kernel symbol, by shape "khxkw cin->cout" (the 1x1 motion-encoder layer streams its 672
            # input channels once and is the memory-side outlier of the four: tools/exp/r03_l2_touch/README.md)
            "by_shape": {k: {"launches": v["launches"], "avg_launch_us": 1000.0 * v["ms"] / v["launches"],
                             "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12,
                             "frac": v["flops"] / (v["ms"] * 1e-3) / 1e12 / peak}
                         for k, v in sorted(d.get("shapes", {}).items(), key=lambda kv: -kv[1]["ms"])},
            "all_variants": {k: {"ms": v["ms"], "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12,
                                 "launches": v["launches"],
                                 "shapes": {sk: [sv["launches"], round(1000.0 * sv["ms"] / sv["launches"], 1),
                                                 round(sv["flops"] / (sv["ms"] * 1e-3) / 1e12, 1)]     # launches, us, TFLOP/s
                                            for sk, sv in v.get("shapes", {}).items()}} for k, v in prof.items()},
            # the HBM-bound stages of the path beside it: algorithmic bytes / HIP-event time
            "hbm_kernels": {k: {"bound": "hbm", "achieved": v["bytes"] / (v["ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": v["bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "launches": v["launches"], "avg_launch_us": 1000.0 * v["ms"] / v["launches"],
                                "share_of_pass": v["ms"] / (1000.0 * t_prof)} for k, v in prof_hbm.items()},
        }
    if result_cpu is not None:
        result["cpu_baseline"] = result_cpu
    # -- the whole job (BASELINE config 2: a 300-frame clip), cold start included, and the drop-in CLI with its cache writer:
    #    fresh child processes, after everything above (this process's engine is released first) ------------------------
    if args.workload == "mof1080p" and world == 1 and not args.no_jobs:
        core.model.release_workspace()
        del feeder
        from vfml.runner import release_buffers
        release_buffers()
        torch.cuda.empty_cache()
        result.update(whole_jobs(precision))
    print(json.dumps(result))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def whole_jobs(precision):
    """job_300 / cli_e2e (tools/job_bench.py, one fresh process each): {"job_300": {...}, "cli_e2e": {...}}."""
    import subprocess
    env = dict(os.environ, VFML_PRECISION=precision or "mixed")
    out = {}
    t_start = time.perf_counter()
    BUDGET_S, CHILD_S = 300.0, 180.0      # (a child takes ~25 s; the line must come out even if one of them hangs)

    def run(*argv):
        if time.perf_counter() - t_start > BUDGET_S:
            return {"error": "skipped: the whole-job figures' time budget was spent"}
        try:
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "job_bench.py"), *argv], env=env,
                               capture_output=True, text=True, timeout=CHILD_S)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode == 0 and line:
                return json.loads(line[-1])
            return {"error": f"rc {p.returncode}: {(p.stderr or p.stdout)[-300:]}"}
        except subprocess.TimeoutExpired:
            return {"error": "timeout"}

    out["job_300"] = run("job300")
    out["cli_e2e"] = {"skip_lods": run("cli"), "lods": run("cli", "--lods")}
    return out


def hbm_traffic_from_profiles(kernel, workload):
    """HBM bytes per launch of `kernel` from the PMC passes over this very command (a live run cannot collect them:
    rocprofv3 has to wrap the process).  A STORED CONSTANT of the profile run named in the note, not a measurement of
    this run; None for any other workload or when no recorded profile has the kernel."""
    if workload != "mof1080p":
        return None, "no PMC passes recorded for this workload"
    pdir = os.path.join(ROOT, "profiles")
    for fn in sorted((f for f in os.listdir(pdir) if f.endswith("_hbm_traffic.json")), reverse=True):
        try:
            with open(os.path.join(pdir, fn)) as f:
                rec = json.load(f)["kernels"].get(kernel)
        except (OSError, ValueError, KeyError):
            rec = None
        if rec:
            return rec["hbm_bytes_per_launch"], (
                "STORED constant, not measured by this run: bytes per launch, mean over %d launches, (2 * FETCH_SIZE + "
                "WRITE_SIZE) * 1024 from separate rocprofv3 --pmc passes over bench.py (profiles/%s)" % (rec["launches"], fn))
    return None, "kernel not in any profiles/*_hbm_traffic.json"


def host_cpu_share(cap=16):
    from vfml.dist import host_cpu_share as share
    return share(cap)


def cpu_baseline(args, proc, clip_np, field_idx, T):
    """The CPU oracle (fp32 PyTorch restatement) on the host cores, on a bounded sample of the
    workload: one field of the same clip/window/weights, by default on a 1440x816 centre crop, scaled
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
    dev = next(proc.core.model.parameters()).device
    eng, _ = proc.core.model.forward_u8(torch.from_numpy(win).to(dev), return_lowres=False)
    got = eng[0, eng.shape[1] // 2].permute(1, 2, 0).cpu()
    if full:
        ratio = 1.0
        sample = f"1 full {W}x{H} seq{T} field (field {field_idx} of the clip, {dt:.1f} s)"
    else:
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
