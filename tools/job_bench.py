"""Whole-job figures of BASELINE config 2 (MOF_sintel seq 5, 1920x1080, 300-frame synthetic clip) in a FRESH process, for
bench.py's `job_300` and `cli_e2e` objects (GPU only).  One JSON line on stdout.

    python tools/job_bench.py job300            the reference's frame loop (flow_processor.py:959-976: one field per frame of
                                                the clip, clip-edge windows included) through vfml.runner.run_sharded, host
                                                memory to host memory, from a fresh engine: cold start INCLUDED (workspace and
                                                pyramid allocations, the eager and capture passes of the iteration graph, first
                                                windows that encode every frame), model load EXCLUDED and reported beside it
    python tools/job_bench.py cli [--lods]      the same job through flow_processor.main with the .npz cache writer on
                                                (--skip-lods unless --lods): the CLI's own timing lines + the wall time of main()

Environment: VFML_PRECISION as bench.py sets it (default here: mixed); frames / size via --frames / --size."""
import argparse
import contextlib
import io
import json
import os
import re
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd"))
os.environ.setdefault("VFML_PRECISION", "mixed")

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["job300", "cli"])
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--seq", type=int, default=5)
    ap.add_argument("--lods", action="store_true")
    args = ap.parse_args()
    W, H = (int(v) for v in args.size.split("x"))
    from vfml import get_cfg
    from vfml.weights import write_seeded_checkpoint
    work = tempfile.mkdtemp(prefix="vfml_job_", dir=os.environ.get("TMPDIR"))
    write_seeded_checkpoint(work, get_cfg(), seed=0)
    os.chdir(work)
    torch.cuda.init()
    torch.zeros(1, device="cuda")
    torch.cuda.synchronize()
    res = {"mode": args.mode, "frames": args.frames, "size": args.size, "seq": args.seq,
           "precision": os.environ.get("VFML_PRECISION")}
    if args.mode == "job300":
        from processing.videoflow_processor import VideoFlowProcessor
        from vfml.runner import ClipFeeder, run_sharded
        from vfml.synth import synthetic_clip
        frames = synthetic_clip(args.frames, H, W)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            proc = VideoFlowProcessor("cuda", sequence_length=args.seq)
            proc.load_model()
        torch.cuda.synchronize()
        res["load_model_s"] = time.perf_counter() - t0
        done = []
        t0 = time.perf_counter()
        feeder = ClipFeeder(frames, "cuda")
        out = run_sharded(proc, None, range(args.frames), feeder=feeder,
                          on_field=lambda k, f, lods: done.append(time.perf_counter()))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert out.shape == (args.frames, H, W, 2) and np.isfinite(out[0]).all() and np.isfinite(out[-1]).all()
        done = [t - t0 for t in done]
        n = args.frames
        tail = done[n // 2:]
        steady = (tail[-1] - tail[0]) / (len(tail) - 1)
        res.update(seconds=dt, fields_per_s=n / dt, first_field_s=done[0], first_8_fields_s=done[min(7, n - 1)],
                   steady_ms_per_field=1e3 * steady, cold_start_s=dt - n * steady,
                   reserved_gib=torch.cuda.memory_reserved() / 2 ** 30,
                   note="frames resident in host memory + model loaded -> last field in host memory; fresh process, fresh "
                        "engine; cold_start_s = seconds - frames x steady-state time per field (second half of the job)")
    else:
        import flow_processor
        argv = ["--input", f"synthetic:{W}x{H}x{args.frames}", "--sequence-length", str(args.seq), "--output",
                os.path.join(work, "out"), "--device", "cuda", "--interactive"] + ([] if args.lods else ["--skip-lods"])
        buf = io.StringIO()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(buf):
            rc = flow_processor.main(argv)
        wall = time.perf_counter() - t0
        text = buf.getvalue()
        m1 = re.search(r"flow fields .* in ([0-9.]+) s = ([0-9.]+) fields/s", text)
        m2 = re.search(r"\(([0-9.]+) fields/s end to end", text)
        cache = [d for d in os.listdir(os.path.join(work, "out")) if "flow_cache" in d]
        files = os.listdir(os.path.join(work, "out", cache[0])) if cache else []
        size = sum(os.path.getsize(os.path.join(work, "out", cache[0], f)) for f in files) if cache else 0
        res.update(rc=rc, lods=bool(args.lods), main_wall_s=wall,
                   compute_loop_s=float(m1.group(1)) if m1 else None, compute_loop_fields_per_s=float(m2 and m1.group(2)) if m1 else None,
                   end_to_end_fields_per_s=float(m2.group(1)) if m2 else None, cache_files=len(files), cache_gb=size / 1e9,
                   npz_deflate=os.environ.get("VFML_NPZ_DEFLATE", "huffman"),
                   note="flow_processor.main: synthetic clip generated + model loaded inside main_wall_s; compute_loop = the "
                        "frame loop with the cache writer attached (its back-pressure included); end_to_end = until the last "
                        "cache file is closed")
        import shutil
        shutil.rmtree(os.path.join(work, "out"), ignore_errors=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
