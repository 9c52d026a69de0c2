"""The N>1 path on CPU: two or three gloo ranks shard (frame, tile) work items (uneven shards included), compute
with a stand-in model through the real processor plumbing, stream their finished fields to rank 0 in chunked
gathers — and must reproduce the serial loop exactly, in the returned array and through `on_field`."""
import contextlib
import io
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class FakeModel(torch.nn.Module):
    def forward(self, x, _):
        B, T, C, H, W = x.shape
        base = x[:, :, :2].mean(dim=1, keepdim=True) + x[:, T // 2:T // 2 + 1, 1:3]
        flows = torch.cat([base * (k + 1) for k in range(2 * (T - 2))], dim=1)
        return flows.view(B, 2 * (T - 2), 2, H, W), None


def _clip(n=7, h=40, w=56):
    rng = np.random.default_rng(3)
    return [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(n)]


def _proc(tile_mode):
    from processing.videoflow_processor import VideoFlowProcessor
    with contextlib.redirect_stdout(io.StringIO()):
        p = VideoFlowProcessor("cpu", tile_mode=tile_mode, sequence_length=5)
    p.core.model = FakeModel()
    grid = p.calculate_tile_grid
    p.calculate_tile_grid = lambda w, h, tile_size=1280: grid(w, h, 24)   # ragged 24-px tiles, same code path
    return p


def _worker(rank, world, port, tile_mode, q, chunk=None, fed=False):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "video-flow-ml_amd"), root):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from vfml import dist as vdist
    from vfml.runner import run_sharded
    r, _, w = vdist.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    from vfml.runner import ClipFeeder
    proc = _proc(tile_mode)
    seen = {}

    def on_field(k, field, lods):
        assert k not in seen and lods is None
        seen[k] = field.copy()

    if fed:         # frames uploaded as the job advances; fields only through the callback
        feeder = ClipFeeder(_clip(), "cpu")
        out = run_sharded(proc, None, range(7), tile_mode=tile_mode, rank=rank, world=world, chunk=chunk,
                          feeder=feeder, on_field=on_field if rank == 0 else None, collect=False)
        assert out is None
        if rank == 0:
            out = np.stack([seen[k] for k in range(7)])
    else:
        clip = proc.upload_clip(_clip())
        # (buffers + one collective of the job's shape up front, as bench.py does before its timed region)
        assert run_sharded(proc, clip, range(clip.shape[0]), tile_mode=tile_mode, rank=rank, world=world, chunk=chunk,
                           prepare_only=True) is None
        out = run_sharded(proc, clip, range(clip.shape[0]), tile_mode=tile_mode, rank=rank, world=world, chunk=chunk,
                          on_field=on_field if rank == 0 else None)
        if rank == 0:
            assert sorted(seen) == list(range(7)) and all(np.array_equal(seen[k], out[k]) for k in seen)
    t = vdist.max_over_ranks(float(rank + 1), torch.device("cpu"))
    vdist.barrier()
    if rank == 0:
        q.put((out, t))
    elif not fed:
        assert out is None
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,tile_mode,chunk,fed", [(2, False, None, False), (2, True, None, False),
                                                       (3, False, 1, True), (3, True, 3, False), (3, True, 2, True)])
def test_gloo_job_equals_serial_loop(world, tile_mode, chunk, fed):
    """7 frames (x 6 ragged tiles) over 2 or 3 ranks: shards of 4+3, 3+2+2, 14+14+14 items; chunk sizes that do and
    do not divide a shard (the last gather of a rank then carries unused slots)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, tile_mode, q, chunk, fed)) for r in range(world)]
    for p in procs:
        p.start()
    out, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == float(world)                                  # MAX over ranks
    serial = _proc(tile_mode)
    frames = _clip()
    for i in range(len(frames)):
        ref = serial.compute_optical_flow_tiled(list(frames), i)   # the reference's serial path
        assert np.array_equal(out[i], ref), i


def test_shard_bounds_cover_everything_contiguously():
    from vfml.dist import shard_bounds, work_items
    for n in (0, 1, 5, 8, 300, 1801):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    assert work_items([4, 5], 3) == [(4, 0), (4, 1), (4, 2), (5, 0), (5, 1), (5, 2)]


def test_single_rank_runner_without_process_group():
    from vfml.runner import run_sharded
    proc = _proc(True)
    clip = proc.upload_clip(_clip(5))
    got = []
    out = run_sharded(proc, clip, [1, 3], tile_mode=True, on_field=lambda k, f, lods: got.append((k, f.copy())))
    ser = _proc(True)
    assert np.array_equal(out[0], ser.compute_optical_flow_tiled(_clip(5), 1))
    assert np.array_equal(out[1], ser.compute_optical_flow_tiled(_clip(5), 3))
    assert [k for k, _ in got] == [0, 1] and np.array_equal(got[1][1], out[1])
