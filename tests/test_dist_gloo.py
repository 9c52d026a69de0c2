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


def _cli_worker(rank, world, port, workdir, argv, q):
    """One rank of `torch.distributed.run ... flow_processor.py` on CPU with a stand-in model."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "video-flow-ml_amd"), root):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    os.chdir(workdir)
    import flow_processor
    import processing.videoflow_core as core_mod

    class Net(FakeModel):
        def load_state_dict(self, sd, strict=True):
            return None

    core_mod.build_network = lambda cfg: Net()
    core_mod.load_checked = lambda model, state, what: model.load_state_dict(state)
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        rc = flow_processor.main(argv)
    q.put((rank, rc, out.getvalue()))


@pytest.mark.parametrize("tile", [False, True])
def test_gloo_cli_ranks_write_one_complete_cache(tmp_path, tile):
    """`flow_processor.py` under three gloo ranks.  Whole frames: EVERY rank writes the cache files of its own fields (no
    gather; reference storage/cache_manager.py:247-262 is one file per frame, so one node's ranks fill one directory);
    tiles: rank 0 pastes and writes.  Either way the directory is what the reference's completeness check accepts
    (storage/cache_manager.py:192-230) and holds the serial loop's fields, LOD files included."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "video-flow-ml_amd"))
    from storage import FlowCacheManager
    from vfml import get_cfg
    from vfml.synth import synthetic_clip
    from vfml.weights import write_seeded_checkpoint
    write_seeded_checkpoint(str(tmp_path), get_cfg(), seed=0)
    world, n = 3, 8
    W, H = (2600, 24) if tile else (64, 48)          # 2600 px: three of the reference's 1280-px tile columns
    argv = ["--input", f"synthetic:{W}x{H}x{n}", "--output", str(tmp_path / "out"), "--device", "cpu", "--sequence-length",
            "3", "--interactive"] + (["--tile"] if tile else [])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cli_worker, args=(r, world, port, str(tmp_path), argv, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [rc for _, rc, _ in res] == [0] * world, res
    assert ["Flow cache written" in o for _, _, o in res] == [True, False, False]         # rank 0 alone reports
    mgr = FlowCacheManager()
    tag = "_tile" if tile else ""
    cache = tmp_path / "out" / f"synthetic_{W}x{H}x{n}_flow_cache_videoflow_mof_sintel_standard_seq3_start0_frames{n}{tag}"
    assert mgr.check_cache_exists(str(cache), n) == (True, "npz", [])
    assert mgr.check_flow_lods_exist(str(cache), n, 5)
    assert len(os.listdir(cache)) == n * 6
    with contextlib.redirect_stdout(io.StringIO()):
        from processing.videoflow_processor import VideoFlowProcessor
        serial = VideoFlowProcessor("cpu", tile_mode=tile, sequence_length=3)
    serial.core.model = FakeModel()
    frames = synthetic_clip(n, H, W)
    for i in range(n):
        z = np.load(cache / f"flow_frame_{i:06d}.npz")
        ref = serial.compute_optical_flow_tiled(list(frames), i)
        assert int(z["frame_idx"]) == i and np.array_equal(z["flow"], ref), i
        lods = mgr.lod_generator.generate_lods(ref, 5)
        for k in (1, 4):
            assert np.array_equal(np.load(cache / f"flow_frame_{i:06d}_lod{k}.npz")["flow"], lods[k])


def test_tiled_items_come_in_frame_major_blocks():
    """Tiled jobs: tile-major inside blocks of a few frames (vfml.runner.tile_items) - a frame is complete, and leaves rank
    0's memory for the writer, once its block's last tile has passed; the whole job is never held."""
    from vfml.runner import run_sharded, tile_items
    assert tile_items([0, 1, 2, 3, 4], 2, block=2) == [(0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (3, 0), (2, 1), (3, 1), (4, 0), (4, 1)]
    assert tile_items([5, 6], 1) == [(5, 0), (6, 0)]
    import vfml.runner as rn
    proc = _proc(True)
    clip = proc.upload_clip(_clip(7))
    order, held = [], []
    old = rn.TILE_BLOCK_FRAMES
    rn.TILE_BLOCK_FRAMES = 3
    try:
        out = run_sharded(proc, clip, range(7), tile_mode=True, collect=False, chunk=2,
                          on_field=lambda k, f, lods: order.append(k))
    finally:
        rn.TILE_BLOCK_FRAMES = old
    assert out is None and sorted(order) == list(range(7))
    # frames of block b are all delivered before any frame of block b + 2 (chunks of 2 items straddle one boundary at most)
    assert max(order[:3]) <= 2 and set(order[:6]) == set(range(6))


def test_feeder_refuses_windows_below_what_a_shard_skipped():
    """A ClipFeeder only uploads what its job reaches (skip_to): a later job on the same feeder that needs earlier frames
    raises instead of reading device memory nobody wrote; after reset() everything is available again."""
    from vfml.runner import ClipFeeder, run_sharded
    proc = _proc(False)
    frames = _clip(16)
    feeder = ClipFeeder(frames, "cpu")
    late = run_sharded(proc, None, [12, 13], feeder=feeder)
    assert feeder.lo == 12 - proc.sequence_length and feeder.next >= 14
    ser = _proc(False)
    assert np.array_equal(late[1], ser.compute_optical_flow(frames, 13))
    with pytest.raises(RuntimeError, match="skipped by an earlier shard"):
        run_sharded(proc, None, [2, 3], feeder=feeder)
    feeder.reset(frames)
    early = run_sharded(proc, None, [2, 3], feeder=feeder)
    assert np.array_equal(early[0], ser.compute_optical_flow(frames, 2))
    assert feeder.clip._vfml_frame_maxima[5] == float(frames[5].max())      # computed when asked for


class _SleepyProc:
    """Stand-in processor for the runner alone: a field costs `ms` of (simulated) device time and is a constant plane."""
    sequence_length = 5

    def __init__(self, ms, rank):
        self.ms, self.rank, self.plane = ms, rank, None

    def compute_optical_flow_resident(self, clip, frame_idx, tile=None):
        import time
        time.sleep(self.ms / 1e3)
        if self.plane is None:
            self.plane = torch.empty((clip.shape[1], clip.shape[2], 2), dtype=torch.float32)
        self.plane[0, 0, 0] = self.plane[-1, -1, 1] = float(frame_idx)       # (the runner copies it into its send buffer)
        return self.plane


def _timing_worker(rank, world, port, q, H, W, per_rank, ms):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "video-flow-ml_amd"), root):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), VFML_RUNNER_TIMING="1")
    torch.set_num_threads(1)
    from vfml import dist as vdist
    from vfml.runner import run_sharded
    vdist.init_distributed(backend="gloo")
    clip = torch.zeros((world * per_rank, H, W, 3), dtype=torch.uint8)
    seen = []
    with contextlib.redirect_stdout(io.StringIO()):
        run_sharded(_SleepyProc(ms, rank), clip, range(world * per_rank), rank=rank, world=world, collect=False,
                    on_field=(lambda k, f, lods: seen.append((k, float(f[0, 0, 0]), float(f[-1, -1, 1])))) if rank == 0 else None)
    vdist.barrier()
    if rank == 0:
        q.put((sorted(seen), run_sharded.last_trace))
    dist.destroy_process_group()


def _world8_once():
    world, per_rank, ms, H, W = 8, 6, 30.0, 1080, 1920
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_timing_worker, args=(r, world, port, q, H, W, per_rank, ms)) for r in range(world)]
    for p in procs:
        p.start()
    seen, trace = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [k for k, _, _ in seen] == list(range(world * per_rank))
    assert all(a == float(k) and b == float(k) for k, a, b in seen)          # every field whole, from the right rank
    chunk_ms = 2 * ms                                                        # default chunk at 8 ranks: two fields
    hand = sorted(1e3 * t for t in trace["handoff"])
    unp = sorted(1e3 * t for t in trace["unpack"])
    print(f"rank 0 per chunk: driver thread {hand[len(hand) // 2]:.2f} ms of hand-off beside {chunk_ms:.0f} ms of compute, "
          f"unpack thread {unp[len(unp) // 2]:.1f} ms for {world * 2 * 16.6:.0f} MB")
    return hand[len(hand) // 2], unp[len(unp) // 2], chunk_ms


def test_world8_rank0_host_work_stays_below_compute_time():
    """Eight gloo ranks, 1080p-sized fields (16.6 MB each), a stand-in model that takes 30 ms per field: the thread of rank
    0 that drives the device spends per chunk far less than the chunk's compute time on bookkeeping (it neither waits for
    copies nor touches field bytes - the unpack thread does), and the unpack thread moves the eight ranks' fields of a
    chunk in less than the chunk's compute time.  (The collective itself is gloo over loopback here, RCCL over xGMI on
    the GPUs: its wait is not host work of the runner and is reported apart, as is the stand-in's own "device" time.)"""
    # (a timing statement on eight processes sharing this machine's cores: one retry before it counts as a failure)
    for attempt in range(2):
        hand, unp, chunk_ms = _world8_once()
        if hand < 0.25 * chunk_ms and unp < chunk_ms:
            return
    assert hand < 0.25 * chunk_ms and unp < chunk_ms, (hand, unp, chunk_ms)



def test_feeder_over_a_clip_of_which_this_process_holds_one_stretch():
    """bench.py --gpus N: a rank generates only its own stretch of the synthetic clip; the other entries of its frame list
    are None, never uploaded, and the read-ahead stops in front of them - but a window that NEEDS one raises."""
    from vfml.runner import ClipFeeder, run_sharded
    proc = _proc(False)
    frames = _clip(16)
    mine = [None] * 16
    mine[4:12] = frames[4:12]                   # fields 6..9 have their whole 5-frame windows in 4..11
    feeder = ClipFeeder(mine, "cpu")
    got = run_sharded(proc, None, [6, 7, 8, 9], feeder=feeder)
    ser = _proc(False)
    for k, i in enumerate((6, 7, 8, 9)):
        assert np.array_equal(got[k], ser.compute_optical_flow(frames, i)), i
    assert feeder.lo == 4 and feeder.next == 12                             # nothing of the other stretches was touched
    with pytest.raises(RuntimeError, match="not held by this process"):
        run_sharded(proc, None, [10], feeder=feeder)                          # its window reaches frame 12
