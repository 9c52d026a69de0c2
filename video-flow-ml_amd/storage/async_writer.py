"""Asynchronous flow-cache writer.

The reference writes one `np.savez_compressed` per frame on the compute thread
(storage/cache_manager.py:247-262 called from flow_processor.py:976 / :1470): zlib on one core at
16.6 MB per 1080p field is far slower than the MI355X engine.  Here the same FlowCacheManager calls —
same file names, same members (`flow`, `frame_idx`, `shape`, `dtype`), same optional LOD files — run on
a small thread pool (zlib releases the GIL), so the cache a later run or the reference's tools read is
unchanged while the writer keeps up with tens of fields per second.
"""
import threading
from concurrent.futures import ThreadPoolExecutor

from .cache_manager import FlowCacheManager


class AsyncFlowCacheWriter:
    """max_pending bounds the fields queued or being written (default 2 per worker): `submit` blocks when the
    compression threads fall behind the GPU, so host memory does not grow with the length of the job, and a
    failed write is re-raised by the next `submit` instead of at `close()`."""

    def __init__(self, cache_dir, save_format='npz', workers=8, num_lods=0, manager=None, max_pending=None):
        self.cache_dir, self.save_format, self.num_lods = cache_dir, save_format, num_lods
        self.manager = manager or FlowCacheManager()
        workers = max(1, workers)
        self._pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="flowcache")
        self._slots = threading.BoundedSemaphore(max_pending or 2 * workers)
        self._lock = threading.Lock()
        self._written, self._error = [], None

    def _write(self, flow, frame_idx, lods):
        try:
            self.manager.save_flow_to_cache(flow, self.cache_dir, frame_idx, self.save_format)
            if self.num_lods and self.save_format in ('npz', 'both'):
                if lods is None:
                    lods = self.manager.lod_generator.generate_lods(flow, self.num_lods)
                self.manager.save_flow_lods(lods, self.cache_dir, frame_idx)
            with self._lock:
                self._written.append(frame_idx)
        except BaseException as e:     # surfaced by the next submit() / close()
            with self._lock:
                if self._error is None:
                    self._error = e
        finally:
            self._slots.release()

    def _raise_pending_error(self):
        with self._lock:
            err, self._error = self._error, None
        if err is not None:
            raise err

    def submit(self, flow, frame_idx, lods=None):
        """flow: numpy [H,W,2] float32, owned by the writer from here on (dropped once written).
        lods: optional precomputed LOD pyramid [flow, lod1, ...] (the GPU's, vfml_flow_lod) - generated on a
        worker thread otherwise.  Blocks while `max_pending` fields are in flight."""
        self._raise_pending_error()
        self._slots.acquire()
        try:
            self._pool.submit(self._write, flow, frame_idx, lods)
        except BaseException:
            self._slots.release()
            raise

    def close(self):
        """Wait for every write; re-raises the first failure.  Returns the frame indices written."""
        self._pool.shutdown(wait=True)
        self._raise_pending_error()
        with self._lock:
            done, self._written = sorted(self._written), []
        return done

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
