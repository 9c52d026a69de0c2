"""Asynchronous flow-cache writer.

The reference writes one `np.savez_compressed` per frame on the compute thread
(storage/cache_manager.py:247-262 called from flow_processor.py:976 / :1470): zlib on one core at
16.6 MB per 1080p field is far slower than the MI355X engine.  Here the same FlowCacheManager calls —
same file names, same members (`flow`, `frame_idx`, `shape`, `dtype`), same optional LOD files — run on
a small thread pool (zlib releases the GIL), so the cache a later run or the reference's tools read is
unchanged while the writer keeps up with tens of fields per second.
"""
from concurrent.futures import ThreadPoolExecutor

from .cache_manager import FlowCacheManager


class AsyncFlowCacheWriter:
    def __init__(self, cache_dir, save_format='npz', workers=8, num_lods=0, manager=None):
        self.cache_dir, self.save_format, self.num_lods = cache_dir, save_format, num_lods
        self.manager = manager or FlowCacheManager()
        self._pool = ThreadPoolExecutor(max_workers=max(1, workers), thread_name_prefix="flowcache")
        self._pending = []

    def _write(self, flow, frame_idx):
        self.manager.save_flow_to_cache(flow, self.cache_dir, frame_idx, self.save_format)
        if self.num_lods and self.save_format in ('npz', 'both'):
            self.manager.save_flow_lods(self.manager.lod_generator.generate_lods(flow, self.num_lods),
                                        self.cache_dir, frame_idx)
        return frame_idx

    def submit(self, flow, frame_idx):
        """flow: numpy [H,W,2] float32 (not modified; the caller must not overwrite it before close())."""
        self._pending.append(self._pool.submit(self._write, flow, frame_idx))

    def close(self):
        """Wait for every write; re-raises the first failure.  Returns the frame indices written."""
        try:
            return sorted(f.result() for f in self._pending)
        finally:
            self._pool.shutdown(wait=True)
            self._pending = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
