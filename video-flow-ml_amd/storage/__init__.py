"""storage — flow cache files and naming (mirror of the reference's storage/ package)."""
from .cache_manager import FlowCacheManager, FlowFileHandler, LODGenerator
from .async_writer import AsyncFlowCacheWriter

__all__ = ["FlowCacheManager", "FlowFileHandler", "LODGenerator", "AsyncFlowCacheWriter"]
