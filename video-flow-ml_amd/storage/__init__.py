"""storage — flow cache files and naming (mirror of the reference's storage/ package)."""
from .cache_manager import FlowCacheManager, FlowFileHandler, LODGenerator

__all__ = ["FlowCacheManager", "FlowFileHandler", "LODGenerator"]
