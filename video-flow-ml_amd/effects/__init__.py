"""Effects that consume flow fields (reference effects/__init__.py)."""
from .taa_processor import TAAComparisonProcessor, TAAProcessor, apply_taa_effect

__all__ = ['TAAProcessor', 'TAAComparisonProcessor', 'apply_taa_effect']
