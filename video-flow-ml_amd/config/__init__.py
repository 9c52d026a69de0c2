"""config — device selection (mirror of the reference's config/ package)."""
from .device_manager import DeviceManager

__all__ = ["DeviceManager"]
