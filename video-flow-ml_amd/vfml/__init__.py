"""vfml — MI355X-native multi-frame optical-flow engine (hand-written HIP kernels behind a C ABI).

What the reference imports from its VideoFlow submodule (processing/videoflow_core.py:28-30) is
provided here instead:  build_network, InputPadder, get_cfg.
"""
from .cfg import get_cfg
from .padder import InputPadder
from .network import build_network, MOFNetHIP

__all__ = ["get_cfg", "InputPadder", "build_network", "MOFNetHIP"]
