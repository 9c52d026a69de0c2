"""processing — inference API of the drop-in (mirror of the reference's processing/ package,
VideoFlow half; the MemFlow half is out of scope for this round)."""
from .flow_inference import VideoFlowInference
from .videoflow_core import VideoFlowCore
from .videoflow_processor import VideoFlowProcessor

__all__ = ["VideoFlowInference", "VideoFlowCore", "VideoFlowProcessor"]
