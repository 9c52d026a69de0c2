"""processing — inference API of the drop-in (mirror of the reference's processing/ package)."""
from .flow_inference import VideoFlowInference
from .videoflow_core import VideoFlowCore
from .videoflow_processor import VideoFlowProcessor
from .memflow_core import MemFlowCore
from .memflow_processor import MemFlowProcessor
from .memflow_inference import MemFlowInference

__all__ = ["VideoFlowInference", "VideoFlowCore", "VideoFlowProcessor",
           "MemFlowInference", "MemFlowCore", "MemFlowProcessor"]
