"""MemFlowInference — backward-compatible facade over MemFlowProcessor (API mirror of reference
processing/memflow_inference.py:20-109: same constructor, attribute mirrors, forwarded methods)."""
from .memflow_processor import MemFlowProcessor

_FORWARDED = ("prepare_frame_sequence", "compute_optical_flow", "compute_optical_flow_with_progress",
              "calculate_tile_grid", "extract_tile", "compute_optical_flow_tiled", "get_memory_usage", "cleanup")


class MemFlowInference:
    def __init__(self, device='cuda', model_path='MemFlow_ckpt/MemFlowNet_sintel.pth', stage='sintel',
                 sequence_length=3):
        self.device, self.model_path, self.stage, self.sequence_length = device, model_path, stage, sequence_length
        self.processor = MemFlowProcessor(device=device, model_path=model_path, stage=stage,
                                          sequence_length=sequence_length)
        self.model = None
        self.cfg = None

    def __getattr__(self, name):
        if name in _FORWARDED:
            return getattr(self.__dict__["processor"], name)
        raise AttributeError(f"{type(self).__name__!s} has no attribute {name!r}")

    def load_model(self):
        self.processor.load_model()
        self.model = self.processor.core_engine.model
        self.cfg = self.processor.core_engine.cfg

    def get_processor(self):
        return self.processor

    def get_core_engine(self):
        return self.processor.get_core_engine()
