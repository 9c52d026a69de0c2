"""VideoFlowInference — backward-compatible facade over VideoFlowProcessor.

API mirror of reference processing/flow_inference.py:24-157: same constructor, same attribute
mirrors (`device … variant`, `model`, `cfg` filled in by load_model :66-72), same method set;
every call is forwarded to the wrapped processor.
"""
from .videoflow_processor import VideoFlowProcessor

_FORWARDED = (
    "calculate_tile_grid", "extract_tile", "prepare_frame_sequence", "compute_optical_flow",
    "compute_optical_flow_with_progress", "compute_optical_flow_tiled", "is_model_loaded",
    "get_memory_usage", "validate_frames",
)


class VideoFlowInference:
    def __init__(self, device, fast_mode=False, tile_mode=False, sequence_length=5,
                 dataset='sintel', architecture='mof', variant='standard'):
        self._processor = VideoFlowProcessor(device, fast_mode, tile_mode, sequence_length,
                                             dataset, architecture, variant)
        self.device, self.fast_mode, self.tile_mode = device, fast_mode, tile_mode
        self.sequence_length, self.dataset = sequence_length, dataset
        self.architecture, self.variant = architecture, variant
        self.model = None
        self.cfg = None

    def __getattr__(self, name):
        # only reached for names not found normally; forwards the processor's public methods
        if name in _FORWARDED:
            return getattr(self.__dict__["_processor"], name)
        raise AttributeError(f"{type(self).__name__!s} has no attribute {name!r}")

    def load_model(self):
        self._processor.load_model()
        self.model = self._processor.core.model
        self.cfg = self._processor.core.cfg

    def get_model_info(self):
        info = self._processor.get_model_info()
        if info["status"] == "loaded":
            info["compatibility_layer"] = "VideoFlowInference"
        return info

    def get_core_engine(self):
        return self._processor.core

    def get_processor(self):
        return self._processor

    def set_tile_mode(self, enabled):
        self.tile_mode = enabled
        self._processor.set_tile_mode(enabled)

    def set_sequence_length(self, length):
        self.sequence_length = length
        self._processor.set_sequence_length(length)
