"""Encoding module - flow fields as 8-bit images (API of the reference's encoding/__init__.py)."""
from .flow_encoders import (FlowEncoder, FlowEncoderFactory, GamedevFlowEncoder, MotionVectorsRG8FlowEncoder,
                            MotionVectorsRGB8FlowEncoder, decode_motion_vectors, encode_flow, encode_motion_vectors)

__all__ = ['FlowEncoder', 'GamedevFlowEncoder', 'MotionVectorsRG8FlowEncoder', 'MotionVectorsRGB8FlowEncoder',
           'FlowEncoderFactory', 'encode_flow', 'encode_motion_vectors', 'decode_motion_vectors']
