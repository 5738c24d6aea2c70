"""visualcla -- MI355X-native drop-in for the reference's `visualcla` package (models/visualcla/__init__.py:1-8):
same exported names, arithmetic executed by libvisualcla_hip.so (hand-written gfx950 kernels)."""
from .configuration_visualcla import VisualCLAConfig, visualcla_7b_config
from .modeling_visualcla import VisualCLAModel
from .processing_visualcla import VisualCLAProcessor
from .modeling_utils import (
    DEFAULT_GENERATION_CONFIG,
    chat,
    chat_in_stream,
    get_model_and_processor,
    get_model_and_tokenizer_and_processor,
    hijack_samplers,
)

__all__ = [
    "VisualCLAModel", "VisualCLAConfig", "VisualCLAProcessor", "get_model_and_tokenizer_and_processor",
    "get_model_and_processor", "chat", "chat_in_stream", "hijack_samplers", "DEFAULT_GENERATION_CONFIG",
    "visualcla_7b_config",
]
