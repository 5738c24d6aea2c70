"""VisualCLA composite configuration (same fields and JSON layout as the reference's
models/visualcla/configuration_visualcla.py:11-39, so existing `config.json` files load unchanged)."""
from __future__ import annotations

from typing import Dict, Optional, Union

from transformers.configuration_utils import PretrainedConfig

try:  # transformers 5.x renamed the base class; both spellings are accepted
    from transformers.configuration_utils import PreTrainedConfig as _Base  # type: ignore
except Exception:  # pragma: no cover
    _Base = PretrainedConfig


def _as_dict(cfg) -> Optional[dict]:
    if cfg is None:
        return None
    return cfg.to_dict() if hasattr(cfg, "to_dict") else dict(cfg)


class VisualCLAConfig(_Base):
    model_type = "visualcla"
    is_composition = True

    def __init__(self, text_config: Union[dict, "PretrainedConfig", None] = None,
                 vision_config: Union[dict, "PretrainedConfig", None] = None,
                 initializer_range: float = 0.02, layer_norm_eps: float = 1e-12,
                 use_visual_resampler: bool = False, visual_resampler_config: Optional[Dict] = None, **kwargs):
        super().__init__(**kwargs)
        self.text_config = _as_dict(text_config)
        self.vision_config = _as_dict(vision_config)
        self.initializer_range = initializer_range
        self.layer_norm_eps = layer_norm_eps
        self.use_visual_resampler = use_visual_resampler
        self.visual_resampler_config = _as_dict(visual_resampler_config)


def visualcla_7b_config() -> VisualCLAConfig:
    """VisualCLA-7B geometry: CLIP-ViT-L/14 @224, 6-layer resampler with 64 queries,
    Chinese-Alpaca-Plus-7B LLaMA (vocab 49954 + 4 image tokens)."""
    return VisualCLAConfig(
        text_config=dict(vocab_size=49958, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                         num_attention_heads=32, num_key_value_heads=32, rms_norm_eps=1e-6,
                         max_position_embeddings=2048, rope_theta=10000.0, hidden_act="silu",
                         bos_token_id=1, eos_token_id=2, tie_word_embeddings=False),
        vision_config=dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                           image_size=224, patch_size=14, num_channels=3, layer_norm_eps=1e-5,
                           hidden_act="quick_gelu"),
        use_visual_resampler=True,
        visual_resampler_config=dict(hidden_size=1024, num_hidden_layers=6, num_attention_heads=16,
                                     intermediate_size=4096, num_query_tokens=64, layer_norm_eps=1e-12,
                                     hidden_act="gelu"),
    )
