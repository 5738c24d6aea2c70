"""Shared test helpers: oracle config -> VisualCLAConfig, HIP model construction from oracle weights."""
from __future__ import annotations

from types import SimpleNamespace

import torch

from oracle import visualcla_oracle as O


def to_vcla_config(cfg: O.OracleCfg):
    from visualcla import VisualCLAConfig
    v, r, t = cfg.vision, cfg.resampler, cfg.text
    return VisualCLAConfig(
        text_config=dict(vocab_size=t.vocab_size, hidden_size=t.hidden_size, intermediate_size=t.intermediate_size,
                         num_hidden_layers=t.num_hidden_layers, num_attention_heads=t.num_attention_heads,
                         num_key_value_heads=t.num_attention_heads, rms_norm_eps=t.rms_norm_eps,
                         max_position_embeddings=t.max_position_embeddings, rope_theta=t.rope_theta),
        vision_config=dict(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size,
                           num_hidden_layers=v.num_hidden_layers, num_attention_heads=v.num_attention_heads,
                           image_size=v.image_size, patch_size=v.patch_size, num_channels=v.num_channels,
                           layer_norm_eps=v.layer_norm_eps, hidden_act=v.hidden_act),
        use_visual_resampler=True,
        visual_resampler_config=dict(hidden_size=r.hidden_size, num_hidden_layers=r.num_hidden_layers,
                                     num_attention_heads=r.num_attention_heads, intermediate_size=r.intermediate_size,
                                     num_query_tokens=r.num_query_tokens, layer_norm_eps=r.layer_norm_eps,
                                     hidden_act=r.hidden_act))


def stub_tokenizer(cfg: O.OracleCfg):
    return SimpleNamespace(img_start_token_id=cfg.img_start_token_id, img_end_token_id=cfg.img_end_token_id,
                           img_token_id=cfg.img_token_id, bos_token_id=1, eos_token_id=2, pad_token_id=0)


def make_hip_model(cfg: O.OracleCfg, W, dtype=torch.bfloat16, device="cuda:0"):
    from visualcla import VisualCLAModel
    m = VisualCLAModel.from_state_dict(to_vcla_config(cfg), W, device=device, torch_dtype=dtype)
    m.tokenizer = stub_tokenizer(cfg)
    m.image_at_head = False
    return m
