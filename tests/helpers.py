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


def cfg_engine_small() -> O.OracleCfg:
    """the smallest model the persistent B = 1 decode step (csrc/decode_engine.hip) accepts: LLaMA-7B widths (hidden 4096 = 32 heads x 128), two layers,
    intermediate 4096 (16 SwiGLU units and 8 down_proj slots per CU), vocabulary 4200 (9 lm_head slots per CU, the last rows past the vocabulary)"""
    return O.OracleCfg(
        vision=O.VisionCfg(hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256, patch_size=14, image_size=56),
        resampler=O.ResamplerCfg(hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256, num_query_tokens=8),
        text=O.TextCfg(hidden_size=4096, num_hidden_layers=2, num_attention_heads=32, intermediate_size=4096, vocab_size=4200, max_position_embeddings=256),
        img_start_token_id=4196, img_end_token_id=4197, img_token_id=4199)


def engine_steps_vs_oracle(m, cfg: O.OracleCfg, W, T: int = 21, n_steps: int = 3):
    """B = 1: prefill a T-token text prompt, then n_steps decode steps on the HIP model (the engine when VCLA_ENGINE != 0), teacher-forced on ITS tokens
    in the oracle.  Returns [(max |dlogits|, mean |dlogits|, hip argmax, oracle argmax, oracle top-2 margin)] per step, step 0 = the prefill."""
    import torch
    from transformers import LogitsProcessorList
    ids = torch.randint(3, cfg.text.vocab_size - 8, (1, T), generator=torch.Generator().manual_seed(4))
    seen = []

    def grab(ids_, scores):
        seen.append(scores.detach().float().cpu().clone())
        return scores
    toks = m.generate(input_ids=ids.cuda(), max_new_tokens=n_steps + 1, do_sample=False, eos_token_id=None, logits_processor=LogitsProcessorList([grab])).cpu()
    out = []
    with torch.no_grad():
        x = O.embed_and_splice(ids, None, W, cfg)
        cache = [None] * cfg.text.num_hidden_layers
        h = O.llama_forward(x, W, cfg.text, torch.ones(1, T, dtype=torch.int64), cache, 0)
        refs = [O.lm_head(h[:, -1:], W)[:, 0]]
        for s in range(n_steps):
            e = W["text_model.model.embed_tokens.weight"][toks[:, s]][:, None, :]
            h = O.llama_forward(e, W, cfg.text, torch.ones(1, T + s + 1, dtype=torch.int64), cache, T + s)
            refs.append(O.lm_head(h, W)[:, 0])
    for s in range(n_steps + 1):
        d = (seen[s] - refs[s]).abs()
        top2 = refs[s].topk(2, dim=-1).values[0]
        out.append((d.max().item(), d.mean().item(), int(seen[s].argmax()), int(refs[s].argmax()), float(top2[0] - top2[1])))
    return out
