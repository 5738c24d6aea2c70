"""TEST INFRASTRUCTURE for tests/test_reference_repl.py: a CPU stand-in for the ARITHMETIC of visualcla.VisualCLAModel.

The reference's REPL (scripts/inference/inference.py) is the drop-in test SURVEY.md names, but this container has no GPU and the GPU box
has no /root/reference.  So the REPL runs HERE, unmodified, against this repo's `visualcla` package -- loader, tokenizer / image-processor
attachment, prompt assembly, history handling, printing: all the package's own host code -- with the model's forward / generate arithmetic
supplied by the CPU oracle instead of libvisualcla_hip.so (whose parity with the oracle is what the `-m gpu` tests establish).  Never
imported by the product."""
from __future__ import annotations

from types import SimpleNamespace

import torch

from oracle import visualcla_oracle as O
import visualcla.modeling_visualcla as M


class OracleBackedModel(M.VisualCLAModel):
    def __init__(self, config, state_dict):          # deliberately NOT calling the HIP constructor
        self.config = config
        self._device = torch.device("cpu")
        self._dtype = torch.float32
        self.image_at_head = True
        self.tokenizer = None
        self.image_processor = None
        self.num_patch = config.visual_resampler_config["num_query_tokens"]
        self.generation_config = None
        self._ctx = None
        self.W = {k: v.float() for k, v in state_dict.items()}
        t, v = config.text_config, config.vision_config
        self.vision_model = SimpleNamespace(config=SimpleNamespace(**v))
        self.text_model = SimpleNamespace(config=SimpleNamespace(**t), get_input_embeddings=self.get_input_embeddings)

    # -- what the loader and the REPL touch
    def float(self):
        return self

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def get_input_embeddings(self):
        return SimpleNamespace(weight=self.W["text_model.model.embed_tokens.weight"])

    def _ocfg(self) -> O.OracleCfg:
        v, r, t = self.config.vision_config, self.config.visual_resampler_config, self.config.text_config
        tk = self.tokenizer
        return O.OracleCfg(
            vision=O.VisionCfg(hidden_size=v["hidden_size"], num_hidden_layers=v["num_hidden_layers"], num_attention_heads=v["num_attention_heads"],
                               intermediate_size=v["intermediate_size"], patch_size=v["patch_size"], image_size=v["image_size"]),
            resampler=O.ResamplerCfg(hidden_size=r["hidden_size"], num_hidden_layers=r["num_hidden_layers"], num_attention_heads=r["num_attention_heads"],
                                     intermediate_size=r["intermediate_size"], num_query_tokens=r["num_query_tokens"],
                                     layer_norm_eps=r.get("layer_norm_eps", 1e-12)),
            text=O.TextCfg(hidden_size=t["hidden_size"], num_hidden_layers=t["num_hidden_layers"], num_attention_heads=t["num_attention_heads"],
                           intermediate_size=t["intermediate_size"], vocab_size=t["vocab_size"], rms_norm_eps=t.get("rms_norm_eps", 1e-6),
                           max_position_embeddings=t["max_position_embeddings"]),
            img_start_token_id=tk.img_start_token_id, img_end_token_id=tk.img_end_token_id, img_token_id=tk.img_token_id)

    def _request_flags(self, ids, am64, lab, q_slot, special, need_tok, prefix_visible):
        """the five answers of `vcla_check_request` (include/visualcla_hip.h) in plain tensor algebra: what the CPU tests of `_check_request` run on, and the
        checker tests/test_gpu_model.py holds the kernel against on random requests"""
        B, T = ids.shape
        V = self.config.text_config["vocab_size"]
        bad_vocab = bool(((ids < 0) | (ids >= V)).any())
        bad_label = bool(((lab != -100) & ((lab < 0) | (lab >= V))).any()) if lab is not None else False
        img_pos, bad_slot = None, False
        if q_slot > 0:
            s_id, e_id, t_id = special
            is_start = ids == s_id
            has = is_start.any(dim=1)
            if need_tok:
                has = has & (ids == t_id).any(dim=1)
            p0 = is_start.int().argmax(dim=1)
            endpos = p0 + q_slot + 1
            ok = (endpos < T) & (ids.gather(1, endpos.clamp(max=T - 1)[:, None])[:, 0] == e_id)
            bad_slot = bool((has & ~ok).any())
            img_pos = torch.where(has, p0, torch.full_like(p0, -1)).to(torch.int32)
        any_masked, hole = False, False
        if am64 is not None:
            vis = am64 != 0
            any_masked = not bool(vis.all())
            if prefix_visible:
                vis = torch.cat([torch.ones(B, 1, dtype=torch.bool, device=vis.device), vis], dim=1)
            masked_after_visible = (~vis) & (vis.int().cummax(dim=1).values > 0)
            hole = bool((vis & (masked_after_visible.int().cummax(dim=1).values > 0)).any())
        return [bad_vocab, bad_slot, any_masked, hole, bad_label], img_pos

    @torch.no_grad()
    def generate(self, input_ids=None, pixel_values=None, attention_mask=None, generation_config=None, logits_processor=None,
                 stopping_criteria=None, **kwargs):
        gc = self._resolve_generation_config(generation_config, kwargs)          # the package's own host logic
        procs = self._processors(gc, logits_processor)
        eos = self._eos_list(gc)

        def select(logits, generated):
            scores = logits
            for p in procs:
                scores = p(generated, scores)
            if gc.do_sample:
                return torch.multinomial(torch.softmax(scores, dim=-1), num_samples=1)[:, 0]
            return scores.argmax(dim=-1)
        n_new = int(gc.max_new_tokens or 20)
        return O.visualcla_generate(input_ids, pixel_values.float() if pixel_values is not None else None, attention_mask, self.W, self._ocfg(),
                                    max_new_tokens=n_new, eos_token_id=eos[0] if eos else None, select_fn=select,
                                    image_at_head=bool(self.image_at_head))


def install():
    """route the package's checkpoint loaders to the stand-in (they end in VisualCLAModel.from_state_dict)"""
    from visualcla import _lib
    _lib.require_device = lambda: None

    def from_state_dict(cls, config, state_dict, device=None, torch_dtype=torch.bfloat16):
        return OracleBackedModel(config, state_dict)
    M.VisualCLAModel.from_state_dict = classmethod(from_state_dict)
