"""Generate golden vectors by running the REFERENCE'S OWN model code.

TEST INFRASTRUCTURE ONLY (see oracle/visualcla_oracle.py header).  Run in the
build container, where /root/reference exists:

    python oracle/make_golden.py            # writes tests/golden/*.npz

It imports `models/visualcla/{configuration_visualcla,modeling_visual_resampler,
modeling_visualcla}.py` from /root/reference *unmodified*, behind the 4-item
in-memory shim that SURVEY.md section 8c documents (the reference targets
transformers >= 4.29, this image has 5.15), loads the seeded synthetic weights of
`oracle.visualcla_oracle.make_weights` into the reference `VisualCLAModel`, runs
`VisualCLAModel.forward` / `.generate` on the seeded inputs, and stores per-stage
tensors (forward hooks on the reference's own sub-modules).  The files it writes
are committed; /root/reference does not exist on the GPU box.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = os.environ.get("VCLA_REFERENCE", "/root/reference")
REF_PKG = os.path.join(REF, "models", "visualcla")

from oracle import visualcla_oracle as O  # noqa: E402


def load_reference():
    """SURVEY.md section 8c recipe, steps (1)-(3)."""
    import transformers
    import transformers.pytorch_utils as pu
    from transformers.modeling_utils import PreTrainedModel

    if not hasattr(pu, "find_pruneable_heads_and_indices"):
        pu.find_pruneable_heads_and_indices = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    if not hasattr(PreTrainedModel, "get_head_mask"):
        PreTrainedModel.get_head_mask = lambda self, hm, n, *a: [None] * n
    pkg = types.ModuleType("visualcla")
    pkg.__path__ = [REF_PKG]
    sys.modules["visualcla"] = pkg
    mods = {}
    for name in ("configuration_visualcla", "modeling_visual_resampler", "modeling_visualcla"):
        spec = importlib.util.spec_from_file_location(f"visualcla.{name}", os.path.join(REF_PKG, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"visualcla.{name}"] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods


def build_reference_model(mods, cfg: O.OracleCfg, W):
    from transformers import LlamaConfig
    from transformers.models.clip.modeling_clip import CLIPVisionConfig

    v, r, t = cfg.vision, cfg.resampler, cfg.text
    text_config = LlamaConfig(
        vocab_size=t.vocab_size, hidden_size=t.hidden_size, intermediate_size=t.intermediate_size,
        num_hidden_layers=t.num_hidden_layers, num_attention_heads=t.num_attention_heads,
        num_key_value_heads=t.num_attention_heads, rms_norm_eps=t.rms_norm_eps,
        max_position_embeddings=t.max_position_embeddings, rope_theta=t.rope_theta,
        tie_word_embeddings=False, attention_bias=False, mlp_bias=False,
        pad_token_id=None, bos_token_id=1, eos_token_id=2, attn_implementation="eager")
    vision_config = CLIPVisionConfig(
        hidden_size=v.hidden_size, intermediate_size=v.intermediate_size,
        num_hidden_layers=v.num_hidden_layers, num_attention_heads=v.num_attention_heads,
        image_size=v.image_size, patch_size=v.patch_size, hidden_act=v.hidden_act,
        layer_norm_eps=v.layer_norm_eps, attn_implementation="eager")
    resampler_config = dict(
        hidden_size=r.hidden_size, num_hidden_layers=r.num_hidden_layers,
        num_attention_heads=r.num_attention_heads, intermediate_size=r.intermediate_size,
        hidden_act=r.hidden_act, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
        layer_norm_eps=r.layer_norm_eps, num_query_tokens=r.num_query_tokens,
        is_decoder=False, add_cross_attention=False)          # shim item (4)
    VisualCLAConfig = mods["configuration_visualcla"].VisualCLAConfig
    VisualCLAModel = mods["modeling_visualcla"].VisualCLAModel
    config = VisualCLAConfig(text_config=text_config.to_dict(), vision_config=vision_config.to_dict(),
                             use_visual_resampler=True, visual_resampler_config=resampler_config)
    model = VisualCLAModel(config).eval().float()
    # shim item (5): transformers 5.x CLIPVisionModel is flat, the reference reads .vision_model.post_layernorm
    object.__setattr__(model.vision_model, "vision_model", model.vision_model)

    sd = {}
    for k, val in W.items():
        k2 = k.replace("vision_model.vision_model.", "vision_model.")     # flat CLIP in 5.x
        sd[k2] = val.float()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [m for m in missing if "position_ids" not in m and "inv_freq" not in m]
    assert not missing and not unexpected, (missing, unexpected)
    model.image_at_head = False                                   # modeling_utils.py:134
    model.tokenizer = types.SimpleNamespace(img_start_token_id=cfg.img_start_token_id,
                                            img_end_token_id=cfg.img_end_token_id,
                                            img_token_id=cfg.img_token_id)
    return model


def run_reference(model, cfg, pixel_values, input_ids, attention_mask, n_new):
    taps = {}
    hooks = []

    def tap(name, idx=None):
        def fn(_m, _i, out):
            o = out[0] if isinstance(out, (tuple, list)) else out
            if hasattr(o, "last_hidden_state"):
                o = o.last_hidden_state
            taps[name] = o.detach().float().clone()
        return fn

    vm = model.vision_model
    hooks.append(vm.pre_layrnorm.register_forward_hook(tap("vit_embed")))
    for i, l in enumerate(vm.encoder.layers):
        hooks.append(l.register_forward_hook(tap(f"vit_layer{i}")))
    hooks.append(vm.post_layernorm.register_forward_hook(tap("vit_post_ln")))
    for i, l in enumerate(model.visual_resampler.encoder.layer):
        hooks.append(l.register_forward_hook(tap(f"resampler_layer{i}")))
    hooks.append(model.image_projection_layer.register_forward_hook(tap("image_embeds")))
    for i, l in enumerate(model.text_model.model.layers):
        hooks.append(l.register_forward_hook(tap(f"llama_layer{i}")))
    hooks.append(model.text_model.model.norm.register_forward_hook(tap("final_norm")))
    with torch.no_grad():
        out = model(input_ids=input_ids, pixel_values=pixel_values, attention_mask=attention_mask,
                    use_cache=False, return_dict=True)
    # the CLS-only post_layernorm HF applies internally fires the same hook first; the reference's
    # all-token call fires last, so the tap holds the [B, N, D] tensor.
    taps["logits"] = out.logits.detach().float()
    for h in hooks:
        h.remove()
    from transformers import GenerationConfig
    gen = GenerationConfig(max_new_tokens=n_new, min_new_tokens=n_new, do_sample=False, num_beams=1,
                           bos_token_id=1, eos_token_id=None, pad_token_id=0)
    with torch.no_grad():
        toks = model.generate(input_ids=input_ids, pixel_values=pixel_values,
                              attention_mask=attention_mask, generation_config=gen)
    taps["generated"] = toks.detach().to(torch.int64)
    return taps


CASES = {
    # name: (cfg builder, batch, seq_len, n_new, which taps to keep (None = all))
    "tiny_b2": (O.cfg_tiny, 2, 24, 6, None),
    "small_b2": (O.cfg_small, 2, 48, 5,
                 ["vit_embed", "vit_layer0", "vit_post_ln", "resampler_layer0", "resampler_layer1",
                  "image_embeds", "llama_layer0", "final_norm", "logits", "generated"]),
}


def main():
    mods = load_reference()
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    for name, (mk, B, T, n_new, keep) in CASES.items():
        cfg = mk()
        W = O.make_weights(cfg, seed=0)
        px, ids, mask = O.make_inputs(cfg, B, T)
        model = build_reference_model(mods, cfg, W)
        taps = run_reference(model, cfg, px, ids, mask, n_new)
        if keep is not None:
            taps = {k: v for k, v in taps.items() if k in keep}
        arrs = {k: (v.numpy().astype(np.float32) if v.dtype != torch.int64 else v.numpy()) for k, v in taps.items()}
        arrs["_input_ids"] = ids.numpy()
        arrs["_meta"] = np.array([B, T, n_new], dtype=np.int64)
        path = os.path.join(outdir, f"ref_{name}.npz")
        np.savez_compressed(path, **arrs)
        print(f"{name}: wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB); taps: {sorted(arrs)}")
        print("   logits std", float(taps['logits'].std()), "generated", taps["generated"].tolist())


if __name__ == "__main__":
    main()
