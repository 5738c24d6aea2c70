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


# --------------------------------------------------------------------------------------------------------------------
# Edge cases of VisualCLAModel.forward / .generate (modeling_visualcla.py:264-392) the two happy-path files above do not reach.
# Every entry stores its INPUTS next to the reference's outputs, so the consumers (tests/test_oracle_vs_golden.py on the CPU,
# tests/test_gpu_model.py on the MI355X) replay exactly what the reference saw.
# --------------------------------------------------------------------------------------------------------------------
def _left_pad(ids, mask, row, k):
    """left-pad `row` by k pad tokens (id 0), dropping its last k ids -- the layout a batched tokenizer with padding_side='left' makes"""
    ids, mask = ids.clone(), mask.clone()
    ids[row] = torch.cat([torch.zeros(k, dtype=ids.dtype), ids[row, :-k]])
    mask[row, :k] = 0
    return ids, mask


def _greedy(model, n_new, **kw):
    from transformers import GenerationConfig
    gen = GenerationConfig(max_new_tokens=n_new, min_new_tokens=n_new, do_sample=False, num_beams=1,
                           bos_token_id=1, eos_token_id=None, pad_token_id=0)
    with torch.no_grad():
        return model.generate(generation_config=gen, **kw).detach().to(torch.int64)


def edge_cases(mods):
    out = {}

    def put(case, **arrs):
        for k, v in arrs.items():
            if isinstance(v, torch.Tensor):
                v = v.detach()
                v = v.numpy() if v.dtype in (torch.int64, torch.int32) else v.float().numpy().astype(np.float32)
            out[f"{case}__{k}"] = np.asarray(v)

    for cname, mk, T, npre, k_pad, n_new in (("tiny", O.cfg_tiny, 34, 5, 3, 6), ("small", O.cfg_small, 48, 6, 5, 5)):
        cfg = mk()
        W = O.make_weights(cfg, seed=0)
        model = build_reference_model(mods, cfg, W)
        px, ids, mask = O.make_inputs(cfg, 2, T, n_prefix=npre)
        # (1) left-padded batch (:307-312 -> HF mask + positions): forward logits and greedy ids
        lids, lmask = _left_pad(ids, mask, 1, k_pad)
        with torch.no_grad():
            lg = model(input_ids=lids, pixel_values=px, attention_mask=lmask, use_cache=False, return_dict=True).logits
        put(f"leftpad_{cname}", input_ids=lids, attention_mask=lmask, logits=lg,
            generated=_greedy(model, n_new, input_ids=lids, pixel_values=px, attention_mask=lmask), meta=np.array([T, npre, n_new]))
        if cname != "tiny":
            continue
        V = cfg.text.vocab_size
        g = torch.Generator().manual_seed(11)
        # (2) text-only forward (:317-319), plain and left-padded
        txt = torch.randint(3, 300, (2, 9), generator=g)
        tmask = torch.ones_like(txt)
        tmask[1, :2] = 0
        with torch.no_grad():
            lg0 = model(input_ids=txt, attention_mask=torch.ones_like(txt), return_dict=True).logits
            lg1 = model(input_ids=txt, attention_mask=tmask, return_dict=True).logits
        put("textonly", input_ids=txt, attention_mask=tmask, logits_full_mask=lg0, logits=lg1)
        # (3) labels -> loss with the image in its slots (image_at_head=False: labels pass through, :321-328)
        labels = ids.clone()
        p0 = 1 + npre
        labels[:, : p0 + cfg.resampler.num_query_tokens + 2] = -100          # supervise the text after </img> only
        labels[1, -2:] = -100
        with torch.no_grad():
            o = model(input_ids=ids, pixel_values=px, attention_mask=mask, labels=labels, return_dict=True)
        put("slot_labels", input_ids=ids, attention_mask=mask, labels=labels, logits=o.logits, loss=o.loss.reshape(1))
        # (4) image_at_head=True with labels (:290-291, :308-310, :313-315): logits AND loss, plus greedy ids in that mode
        hids = torch.cat([torch.tensor([[1, cfg.img_start_token_id, cfg.img_end_token_id]] * 2), torch.randint(3, 300, (2, 9), generator=g)], dim=1)
        hmask = torch.ones_like(hids)
        hlabels = hids.clone()
        hlabels[:, 0] = -100                 # labels[:, 1] stays supervised: the reference's label placement (:315) then shows in the loss
        hlabels[0, 7] = -100
        model.image_at_head = True
        with torch.no_grad():
            o = model(input_ids=hids, pixel_values=px, attention_mask=hmask, labels=hlabels, return_dict=True)
        put("head_labels", input_ids=hids, attention_mask=hmask, labels=hlabels, logits=o.logits, loss=o.loss.reshape(1),
            generated=_greedy(model, 5, input_ids=hids, pixel_values=px, attention_mask=hmask))
        model.image_at_head = False
        # (5) past_key_values pass-through (:321-328): prompt with use_cache=True, then two single-token forwards on the cache
        with torch.no_grad():
            first = model(input_ids=ids[:, :-2], pixel_values=px, attention_mask=mask[:, :-2], use_cache=True, return_dict=True)
            s1 = model(input_ids=ids[:, -2:-1], attention_mask=mask[:, :-1], past_key_values=first.past_key_values, use_cache=True, return_dict=True)
            s2 = model(input_ids=ids[:, -1:], attention_mask=mask, past_key_values=s1.past_key_values, use_cache=True, return_dict=True)
        put("cache", input_ids=ids, attention_mask=mask, prompt_logits=first.logits, step1_logits=s1.logits, step2_logits=s2.logits)
        # (6) a batch where one row has no image slot (:297-299: that row's embeds pass through untouched)
        mids = ids.clone()
        mids[1] = torch.randint(3, 300, (T,), generator=g)
        with torch.no_grad():
            lg = model(input_ids=mids, pixel_values=px, attention_mask=mask, return_dict=True).logits
        put("mixed_rows", input_ids=mids, attention_mask=mask, logits=lg,
            generated=_greedy(model, 4, input_ids=mids, pixel_values=px, attention_mask=mask))
        # (7) masks with zeros BETWEEN visible tokens, forward only (:307-328 never forwards position_ids: HF rotates by arange positions whatever
        #     the mask says): image_at_head=True with a left-padded text mask = [1] * Q ++ [0, 0, 0, 1, ...], and a text-only prompt with a hole
        pids = torch.cat([torch.tensor([[1, cfg.img_start_token_id, cfg.img_end_token_id]] * 2), torch.randint(3, 300, (2, 9), generator=g)], dim=1)
        pids, pmask = _left_pad(pids, torch.ones_like(pids), 1, 3)
        model.image_at_head = True
        plabels = pids.clone()               # the reference's image_at_head forward needs labels (:315 indexes them unconditionally)
        plabels[:, 0] = -100
        plabels[1, :4] = -100
        with torch.no_grad():
            o = model(input_ids=pids, pixel_values=px, attention_mask=pmask, labels=plabels, return_dict=True)
        model.image_at_head = False
        put("head_leftpad", input_ids=pids, attention_mask=pmask, labels=plabels, logits=o.logits, loss=o.loss.reshape(1))
        hole = torch.randint(3, 300, (2, 10), generator=g)
        hmask2 = torch.ones_like(hole)
        hmask2[0, 4] = 0
        hmask2[1, 2:4] = 0
        with torch.no_grad():
            lg = model(input_ids=hole, attention_mask=hmask2, return_dict=True).logits
        put("text_hole", input_ids=hole, attention_mask=hmask2, logits=lg)
        # (8) beam search: the reference forwards num_beams / length_penalty / early_stopping / num_return_sequences to HF generate (:382-391).
        #     (a) no eos: every beam runs to the length limit; (b) an eos id that the beams do produce (the third token of (a)'s best hypothesis of row 0):
        #     hypotheses finish at different lengths, the length penalty ranks them; (c) 4 beams, early_stopping=True, two returned hypotheses per prompt
        from transformers import GenerationConfig

        def beams(**kw):
            gen = GenerationConfig(do_sample=False, bos_token_id=1, pad_token_id=0, **kw)
            with torch.no_grad():
                return model.generate(input_ids=ids, pixel_values=px, attention_mask=mask, generation_config=gen).detach().to(torch.int64)
        ba = beams(num_beams=3, max_new_tokens=6, eos_token_id=None)
        eos_b = int(ba[0, 2])
        bb = beams(num_beams=3, max_new_tokens=8, eos_token_id=eos_b)
        bc = beams(num_beams=4, max_new_tokens=8, eos_token_id=eos_b, early_stopping=True, num_return_sequences=2, length_penalty=0.6)
        put("beams", input_ids=ids, attention_mask=mask, a_generated=ba, b_generated=bb, c_generated=bc, eos=np.array([eos_b]))
    return out


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
    edge = edge_cases(mods)
    path = os.path.join(outdir, "ref_edge_cases.npz")
    np.savez_compressed(path, **edge)
    print(f"edge cases: wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB): {sorted(edge)}")


if __name__ == "__main__":
    main()
