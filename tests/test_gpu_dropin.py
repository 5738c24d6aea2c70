"""Drop-in test of the chat-level API: a synthetic checkpoint in the MERGED ON-DISK LAYOUT the reference writes
(scripts/merge_llama_with_visualcla_lora.py:87-97: config.json, pytorch_model.bin with visual_resampler.* +
image_projection_layer.*, text_encoder/, vision_encoder/, tokenizer + preprocessor files) is loaded with
`visualcla.get_model_and_tokenizer_and_processor`, driven through `visualcla.chat` / `chat_in_stream` with a PIL image,
and the generated ids are compared with the CPU oracle run on the same tokenised prompt and preprocessed pixels."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import visualcla_oracle as O
from tests.helpers import to_vcla_config

pytestmark = pytest.mark.gpu


def _tiny_cfg():
    cfg = O.cfg_tiny()
    cfg.text.vocab_size = 128
    cfg.img_start_token_id, cfg.img_end_token_id, cfg.img_token_id = 3, 4, 6   # order of user_defined_symbols below
    return cfg


def make_merged_dir(path, cfg, W):
    import sentencepiece as spm
    os.makedirs(path, exist_ok=True)
    corpus = os.path.join(path, "corpus.txt")
    with open(corpus, "w") as f:
        f.write("\n".join(["Below is an instruction that describes a task. Write a response that appropriately completes the request.",
                           "### Instruction: what is in the image? ### Response: a cat sitting on a mat",
                           "hello world this is a tiny corpus for a tiny tokenizer"] * 50))
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=os.path.join(path, "tokenizer"), vocab_size=120, model_type="bpe",
                                   user_defined_symbols=["<img>", "</img>", "<pad>", "<img_token>"], pad_id=-1, unk_id=0,
                                   bos_id=1, eos_id=2, character_coverage=1.0, hard_vocab_limit=False, minloglevel=2)
    vc = to_vcla_config(cfg)
    vc.save_pretrained(path)
    top = {k: v for k, v in W.items() if k.startswith(("visual_resampler.", "image_projection_layer."))}
    torch.save(top, os.path.join(path, "pytorch_model.bin"))
    for sub, prefix, sub_cfg in (("text_encoder", "text_model.", vc.text_config), ("vision_encoder", "vision_model.", vc.vision_config)):
        d = os.path.join(path, sub)
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "config.json"), "w") as f:
            json.dump(sub_cfg, f)
        torch.save({k[len(prefix):]: v for k, v in W.items() if k.startswith(prefix)}, os.path.join(d, "pytorch_model.bin"))
    s = cfg.vision.image_size
    with open(os.path.join(path, "preprocessor_config.json"), "w") as f:
        json.dump({"image_processor_type": "CLIPImageProcessor", "do_resize": True, "size": {"shortest_edge": s},
                   "do_center_crop": True, "crop_size": {"height": s, "width": s}, "do_rescale": True, "rescale_factor": 1 / 255,
                   "do_normalize": True, "image_mean": [0.48145466, 0.4578275, 0.40821073],
                   "image_std": [0.26862954, 0.26130258, 0.27577711], "do_convert_rgb": True, "resample": 3}, f)
    return path


@pytest.fixture(scope="module")
def loaded(tmp_path_factory):
    import visualcla
    cfg = _tiny_cfg()
    W = O.make_weights(cfg, seed=0)
    path = make_merged_dir(str(tmp_path_factory.mktemp("merged")), cfg, W)
    model, tokenizer, image_processor = visualcla.get_model_and_tokenizer_and_processor(
        visualcla_model=path, torch_dtype=torch.float32, default_device="cuda:0")
    return visualcla, model, tokenizer, image_processor, cfg, W


def _image():
    from PIL import Image
    rng = np.random.default_rng(0)
    return Image.fromarray((rng.random((90, 120, 3)) * 255).astype(np.uint8))


def test_loader_attaches_what_callers_use(loaded):
    visualcla, model, tokenizer, image_processor, cfg, W = loaded
    assert model.image_at_head is False and model.num_patch == cfg.resampler.num_query_tokens
    assert model.tokenizer is tokenizer and model.image_processor is image_processor
    assert (tokenizer.img_start_token_id, tokenizer.img_end_token_id, tokenizer.img_token_id) == (3, 4, 6)
    assert image_processor.patch_size == cfg.vision.patch_size
    assert model.device == torch.device("cuda:0")
    assert model.get_input_embeddings().weight.shape == (cfg.text.vocab_size, cfg.text.hidden_size)
    sd = model.state_dict()
    assert torch.equal(sd["image_projection_layer.weight"], W["image_projection_layer.weight"])


def test_chat_matches_oracle_and_mutates_history(loaded, capsys):
    from transformers import GenerationConfig
    visualcla, model, tokenizer, image_processor, cfg, W = loaded
    img = _image()
    history = []
    gc = GenerationConfig(max_new_tokens=6, do_sample=False, eos_token_id=None)
    response, hist = visualcla.chat(model, img, "what is this?", history=history, generation_config=gc)
    out = capsys.readouterr().out
    assert "Response:" in out and "History:" in out                      # the reference prints both
    assert hist is history and [h["type"] for h in history] == ["instruction", "response"]
    assert history[0].get("first_instruction") is True and history[1]["value"] == response
    # oracle on the same tokenised prompt + the same preprocessed pixels
    from visualcla.modeling_utils import encoding_text
    enc = encoding_text([], "what is this?", model.num_patch, tokenizer)
    px = image_processor(img, return_tensors="pt").pixel_values
    want = O.visualcla_generate(enc.input_ids, px, enc.attention_mask, W, cfg, max_new_tokens=6)
    assert response == tokenizer.decode(want[0], skip_special_tokens=True)
    # second turn re-uses the history (image slot only in the first instruction)
    r2, _ = visualcla.chat(model, img, "and now?", history=history, generation_config=gc)
    assert len(history) == 4 and isinstance(r2, str)


def test_chat_stops_at_the_models_own_eos(loaded):
    """The reference's DEFAULT_GENERATION_CONFIG leaves eos_token_id at None; HF's generate() then falls back on the model's own
    generation config (text_encoder/generation_config.json or the ids in the LLaMA config.json).  chat() must stop there."""
    from transformers import GenerationConfig
    from visualcla.modeling_utils import encoding_text
    visualcla, model, tokenizer, image_processor, cfg, W = loaded
    img = _image()
    enc = encoding_text([], "what is this?", model.num_patch, tokenizer)
    px = image_processor(img, return_tensors="pt").pixel_values
    want = O.visualcla_generate(enc.input_ids, px, enc.attention_mask, W, cfg, max_new_tokens=6)
    eos = int(want[0, 2])
    n_stop = int((want[0] == eos).nonzero()[0]) + 1                 # tokens up to and including the first eos
    old = model.generation_config
    try:
        model.generation_config = GenerationConfig(eos_token_id=eos, pad_token_id=0)
        gc = GenerationConfig(max_new_tokens=6, do_sample=False)      # eos_token_id None, as in the reference's default config
        response, _ = visualcla.chat(model, img, "what is this?", history=[], generation_config=gc)
        assert response == tokenizer.decode(want[0, :n_stop], skip_special_tokens=True)
        pieces = list(visualcla.chat_in_stream(model, img, "what is this?", history=[], generation_config=gc))
        assert len(pieces) == n_stop                               # the stream ends with the eos token too
        # an explicit keyword still overrides (the benchmark disables the stop this way)
        toks = model.generate(input_ids=enc.input_ids.cuda(), pixel_values=px.cuda(), attention_mask=enc.attention_mask.cuda(),
                              generation_config=gc, eos_token_id=None)
        assert toks.shape[1] == 6
    finally:
        model.generation_config = old


def test_chat_in_stream_and_default_sampling(loaded):
    from transformers import GenerationConfig
    visualcla, model, tokenizer, image_processor, cfg, W = loaded
    img = _image()
    gc = GenerationConfig(max_new_tokens=5, do_sample=False, eos_token_id=None)
    final, _ = visualcla.chat(model, img, "describe", history=[], generation_config=gc)
    pieces = list(visualcla.chat_in_stream(model, img, "describe", history=[], generation_config=gc))
    assert len(pieces) == 5                                # one yield per generated token, from the worker thread
    assert pieces[-1][0].strip() == final.strip()
    # the reference's default sampling config (top-k/top-p/temperature/repetition penalty/no-repeat-ngram) must run
    gs = GenerationConfig(max_new_tokens=4, do_sample=True, top_p=0.9, top_k=40, temperature=0.5, repetition_penalty=1.1,
                          no_repeat_ngram_size=15, eos_token_id=None)
    r, _ = visualcla.chat(model, img, "describe", history=[], generation_config=gs)
    assert isinstance(r, str)


def test_bad_history_and_missing_path_errors(loaded):
    visualcla, model, *_ = loaded
    with pytest.raises(ValueError):
        visualcla.chat(model, _image(), "x", history=[{"type": "bogus", "value": "y"}])
    with pytest.raises(ValueError):
        visualcla.VisualCLAModel.from_merged_pretrained("/nonexistent/dir", torch_dtype=torch.float16, default_device="cuda:0",
                                                        device_map=None, load_in_8bit=False)


def test_load_in_8bit_selects_the_fp8_weight_path(tmp_path):
    """the reference's `load_in_8bit=True` (bitsandbytes int8 on the LLaMA, modeling_visualcla.py:151-156) maps onto the fp8
    weight copies (W8A16: fp8 weights, bf16 activations) instead of raising; chat() runs on them"""
    import visualcla
    cfg = _tiny_cfg()
    W = O.make_weights(cfg, seed=0)
    path = make_merged_dir(str(tmp_path / "merged8"), cfg, W)
    model, tokenizer, image_processor = visualcla.get_model_and_tokenizer_and_processor(
        visualcla_model=path, torch_dtype=torch.float16, default_device="cuda:0", load_in_8bit=True)
    assert model.fp8_decode and not getattr(model, "_fp8_mfma", False)
    from transformers import GenerationConfig
    resp, hist = visualcla.chat(model, _image(), "what is in the image?", history=[],
                                generation_config=GenerationConfig(max_new_tokens=4, do_sample=False))
    assert isinstance(resp, str) and len(hist) == 2


def test_gpu_preprocess_gives_the_same_chat(loaded, tmp_path):
    """next row N1: the loader's gpu_preprocess=True processor feeds the same pixels, so chat() answers identically"""
    from transformers import GenerationConfig
    from visualcla.preprocess import GpuClipImageProcessor
    visualcla, model, tokenizer, image_processor, cfg, W = loaded
    img = _image()
    gc = GenerationConfig(max_new_tokens=6, do_sample=False, eos_token_id=None)
    want, _ = visualcla.chat(model, img, "what is this?", history=[], generation_config=gc)
    gpu_proc = GpuClipImageProcessor.from_hf(image_processor, device=model.device)
    assert torch.equal(gpu_proc(img).pixel_values.cpu(), image_processor(img, return_tensors="pt").pixel_values)
    model.image_processor = gpu_proc
    try:
        got, _ = visualcla.chat(model, img, "what is this?", history=[], generation_config=gc)
        p = str(tmp_path / "img.png")
        img.save(p)
        got_path, _ = visualcla.chat(model, p, "what is this?", history=[], generation_config=gc)
    finally:
        model.image_processor = image_processor
    assert got == want and got_path == want


# ---------------------------------------------------------------- next row N3: the un-merged release layout (base + LoRA)
def _write_hf_dir(path, sub_cfg, sd):
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(sub_cfg, f)
    torch.save(sd, os.path.join(path, "pytorch_model.bin"))


def test_unmerged_lora_checkpoint_is_folded_at_load(tmp_path):
    """text_model + vision_model + lora_model (adapter_config.json / adapter_model.bin, README_EN.md:122-133): the adapter
    is folded into the packed weights at load -- W + (alpha/r) B A for the LoRA targets, modules_to_save replaced whole
    (embeddings grown to the tokenizer) -- which is what merge_llama_with_visualcla_lora.py:78-85 does through peft."""
    import shutil
    import visualcla
    cfg = _tiny_cfg()
    W = O.make_weights(cfg, seed=0)                     # plays the fine-tuned ("merged") model
    merged = make_merged_dir(str(tmp_path / "merged"), cfg, W)
    vc = to_vcla_config(cfg)
    g = torch.Generator().manual_seed(5)
    r, alpha = 4, 16
    s = alpha / r
    base_vocab = cfg.text.vocab_size - 4                # the base LLaMA has no image tokens yet
    text_base, vis_base, adapter = {}, {}, {}
    for k, v in W.items():
        if k.startswith("text_model."):
            name = k[len("text_model."):]
            if name.endswith(("q_proj.weight", "v_proj.weight", "down_proj.weight")):
                A, B = torch.randn(r, v.shape[1], generator=g) * 0.05, torch.randn(v.shape[0], r, generator=g) * 0.05
                text_base[name] = v.clone()
                adapter[f"base_model.model.{k[:-len('.weight')]}.lora_A.weight"] = A
                adapter[f"base_model.model.{k[:-len('.weight')]}.lora_B.weight"] = B
            elif name in ("model.embed_tokens.weight", "lm_head.weight"):
                text_base[name] = v[:base_vocab].clone()                       # base: smaller vocabulary
                adapter["base_model.model." + k] = v.clone()                   # modules_to_save: grown, saved whole
            else:
                text_base[name] = v.clone()
        elif k.startswith("vision_model."):
            name = k[len("vision_model."):]
            vis_base[name] = v.clone()
            if name.endswith(("q_proj.weight", "v_proj.weight")):
                A, B = torch.randn(r, v.shape[1], generator=g) * 0.05, torch.randn(v.shape[0], r, generator=g) * 0.05
                adapter[f"base_model.model.{k[:-len('.weight')]}.lora_A.weight"] = A
                adapter[f"base_model.model.{k[:-len('.weight')]}.lora_B.weight"] = B
        else:
            adapter["base_model.model." + k] = v.clone()                       # visual_resampler.*, image_projection_layer.*
    text_cfg = dict(vc.text_config, vocab_size=base_vocab)
    _write_hf_dir(str(tmp_path / "text"), text_cfg, text_base)
    _write_hf_dir(str(tmp_path / "vision"), vc.vision_config, vis_base)
    lora = str(tmp_path / "lora")
    os.makedirs(lora)
    for f in os.listdir(merged):
        if f.startswith(("tokenizer", "special_tokens", "added_tokens", "preprocessor_config")) or f == "config.json":
            shutil.copy(os.path.join(merged, f), lora)
    shutil.copy(os.path.join(merged, "preprocessor_config.json"), str(tmp_path / "vision"))
    with open(os.path.join(lora, "adapter_config.json"), "w") as f:
        json.dump({"peft_type": "LORA", "r": r, "lora_alpha": alpha, "fan_in_fan_out": False,
                   "target_modules": ["q_proj", "v_proj", "down_proj"],
                   "modules_to_save": ["embed_tokens", "lm_head", "visual_resampler", "image_projection_layer"]}, f)
    torch.save(adapter, os.path.join(lora, "adapter_model.bin"))

    model, tokenizer, proc = visualcla.get_model_and_tokenizer_and_processor(
        text_model=str(tmp_path / "text"), vision_model=str(tmp_path / "vision"), lora_model=lora,
        torch_dtype=torch.float32, default_device="cuda:0")
    assert model.config.text_config["vocab_size"] == cfg.text.vocab_size
    sd = model.state_dict()
    bf = lambda t: t.float().to(torch.bfloat16).float()
    for k, v in W.items():
        if "pooler" in k:
            continue
        stem = "base_model.model." + k[:-len(".weight")] if k.endswith(".weight") else None
        if stem and stem + ".lora_A.weight" in adapter:
            want = bf(v.float() + (adapter[stem + ".lora_B.weight"] @ adapter[stem + ".lora_A.weight"]) * s)
        else:
            want = bf(v)
        assert torch.equal(sd[k].reshape(want.shape), want), k
    # and it generates what the oracle generates from those folded weights
    from transformers import GenerationConfig
    img = _image()
    gc = GenerationConfig(max_new_tokens=5, do_sample=False, eos_token_id=None)
    response, _ = visualcla.chat(model, img, "what is this?", history=[], generation_config=gc)
    from visualcla.modeling_utils import encoding_text
    enc = encoding_text([], "what is this?", model.num_patch, tokenizer)
    px = proc(img, return_tensors="pt").pixel_values
    want = O.visualcla_generate(enc.input_ids, px, enc.attention_mask, {k: v for k, v in sd.items()}, cfg, max_new_tokens=5)
    assert response == tokenizer.decode(want[0], skip_special_tokens=True)
    # the reference's base-only behaviour stays reachable
    base_only, *_ = visualcla.get_model_and_tokenizer_and_processor(
        text_model=str(tmp_path / "text"), vision_model=str(tmp_path / "vision"), lora_model=lora,
        torch_dtype=torch.float32, default_device="cuda:0", fold_lora_adapter=False)
    assert base_only.config.text_config["vocab_size"] == base_vocab


# ---------------------------------------------------------------- next row N4: text-generation-webui's vision half
def test_tgwebui_pipeline_embeds_images_like_the_full_model(loaded, tmp_path):
    from visualcla import tgwebui as T
    visualcla, model, tokenizer, image_processor, cfg, W = loaded
    merged = make_merged_dir(str(tmp_path / "merged"), cfg, W)
    imgs = [_image(), _image().rotate(90, expand=True)]
    for gpu_pre in (False, True):
        pipe = T.get_pipeline("visualcla-7b", {"visualcla_merged_model": merged, "vision_bits": 32, "visualcla_gpu_preprocess": gpu_pre})
        feats = pipe.embed_images(imgs)
        assert feats.shape == (2, cfg.resampler.num_query_tokens, cfg.text.hidden_size) and feats.is_cuda
        px = image_processor(imgs, return_tensors="pt").pixel_values
        assert torch.equal(feats, model.embed_images(px))                        # same kernels, same weights
        want = O.image_embeds(px, W, cfg)
        assert (feats.cpu() - want).abs().max() < 1e-3
    # a vision-only context holds no decoder: the llama entry points refuse it, loudly
    with pytest.raises(Exception):
        pipe.vision.model.generate(input_ids=torch.ones(1, 4, dtype=torch.long), max_new_tokens=2)


def test_tgwebui_vision_lora_branch(tmp_path):
    """`visualcla_vision_lora_model`: base CLIP dir + vision-only adapter + resampler / projector files (reference :62-82)"""
    from visualcla import tgwebui as T
    cfg = _tiny_cfg()
    W = O.make_weights(cfg, seed=3)
    vc = to_vcla_config(cfg)
    g = torch.Generator().manual_seed(9)
    r, alpha = 2, 4
    clip = {k[len("vision_model."):]: v.clone() for k, v in W.items() if k.startswith("vision_model.")}
    adapter, folded = {}, dict(W)
    for k, v in clip.items():
        if k.endswith(("k_proj.weight", "out_proj.weight")):
            A, B = torch.randn(r, v.shape[1], generator=g) * 0.05, torch.randn(v.shape[0], r, generator=g) * 0.05
            adapter[f"base_model.model.{k[:-len('.weight')]}.lora_A.weight"] = A
            adapter[f"base_model.model.{k[:-len('.weight')]}.lora_B.weight"] = B
            folded["vision_model." + k] = (v.float() + (B @ A) * (alpha / r)).to(torch.bfloat16).float()
    clip_dir, lora = str(tmp_path / "clip"), str(tmp_path / "vlora")
    _write_hf_dir(clip_dir, vc.vision_config, clip)
    s_ = cfg.vision.image_size
    with open(os.path.join(clip_dir, "preprocessor_config.json"), "w") as f:
        json.dump({"image_processor_type": "CLIPImageProcessor", "do_resize": True, "size": {"shortest_edge": s_}, "do_center_crop": True,
                   "crop_size": {"height": s_, "width": s_}, "do_rescale": True, "rescale_factor": 1 / 255, "do_normalize": True,
                   "image_mean": [0.48145466, 0.4578275, 0.40821073], "image_std": [0.26862954, 0.26130258, 0.27577711],
                   "do_convert_rgb": True, "resample": 3}, f)
    os.makedirs(lora)
    with open(os.path.join(lora, "adapter_config.json"), "w") as f:
        json.dump({"r": r, "lora_alpha": alpha, "target_modules": ["k_proj", "out_proj"]}, f)
    torch.save(adapter, os.path.join(lora, "adapter_model.bin"))
    with open(os.path.join(lora, "visual_resampler_config.json"), "w") as f:
        json.dump(vc.visual_resampler_config, f)
    torch.save({k[len("visual_resampler."):]: v for k, v in W.items() if k.startswith("visual_resampler.")}, os.path.join(lora, "visual_resampler_model.bin"))
    torch.save({k[len("image_projection_layer."):]: v for k, v in W.items() if k.startswith("image_projection_layer.")},
               os.path.join(lora, "image_projection_layer_model.bin"))
    pipe = T.VisualCLA_7B_Pipeline({"visualcla_vision_lora_model": lora, "visualcla_clip_model": clip_dir, "vision_bits": 32})
    img = _image()
    feats = pipe.embed_images([img])
    px = pipe.image_processor([img], return_tensors="pt").pixel_values
    want = O.image_embeds(px, folded, cfg)
    assert (feats.cpu() - want).abs().max() < 1e-3
