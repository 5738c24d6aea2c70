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
