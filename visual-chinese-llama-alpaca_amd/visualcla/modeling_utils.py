"""Chat-level API of VisualCLA on MI355X.  Signatures, prompt template, history format, printing and
threading contract follow the reference's models/visualcla/modeling_utils.py (loader :83-141, prompt
builder :49-80, chat :144-178, chat_in_stream :181-247, Stream/Iteratorize :404-472) so that
scripts/inference/inference.py and gradio_demo.py run unmodified against this package.

Host-side only: tokenisation, PIL preprocessing and prompt assembly stay in Python exactly where the
reference has them; the model object they drive is the HIP-backed VisualCLAModel.
"""
from __future__ import annotations

import os

import gc
import logging
import traceback
from copy import deepcopy
from queue import Queue
from threading import Thread
from typing import Union

import torch
from PIL import Image

from .configuration_visualcla import VisualCLAConfig
from .modeling_visualcla import VisualCLAModel

logger = logging.getLogger(__name__)

PROMPT_TEMPLATE_MULTIMODAL = (
    "Below is an instruction that describes a task. "
    "Write a response that appropriately completes the request.\n\n"
)
prompt_sep_before = "### "
prompt_sep_after = "\n\n"


def _default_generation_config():
    from transformers import GenerationConfig
    # the reference's sampling defaults (modeling_utils.py:36-47)
    return GenerationConfig(max_new_tokens=512, min_length=0, do_sample=True, top_p=0.9, top_k=40, num_beams=1,
                            temperature=0.5, num_return_sequences=1, no_repeat_ngram_size=15, repetition_penalty=1.1)


DEFAULT_GENERATION_CONFIG = _default_generation_config()


def _instruction_block(text: str, with_image: bool) -> str:
    body = ("<image_placeholder>\n" + text) if with_image else text
    return prompt_sep_before + "Instruction" + ": \n" + body + prompt_sep_after


def encoding_text(history, text, num_patch, tokenizer):
    """Alpaca-style multi-turn prompt; the first instruction carries the image slot
    `<img>` + num_patch x `<img_token>` + `</img>`; BOS is prepended as text and the tokenizer is called
    with add_special_tokens=False (reference :49-80)."""
    prompt = _instruction_block(text, with_image=(history == [])) + prompt_sep_before + "Response" + ":"
    for turn in reversed(history):
        kind = turn["type"]
        if kind == "instruction":
            prompt = _instruction_block(turn["value"], with_image=("first_instruction" in turn)) + prompt
        elif kind == "response":
            prompt = prompt_sep_before + "Response" + ":" + turn["value"] + prompt_sep_after + prompt
        else:
            raise ValueError(f"Except 'type' are 'instruction' and 'response', but get '{kind}'.")
    prompt = PROMPT_TEMPLATE_MULTIMODAL + prompt
    slot = tokenizer.img_start_token + num_patch * tokenizer.img_token + tokenizer.img_end_token
    prompt = prompt.replace("<image_placeholder>", slot)
    return tokenizer(tokenizer.bos_token + prompt, return_tensors="pt", add_special_tokens=False)


def attach_special_tokens(tokenizer):
    """The three image tokens + pad token the reference sets on the tokenizer (:96-102)."""
    tokenizer.pad_token = "<pad>"
    tokenizer.img_start_token = "<img>"
    tokenizer.img_end_token = "</img>"
    tokenizer.img_token = "<img_token>"
    tokenizer.img_start_token_id = tokenizer.convert_tokens_to_ids(tokenizer.img_start_token)
    tokenizer.img_end_token_id = tokenizer.convert_tokens_to_ids(tokenizer.img_end_token)
    tokenizer.img_token_id = tokenizer.convert_tokens_to_ids(tokenizer.img_token)
    return tokenizer


def attach_runtime(model: VisualCLAModel, tokenizer, image_processor):
    """What the loader hangs on the model object (:130-139)."""
    image_processor.patch_size = model.vision_model.config.patch_size
    model.tokenizer = tokenizer
    model.image_processor = image_processor
    model.image_at_head = False
    nq = model.config.visual_resampler_config["num_query_tokens"]
    if nq != -1:
        model.num_patch = nq
    else:
        model.num_patch = (image_processor.size["shortest_edge"] // image_processor.patch_size) ** 2 + 1
    return model


def get_model_and_tokenizer_and_processor(visualcla_model=None, text_model=None, vision_model=None, lora_model=None,
                                          torch_dtype=torch.float16, default_device=None, device_map=None,
                                          load_in_8bit=False, gpu_preprocess=False, fold_lora_adapter=True):
    """-> (model, tokenizer, image_processor).  `torch_dtype=float16` (the reference default) selects the bf16
    MI355X path; float32 selects the fp32 parity mode.  `gpu_preprocess=True` (not in the reference) swaps the returned
    CLIPImageProcessor for `preprocess.GpuClipImageProcessor`: same call, same pixels, computed on the device.
    With text_model / vision_model / lora_model the reference returns the BASE model and its callers wrap it in
    peft.PeftModel (inference.py:66-75, merge_llama_with_visualcla_lora.py:78-85); here the adapter found in `lora_model` is
    folded into the weights at load (`fold_lora_adapter=False` restores the reference's base-only behaviour)."""
    from transformers import CLIPImageProcessor, LlamaTokenizer
    tokenizer = attach_special_tokens(LlamaTokenizer.from_pretrained(visualcla_model or lora_model))
    if visualcla_model is not None:
        logger.info("Init VisualCLA model from pretrained")
        model = VisualCLAModel.from_merged_pretrained(visualcla_model, torch_dtype=torch_dtype,
                                                      default_device=default_device, device_map=device_map,
                                                      load_in_8bit=load_in_8bit)
    else:
        assert text_model is not None and vision_model is not None
        logger.info("Init VisualCLA model with pretrained text/image encoders")
        model = VisualCLAModel.from_vision_text_pretrained(vision_model, text_model,
                                                           visualcla_config=VisualCLAConfig.from_pretrained(lora_model),
                                                           torch_dtype=torch_dtype, default_device=default_device,
                                                           device_map=device_map, load_in_8bit=load_in_8bit,
                                                           lora_model=lora_model if fold_lora_adapter and os.path.isfile(
                                                               os.path.join(lora_model, "adapter_model.bin")) else None)
    image_processor = CLIPImageProcessor.from_pretrained(vision_model or visualcla_model)
    if gpu_preprocess:
        from .preprocess import GpuClipImageProcessor
        image_processor = GpuClipImageProcessor.from_hf(image_processor, device=model.device)
    attach_runtime(model, tokenizer, image_processor)
    return model, tokenizer, image_processor


# BASELINE.json's wording; same 3-tuple
get_model_and_processor = get_model_and_tokenizer_and_processor


def _prepare(model, image, text, history, generation_config):
    generation_config = generation_config or DEFAULT_GENERATION_CONFIG
    generation_config.bos_token_id = generation_config.bos_token_id or model.tokenizer.bos_token_id
    if isinstance(image, str):
        pixel_values = model.image_processor(Image.open(image), return_tensors="pt").pixel_values
    elif isinstance(image, Image.Image):
        pixel_values = model.image_processor(image, return_tensors="pt").pixel_values
    else:
        pixel_values = image
    enc = encoding_text(history, text, model.num_patch, model.tokenizer)
    enc["pixel_values"] = pixel_values.to(model.dtype)   # reference: .half() on GPU (:156-159); dtype follows the model here
    enc = enc.to(model.device)
    if len(history) == 0:
        history.append({"type": "instruction", "value": text, "first_instruction": True})
    else:
        history.append({"type": "instruction", "value": text})
    return generation_config, enc


@torch.inference_mode()
def chat(model, image: Union[str, "Image.Image", torch.Tensor], text: str, history=[], generation_config=None):
    """-> (response, history); appends to / mutates the caller's `history` and prints both, as the reference does."""
    generation_config, enc = _prepare(model, image, text, history, generation_config)
    outputs = model.generate(input_ids=enc.input_ids, attention_mask=enc.attention_mask,
                             pixel_values=enc.pixel_values, generation_config=generation_config)
    response = model.tokenizer.decode(outputs[0], skip_special_tokens=True)
    history.append({"type": "response", "value": response})
    print("Response:", response)
    print("History:", history)
    return response, history


@torch.inference_mode()
def chat_in_stream(model, image: Union[str, "Image.Image", torch.Tensor], text: str, history=[], generation_config=None):
    """Generator of (response_so_far, history).  generate() runs in a worker thread and hands the growing id
    sequence over a queue from a per-token stopping-criteria callback (reference :215-225, :404-472)."""
    from transformers import LlamaTokenizer
    generation_config, enc = _prepare(model, image, text, history, generation_config)
    origin_size = len(enc.input_ids[0])
    eos_token_id = model.tokenizer.eos_token_id
    response = ""
    old_history = deepcopy(history)

    # to_dict() spells out every unset field as None; passed as keyword arguments those Nones would override the model's
    # own eos / pad ids (explicit kwargs win in generate()), and the stream would run to max_new_tokens: drop them
    params = {k: v for k, v in generation_config.to_dict().items() if v is not None}
    params["input_ids"] = enc.input_ids
    params["attention_mask"] = enc.attention_mask
    params["pixel_values"] = enc.pixel_values

    def generate_with_callback(callback=None, **kw):
        kw.setdefault("stopping_criteria", [])
        kw["stopping_criteria"] = list(kw["stopping_criteria"]) + [Stream(callback_func=callback)]
        clear_torch_cache()
        with torch.no_grad():
            model.generate(**kw)

    with Iteratorize(generate_with_callback, params, callback=None) as generator:
        for next_token_ids in generator:
            if len(next_token_ids) > 0 and next_token_ids[0] == eos_token_id:
                break
            next_tokens = model.tokenizer.decode(next_token_ids, skip_special_tokens=True)
            if type(model.tokenizer) is LlamaTokenizer and len(next_token_ids) > 0:
                if model.tokenizer.convert_ids_to_tokens(int(next_token_ids[0])).startswith("▁"):
                    next_tokens = " " + next_tokens
            response = next_tokens
            history = deepcopy(old_history)
            history.append({"type": "response", "value": response})
            yield response, history
            if len(enc.input_ids[0]) > origin_size + generation_config.max_new_tokens:
                break
        print("Response:", response)
        print("History:", history)


def hijack_samplers():
    """The reference monkey-patches HF's sampler factory to add TailFree/TopA/Mirostat warpers (:395-400) and never
    calls it.  Those warpers are outside the greedy hot path; kept as an importable no-op."""
    logger.warning("hijack_samplers(): extra samplers are not part of the MI355X hot path; ignored")


class Stream:
    """Stopping-criteria shaped callback: forwards the first sequence's ids after every token, never stops."""

    def __init__(self, callback_func=None):
        self.callback_func = callback_func

    def __call__(self, input_ids, scores) -> bool:
        if self.callback_func is not None:
            self.callback_func(input_ids[0])
        return False


class Iteratorize:
    """Turns a function that reports progress through a callback into a lazy iterator (worker thread + queue).
    Leaving the `with` block makes the next callback raise, which unwinds generate() in the worker."""

    def __init__(self, func, kwargs=None, callback=None):
        self.mfunc = func
        self.c_callback = callback
        self.q = Queue()
        self.sentinel = object()
        self.kwargs = kwargs or {}
        self.stop_now = False

        def _callback(val):
            if self.stop_now:
                raise ValueError
            self.q.put(val)

        def gentask():
            ret = None
            try:
                ret = self.mfunc(callback=_callback, **self.kwargs)
            except ValueError:
                pass
            except Exception:
                traceback.print_exc()
            self.q.put(self.sentinel)
            if self.c_callback:
                self.c_callback(ret)

        self.thread = Thread(target=gentask)
        self.thread.start()

    def __iter__(self):
        return self

    def __next__(self):
        obj = self.q.get(True, None)
        if obj is self.sentinel:
            raise StopIteration
        return obj

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.stop_now = True
        clear_torch_cache()


def clear_torch_cache():
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
