"""text-generation-webui multimodal pipeline for VisualCLA on MI355X (next row N4).

Mirrors the reference's extension (scripts/inference/text_generation_webui/visualcla/visualcla.py + pipelines.py): same
class names, static prompt helpers, settings keys and `embed_images(List[PIL.Image]) -> [n, 64, text_hidden]` contract, so
dropping this file's classes into `extensions/multimodal/pipelines/visualcla/` replaces the HF CLIP + resampler + projector
modules the reference instantiates there (:33-82) with ONE vision-only HIP context (vcla_vision_forward: ViT -> post-LN ->
Resampler -> projection, reference :116-129).  The webui's own LLM (`shared.model`) still embeds the text tokens.

text-generation-webui itself (`modules.shared`, `extensions.multimodal.abstract_pipeline`) is not in this image: the base
class and `shared` are imported lazily and only where the reference uses them, so `VisionHalf` is usable stand-alone.
"""
from __future__ import annotations

import json
import os
import time
from typing import List, Optional, Tuple

import torch

from .configuration_visualcla import VisualCLAConfig
from .modeling_visualcla import VisualCLAModel
from .weights import fold_lora

try:   # inside text-generation-webui
    from extensions.multimodal.abstract_pipeline import AbstractMultimodalPipeline
except Exception:   # stand-alone: same abstract surface, nothing else
    class AbstractMultimodalPipeline:   # type: ignore[no-redef]
        def _get_device(self, setting_name: str, params: dict):
            return torch.device(params.get(setting_name) or "cuda:0")

        def _get_dtype(self, setting_name: str, params: dict):
            return torch.float32 if int(params.get(setting_name) or 16) == 32 else torch.float16


def _settings(params: dict) -> dict:
    try:
        from modules import shared
        merged = dict(shared.settings)
    except Exception:
        merged = {}
    merged.update({k: v for k, v in (params or {}).items() if k.startswith("visualcla_")})
    return merged


class VisionHalf:
    """ViT + Resampler + projection of a VisualCLA checkpoint in a vision-only HIP context (no decoder weights in HBM)."""

    def __init__(self, model: VisualCLAModel, image_processor):
        self.model, self.image_processor = model, image_processor

    @staticmethod
    def _vision_only_config(config: VisualCLAConfig, text_hidden: int) -> VisualCLAConfig:
        config.text_config = {"hidden_size": text_hidden, "num_hidden_layers": 0, "vocab_size": 0, "intermediate_size": 64,
                              "num_attention_heads": max(text_hidden // 128, 1), "max_position_embeddings": 0}
        return config

    @classmethod
    def from_merged(cls, merged_dir: str, device="cuda:0", torch_dtype=torch.float16, gpu_preprocess: bool = False) -> "VisionHalf":
        """`visualcla_merged_model` branch (reference :40-61): vision_encoder/ + the top-level pytorch_model.bin"""
        from transformers import CLIPImageProcessor
        if not os.path.isdir(merged_dir):
            raise ValueError(f"visualcla_merged_model '{merged_dir}' is not a local directory")
        config = VisualCLAConfig.from_pretrained(merged_dir)
        with open(os.path.join(merged_dir, "vision_encoder", "config.json")) as f:
            vc = json.load(f)
        config.vision_config = vc.get("vision_config", vc)
        top = VisualCLAModel._read_checkpoint_dir(merged_dir)
        sd = {k: v for k, v in top.items() if k.startswith(("visual_resampler.", "image_projection_layer."))}
        sd.update({"vision_model." + k: v for k, v in VisualCLAModel._read_checkpoint_dir(os.path.join(merged_dir, "vision_encoder")).items()})
        cls._vision_only_config(config, sd["image_projection_layer.weight"].shape[0])
        model = VisualCLAModel.from_state_dict(config, sd, device, torch_dtype)
        return cls(model, cls._processor(CLIPImageProcessor.from_pretrained(merged_dir), model, gpu_preprocess))

    @classmethod
    def from_vision_lora(cls, clip_dir: str, lora_dir: str, device="cuda:0", torch_dtype=torch.float16,
                         gpu_preprocess: bool = False) -> "VisionHalf":
        """`visualcla_vision_lora_model` branch (reference :62-82): base CLIP + a vision-only peft adapter, plus
        visual_resampler_config.json / visual_resampler_model.bin / image_projection_layer_model.bin saved beside it."""
        from transformers import CLIPImageProcessor
        config = VisualCLAConfig(use_visual_resampler=True)
        with open(os.path.join(clip_dir, "config.json")) as f:
            vc = json.load(f)
        config.vision_config = vc.get("vision_config", vc)
        with open(os.path.join(lora_dir, "visual_resampler_config.json")) as f:
            config.visual_resampler_config = json.load(f)
        sd = {"vision_model." + k: v for k, v in VisualCLAModel._read_checkpoint_dir(clip_dir).items()}
        if os.path.isfile(os.path.join(lora_dir, "adapter_model.bin")):
            with open(os.path.join(lora_dir, "adapter_config.json")) as f:
                acfg = json.load(f)
            # the adapter was trained on the bare CLIPVisionModel: its keys have no `vision_model.` model prefix
            adapter = torch.load(os.path.join(lora_dir, "adapter_model.bin"), map_location="cpu", weights_only=True)
            adapter = {("base_model.model.vision_model." + k[len("base_model.model."):]) if k.startswith("base_model.model.") else
                       "vision_model." + k: v for k, v in adapter.items()}
            fold_lora(sd, adapter, acfg)
        for k, v in torch.load(os.path.join(lora_dir, "visual_resampler_model.bin"), map_location="cpu", weights_only=True).items():
            sd["visual_resampler." + k] = v
        for k, v in torch.load(os.path.join(lora_dir, "image_projection_layer_model.bin"), map_location="cpu", weights_only=True).items():
            sd["image_projection_layer." + k] = v
        cls._vision_only_config(config, sd["image_projection_layer.weight"].shape[0])
        model = VisualCLAModel.from_state_dict(config, sd, device, torch_dtype)
        return cls(model, cls._processor(CLIPImageProcessor.from_pretrained(clip_dir), model, gpu_preprocess))

    @staticmethod
    def _processor(hf_processor, model, gpu_preprocess):
        if not gpu_preprocess:
            return hf_processor
        from .preprocess import GpuClipImageProcessor
        return GpuClipImageProcessor.from_hf(hf_processor, device=model.device)

    @torch.no_grad()
    def embed_images(self, images: List) -> torch.Tensor:
        from collections.abc import Mapping
        out = self.image_processor(images, return_tensors="pt")
        return self.model.embed_images(out["pixel_values"] if isinstance(out, Mapping) else out.pixel_values)


class VisualCLA_Pipeline(AbstractMultimodalPipeline):
    CLIP_REPO = "openai/clip-vit-large-patch14"

    def __init__(self, params: dict) -> None:
        super().__init__()
        params = params or {}
        self.clip_device = self._get_device("vision_device", params)
        self.clip_dtype = self._get_dtype("vision_bits", params)
        self.projector_device = self._get_device("projector_device", params)
        self.projector_dtype = self._get_dtype("projector_bits", params)
        self.vision = self._load_models(params)
        self.image_processor = self.vision.image_processor

    def _load_models(self, params: dict) -> VisionHalf:
        start_ts = time.time()
        settings = _settings(params)
        if "visualcla_merged_model" not in settings and "visualcla_vision_lora_model" not in settings:
            raise KeyError("Except one of 'visualcla_merged_model' and 'visualcla_vision_lora_model' is set in "
                           "setting-visualcla.yaml, but neither was set.")
        gpu_pre = bool(settings.get("visualcla_gpu_preprocess", False))
        if "visualcla_merged_model" in settings:
            vision = VisionHalf.from_merged(settings["visualcla_merged_model"], self.clip_device, self.clip_dtype, gpu_pre)
        else:   # the reference pulls CLIP_REPO from the hub here; offline it must be a local directory
            clip_dir = settings.get("visualcla_clip_model", self.CLIP_REPO)
            if not os.path.isdir(clip_dir):
                raise ValueError(f"'{clip_dir}' is not a local CLIP checkpoint directory (set visualcla_clip_model; no network access)")
            vision = VisionHalf.from_vision_lora(clip_dir, settings["visualcla_vision_lora_model"], self.clip_device, self.clip_dtype, gpu_pre)
        self.load_seconds = time.time() - start_ts
        return vision

    @staticmethod
    def image_start() -> str:
        return "<img>"

    @staticmethod
    def image_end() -> str:
        return "</img>"

    @staticmethod
    def image_placeholder() -> str:
        return "<img_token>"

    @staticmethod
    def num_image_embeds() -> int:
        return 64

    @staticmethod
    def embed_tokens(input_ids: torch.Tensor) -> torch.Tensor:
        from modules import shared
        m = shared.model.model
        func = m.embed_tokens if hasattr(m, "embed_tokens") else m.model.embed_tokens   # AutoGPTQ case
        return func(input_ids).to(shared.model.device, dtype=shared.model.dtype)

    @classmethod
    def placeholder_embeddings(cls) -> torch.Tensor:
        from modules.text_generation import encode
        return cls.embed_tokens(encode(cls.image_placeholder() * cls.num_image_embeds(), add_bos_token=False)[0])

    def embed_images(self, images: List) -> torch.Tensor:
        feats = self.vision.embed_images(images)
        try:   # hand the features to the webui's LLM where it lives, as the reference does (:129)
            from modules import shared
            return feats.to(shared.model.device, dtype=shared.model.dtype)
        except Exception:
            return feats

    @staticmethod
    def visualcla_projector_shape() -> Tuple[int, int]:
        raise NotImplementedError


class VisualCLA_7B_Pipeline(VisualCLA_Pipeline):
    @staticmethod
    def name() -> str:
        return "visualcla-7b"

    @staticmethod
    def placeholder_token_id() -> int:
        return 49957

    @staticmethod
    def visualcla_projector_shape() -> Tuple[int, int]:
        return (1024, 4096)


available_pipelines = ["visualcla-7b"]


def get_pipeline(name: str, params: dict) -> Optional[AbstractMultimodalPipeline]:
    return VisualCLA_7B_Pipeline(params) if name == "visualcla-7b" else None


def get_pipeline_from_model_name(model_name: str, params: dict) -> Optional[AbstractMultimodalPipeline]:
    if "visualcla" not in model_name.lower():
        return None
    return VisualCLA_7B_Pipeline(params) if "7b" in model_name.lower() else None
