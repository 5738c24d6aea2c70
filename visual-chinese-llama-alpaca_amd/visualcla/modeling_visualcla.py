"""VisualCLAModel on MI355X: same Python surface as the reference's
models/visualcla/modeling_visualcla.py (forward :264-330, generate :334-392, loaders :121-261), with every
tensor operation of the hot path executed by libvisualcla_hip.so (hand-written gfx950 kernels).

Host responsibilities kept here (Python, like the reference): argument plumbing, image-slot validation
(ValueError convention of modeling_visualcla.py:300-302/:366-367), the generation loop's control flow
(EOS / stopping criteria / logits processors), checkpoint reading.  There is no CPU or eager-PyTorch
fallback for the arithmetic: without the shared library and a gfx950 device the model raises.
"""
from __future__ import annotations

import ctypes as C
import glob
import os
from types import SimpleNamespace
from typing import Dict, List, Optional, Union

import torch

from . import _lib
from .configuration_visualcla import VisualCLAConfig
from .weights import add_fp8_copies, extend_position_embedding, fold_lora, pack_state_dict, random_packed, unpack_state_dict  # noqa: F401


def _act_dtype(torch_dtype) -> torch.dtype:
    """fp32 -> fp32 parity mode; fp16 / bf16 -> the bf16 product path (the reference's GPU dtype is fp16,
    modeling_utils.py:88; MI355X MFMA kernels here are bf16)."""
    if torch_dtype in (None, torch.float16, torch.bfloat16, "float16", "bfloat16", "auto"):
        return torch.bfloat16
    if torch_dtype in (torch.float32, "float32"):
        return torch.float32
    raise ValueError(f"unsupported torch_dtype {torch_dtype}")


class VclaCache:
    """KV cache handle returned as `past_key_values`: one device tensor [L, 2, B, H, ctx_max, d]."""

    def __init__(self, kv: torch.Tensor, length: int, ctx_max: int):
        self.kv, self.length, self.ctx_max = kv, length, ctx_max

    def get_seq_length(self) -> int:
        return self.length


class _Embedding:
    """Minimal stand-in for nn.Embedding so `model.get_input_embeddings().weight` keeps working
    (scripts/inference/inference.py:67, scripts/merge_llama_with_visualcla_lora.py:60)."""

    def __init__(self, weight: torch.Tensor):
        self.weight = weight
        self.num_embeddings, self.embedding_dim = weight.shape

    def __call__(self, ids: torch.Tensor) -> torch.Tensor:
        raise RuntimeError("token embedding is fused into vcla_embed_splice; call VisualCLAModel.forward/generate")


# transformers' global generation defaults (GenerationConfig._get_default_generation_params in 5.x; the attribute defaults of GenerationConfig before),
# for the fields this path reads; applied to whatever the caller's and the model's configs leave at None, as hf generation/utils.py
# `_prepare_generation_config` does
_HF_GLOBAL_GENERATION_DEFAULTS = dict(do_sample=False, num_beams=1, temperature=1.0, top_k=50, top_p=1.0, typical_p=1.0, repetition_penalty=1.0,
                                      length_penalty=1.0, no_repeat_ngram_size=0, num_return_sequences=1, early_stopping=False, epsilon_cutoff=0.0,
                                      eta_cutoff=0.0, num_beam_groups=1, diversity_penalty=0.0, encoder_repetition_penalty=1.0,
                                      encoder_no_repeat_ngram_size=0, remove_invalid_values=False, use_cache=True)


class VisualCLAModel:
    config_class = VisualCLAConfig
    base_model_prefix = "visualcla"

    # ------------------------------------------------------------------ construction
    def __init__(self, config: VisualCLAConfig, packed: Optional[Dict[str, torch.Tensor]] = None,
                 device: Union[str, torch.device, None] = None, torch_dtype=torch.bfloat16, seed: int = 0):
        if not config.use_visual_resampler:
            raise ValueError("VisualCLA-7B always uses the visual resampler (use_visual_resampler=True)")
        _lib.require_device()
        self.config = config
        self._device = torch.device(device if device is not None else "cuda:0")
        if self._device.type != "cuda":
            raise _lib.VclaError("VisualCLAModel runs on an MI355X only (device must be cuda:N); there is no CPU fallback")
        self._dtype = _act_dtype(torch_dtype)
        self.image_at_head = True          # reference default (modeling_visualcla.py:108); the loader flips it
        self.tokenizer = None
        self.image_processor = None
        self.num_patch = config.visual_resampler_config["num_query_tokens"]
        self.generation_config = None
        self._ctx = None
        self._ws: Dict[str, torch.Tensor] = {}
        with torch.cuda.device(self._device):
            self._packed = packed if packed is not None else random_packed(config, self._device, self._dtype, seed)
            self._build_ctx()
        t, v = config.text_config, config.vision_config
        self.vision_embed_dim, self.text_embed_dim = v["hidden_size"], t["hidden_size"]
        self.vision_model = SimpleNamespace(config=SimpleNamespace(**v))
        self.text_model = SimpleNamespace(config=SimpleNamespace(**t), get_input_embeddings=self.get_input_embeddings,
                                          get_output_embeddings=self.get_output_embeddings)

    def _cfg_struct(self) -> _lib.ModelCfg:
        v, r, t = self.config.vision_config, self.config.visual_resampler_config, self.config.text_config
        if t.get("num_key_value_heads") not in (None, t["num_attention_heads"]):
            raise ValueError("grouped-query attention is not used by Chinese-Alpaca-7B and is not supported")
        c = _lib.ModelCfg()
        c.act_dtype = _lib.dtype_code(self._dtype)
        c.v_hidden, c.v_layers, c.v_heads = v["hidden_size"], v["num_hidden_layers"], v["num_attention_heads"]
        c.v_inter, c.v_patch, c.v_image = v["intermediate_size"], v["patch_size"], v["image_size"]
        c.v_channels, c.v_eps = v.get("num_channels", 3), v.get("layer_norm_eps", 1e-5)
        c.r_hidden, c.r_layers, c.r_heads = r["hidden_size"], r["num_hidden_layers"], r["num_attention_heads"]
        c.r_inter, c.r_queries, c.r_eps = r["intermediate_size"], r["num_query_tokens"], r.get("layer_norm_eps", 1e-12)
        c.t_hidden, c.t_layers, c.t_heads = t["hidden_size"], t["num_hidden_layers"], t["num_attention_heads"]
        c.t_inter, c.t_vocab, c.t_max_pos = t["intermediate_size"], t["vocab_size"], t["max_position_embeddings"]
        c.t_eps = t.get("rms_norm_eps", 1e-6)
        c.t_rope_theta = float(t.get("rope_theta") or (t.get("rope_parameters") or {}).get("rope_theta") or 10000.0)
        c.t_fp8_mfma = int(bool(getattr(self, "_fp8_mfma", False)) and self._dtype == torch.bfloat16)
        c.t_kv_fp8 = int(bool(getattr(self, "_kv_fp8", False)) and self._dtype == torch.bfloat16)
        if v.get("hidden_act", "quick_gelu") != "quick_gelu" or r.get("hidden_act", "gelu") != "gelu":
            raise ValueError("only quick_gelu (CLIP) and gelu (resampler) activations are implemented")
        return c

    def _build_ctx(self) -> None:
        lib = _lib.load()
        self._destroy_ctx()
        handle = C.c_void_p()
        cfg = self._cfg_struct()
        _lib.check(lib.vcla_ctx_create(C.byref(cfg), C.byref(handle)))
        self._ctx = handle
        for name, t in self._packed.items():
            if not t.is_contiguous():
                raise ValueError(f"packed tensor {name} is not contiguous")
            _lib.check(lib.vcla_ctx_set_tensor(self._ctx, name.encode(), t.data_ptr(), t.numel() * t.element_size()))
        _lib.check(lib.vcla_ctx_finalize(self._ctx))
        self._pos_dev = torch.zeros(1, dtype=torch.int32, device=self._device)

    def _destroy_ctx(self) -> None:
        if getattr(self, "_ctx", None):
            _lib.load().vcla_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self._destroy_ctx()
        except Exception:
            pass

    @classmethod
    def from_state_dict(cls, config: VisualCLAConfig, state_dict: Dict[str, torch.Tensor], device=None,
                        torch_dtype=torch.bfloat16) -> "VisualCLAModel":
        _lib.require_device()          # fail loudly (no CPU fallback) before touching the device
        dev = torch.device(device if device is not None else "cuda:0")
        packed = pack_state_dict(state_dict, config, dev, _act_dtype(torch_dtype))
        return cls(config, packed, dev, torch_dtype)

    @classmethod
    def from_random(cls, config: VisualCLAConfig, device=None, torch_dtype=torch.bfloat16, seed: int = 0):
        return cls(config, None, device, torch_dtype, seed)

    @staticmethod
    def _read_checkpoint_dir(path: str) -> Dict[str, torch.Tensor]:
        sd: Dict[str, torch.Tensor] = {}
        files = sorted(glob.glob(os.path.join(path, "pytorch_model*.bin"))) + sorted(glob.glob(os.path.join(path, "*.safetensors")))
        if not files:
            raise ValueError(f"no pytorch_model*.bin / *.safetensors under {path}")
        for f in files:
            if f.endswith(".safetensors"):
                from safetensors.torch import load_file
                sd.update(load_file(f, device="cpu"))
            else:
                sd.update(torch.load(f, map_location="cpu", weights_only=True))
        return sd

    @classmethod
    def from_merged_pretrained(cls, visualcla_model_name_or_path: str = None, *args, torch_dtype=torch.float16,
                               default_device=None, device_map=None, load_in_8bit=False, **kwargs):
        """Reads the merged on-disk layout written by scripts/merge_llama_with_visualcla_lora.py:87-97:
        `<dir>/config.json`, `<dir>/pytorch_model*.bin` (visual_resampler.* + image_projection_layer.*),
        `<dir>/text_encoder/`, `<dir>/vision_encoder/` (modeling_visualcla.py:141-179)."""
        import json
        path = visualcla_model_name_or_path
        if path is None or not os.path.isdir(path):
            raise ValueError(f"visualcla model path '{path}' is not a local directory (no network access here)")
        config = VisualCLAConfig.from_pretrained(path)
        top = cls._read_checkpoint_dir(path)
        sd = {k: v for k, v in top.items() if k.startswith(("visual_resampler.", "image_projection_layer."))}
        for sub, prefix in (("text_encoder", "text_model."), ("vision_encoder", "vision_model.")):
            d = os.path.join(path, sub)
            with open(os.path.join(d, "config.json")) as f:
                sub_cfg = json.load(f)
            if sub == "text_encoder":
                config.text_config = sub_cfg
            else:
                config.vision_config = sub_cfg.get("vision_config", sub_cfg)
            for k, v in cls._read_checkpoint_dir(d).items():
                sd[prefix + k] = v
        model = cls.from_state_dict(config, sd, default_device, torch_dtype)
        model.generation_config = cls._load_generation_config(os.path.join(path, "text_encoder"), config.text_config)
        if load_in_8bit:
            # the reference quantises the LLaMA WEIGHTS only (bitsandbytes int8, modeling_visualcla.py:151-156); the MI355X analogue
            # is the OCP fp8 (e4m3fn) weight copies with bf16 activations (W8A16: decode kernels dequantise in registers, the
            # prefill keeps the bf16 MFMA tiles).  The fp8 x fp8 prefill (W8A8, fp8 MFMA pipe) stays an explicit opt-in
            # (enable_fp8_decode(prefill=True)): e4m3 activations add ~2.6 % rms per GEMM whatever the scale granularity
            # (tools/fp8_scale_study.py, profiles/r03_fp8_scale_study.txt).
            model.enable_fp8_decode(True, prefill=False)
        return model

    @staticmethod
    def _load_generation_config(text_dir: Optional[str], text_config: dict):
        """The decoder's generation defaults (eos / bos / pad ids): `<text dir>/generation_config.json` when the checkpoint
        ships one, else the ids in the LLaMA config.json -- what `LlamaForCausalLM.from_pretrained` leaves in
        `model.generation_config` and HF's generate() falls back on for fields the caller's config leaves at None
        (the reference's DEFAULT_GENERATION_CONFIG has eos_token_id=None, modeling_utils.py:36-47)."""
        from transformers import GenerationConfig
        if text_dir and os.path.isfile(os.path.join(text_dir, "generation_config.json")):
            return GenerationConfig.from_pretrained(text_dir)
        ids = {k: text_config.get(k) for k in ("eos_token_id", "bos_token_id", "pad_token_id") if text_config.get(k) is not None}
        return GenerationConfig(**ids) if ids else None

    @classmethod
    def from_vision_text_pretrained(cls, vision_model_name_or_path: str = None, text_model_name_or_path: str = None,
                                    visualcla_config: Union[str, VisualCLAConfig] = None, torch_dtype=torch.float16,
                                    default_device=None, device_map=None, load_in_8bit=False, lora_model: str = None,
                                    **kwargs):
        """Separate CLIP / LLaMA checkpoints + a VisualCLA config; the resampler and projection are
        random-initialised exactly as the reference does before its caller attaches LoRA weights
        (modeling_visualcla.py:184-261).  `lora_model` (not in the reference, whose callers wrap the result in
        `peft.PeftModel`, scripts/inference/inference.py:66-75) folds the un-merged release adapter into the weights at
        load time (weights.fold_lora), since this model is not an nn.Module peft could wrap."""
        import json
        if vision_model_name_or_path is None:
            raise ValueError("If `vision_model` is not defined as an argument, a `vision_model_name_or_path` has to be defined")
        if text_model_name_or_path is None:
            raise ValueError("If `text_model` is not defined as an argument, a `text_model_name_or_path` has to be defined")
        if isinstance(visualcla_config, str):
            visualcla_config = VisualCLAConfig.from_pretrained(visualcla_config)
        with open(os.path.join(text_model_name_or_path, "config.json")) as f:
            visualcla_config.text_config = json.load(f)
        with open(os.path.join(vision_model_name_or_path, "config.json")) as f:
            vc = json.load(f)
            visualcla_config.vision_config = vc.get("vision_config", vc)
        sd = {"text_model." + k: v for k, v in cls._read_checkpoint_dir(text_model_name_or_path).items()}
        sd.update({"vision_model." + k: v for k, v in cls._read_checkpoint_dir(vision_model_name_or_path).items()})
        r, t = visualcla_config.visual_resampler_config, visualcla_config.text_config
        g = torch.Generator().manual_seed(0)
        std = visualcla_config.initializer_range
        Dr, Ir = r["hidden_size"], r["intermediate_size"]
        sd["visual_resampler.query_embeddding"] = torch.zeros(1, r["num_query_tokens"], Dr)
        for i in range(r["num_hidden_layers"]):
            p = f"visual_resampler.encoder.layer.{i}."
            for nm, (n, k) in {"crossattention.self.query": (Dr, Dr), "crossattention.self.key": (Dr, Dr),
                               "crossattention.self.value": (Dr, Dr), "crossattention.output.dense": (Dr, Dr),
                               "intermediate.dense": (Ir, Dr), "output.dense": (Dr, Ir)}.items():
                sd[p + nm + ".weight"] = torch.randn(n, k, generator=g) * std
                sd[p + nm + ".bias"] = torch.zeros(n)
            for nm in ("crossattention.output.LayerNorm", "output.LayerNorm"):
                sd[p + nm + ".weight"], sd[p + nm + ".bias"] = torch.ones(Dr), torch.zeros(Dr)
        sd["image_projection_layer.weight"] = torch.randn(t["hidden_size"], Dr, generator=g) * std
        sd["image_projection_layer.bias"] = torch.zeros(t["hidden_size"])
        if lora_model is not None:
            cfg_path, bin_path = os.path.join(lora_model, "adapter_config.json"), os.path.join(lora_model, "adapter_model.bin")
            if not (os.path.isfile(cfg_path) and os.path.isfile(bin_path)):
                raise ValueError(f"'{lora_model}' holds no adapter_config.json + adapter_model.bin")
            with open(cfg_path) as f:
                fold_lora(sd, torch.load(bin_path, map_location="cpu", weights_only=True), json.load(f))
            # modules_to_save grows the embeddings to the tokenizer's size (merge script :68-75)
            visualcla_config.text_config["vocab_size"] = sd["text_model.model.embed_tokens.weight"].shape[0]
        model = cls.from_state_dict(visualcla_config, sd, default_device, torch_dtype)
        model.generation_config = cls._load_generation_config(text_model_name_or_path, visualcla_config.text_config)
        if load_in_8bit:
            model.enable_fp8_decode(True, prefill=False)   # fp8 weights, bf16 activations (see from_merged_pretrained)
        return model

    # ------------------------------------------------------------------ nn.Module-like surface
    @property
    def device(self) -> torch.device:
        return self._device

    @property
    def dtype(self) -> torch.dtype:
        return self._dtype

    def eval(self):
        return self

    def train(self, mode: bool = False):
        if mode:
            raise RuntimeError("the MI355X VisualCLA path is inference-only")
        return self

    def requires_grad_(self, flag: bool = False):
        return self

    def _switch_dtype(self, dt: torch.dtype):
        if dt == self._dtype:
            return self
        sd = unpack_state_dict(self._packed, self.config)
        self._dtype = dt
        self._packed = pack_state_dict(sd, self.config, self._device, dt)
        self._ws.clear()
        self._build_ctx()
        return self

    def float(self):
        return self._switch_dtype(torch.float32)

    def half(self):
        return self._switch_dtype(torch.bfloat16)

    def bfloat16(self):
        return self._switch_dtype(torch.bfloat16)

    def to(self, *args, **kwargs):
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, torch.dtype):
                self._switch_dtype(_act_dtype(a))
            elif isinstance(a, (str, torch.device)):
                dev = torch.device(a)
                if dev.type != "cuda":
                    raise _lib.VclaError("VisualCLAModel cannot be moved off the GPU: there is no CPU fallback")
                if dev != self._device and dev.index is not None:
                    self._packed = {k: v.to(dev) for k, v in self._packed.items()}
                    self._device = dev
                    self._ws.clear()
                    with torch.cuda.device(dev):
                        self._build_ctx()
        return self

    def enable_fp8_decode(self, enabled: bool = True, prefill: bool = True, kv_cache: bool = False):
        """BASELINE configs[4] weight path: OCP fp8 (e4m3fn, per-row scale) copies of the LLaMA projection / lm_head
        matrices.  The HBM-bound decode kernels (M <= 128) stream them (half the bytes, dequantised in registers); with
        `prefill` (default) the prefill GEMMs (M > 128) run fp8 x fp8 on the fp8 MFMA pipe
        (v_mfma_scale_f32_16x16x128_f8f6f4), activations quantised per row on the fly.  `kv_cache=True` also stores the K / V cache
        as e4m3 bytes (unit scale; at B = 64 the bf16 cache is as many bytes per decode step as the fp8 weights): the prompt's own
        attention still runs on exact bf16 rows, the decode steps read the 1-byte cache; a multi-token forward onto an existing
        cache is refused in that mode.  The vision stack stays bf16.  The MI355X analogue of the reference's `load_in_8bit`
        (bitsandbytes on the LLaMA only, modeling_visualcla.py:155)."""
        if self._dtype != torch.bfloat16:
            raise ValueError("fp8 decode weights need the bf16 activation mode")
        has = any(k.endswith(".q8") for k in self._packed)
        want_mfma = bool(enabled and prefill)
        want_kv = bool(enabled and kv_cache)
        rebuild = False
        if bool(getattr(self, "_fp8_mfma", False)) != want_mfma or bool(getattr(self, "_kv_fp8", False)) != want_kv:
            self._fp8_mfma, self._kv_fp8 = want_mfma, want_kv
            self._ws.clear()
            rebuild = True
        if enabled and not has:
            add_fp8_copies(self._packed)
            rebuild = True
        elif not enabled and has:
            for k in [k for k in self._packed if k.endswith((".q8", ".q8f", ".s8"))]:
                del self._packed[k]
            rebuild = True
        if rebuild:
            self._build_ctx()
        return self

    @property
    def fp8_decode(self) -> bool:
        return any(k.endswith(".q8") for k in self._packed)

    def set_image_size(self, image_size: int):
        """Re-target the vision tower to another input resolution (336 px -> 577 tokens): bicubic position-embedding
        interpolation (reference helper semantics) + context rebuild.  Kernels are shape-generic in the token count."""
        v = self.config.vision_config
        if image_size == v["image_size"]:
            return self
        if image_size % v["patch_size"]:
            raise ValueError(f"image_size {image_size} is not a multiple of the patch size {v['patch_size']}")
        # only the position embedding depends on the resolution: interpolate it on the host (the same arithmetic as the oracle /
        # the reference helper) and swap that one tensor -- every other packed tensor, incl. the fp8 / fragment-major copies, stays
        # Always interpolate FROM the embedding the checkpoint came with (kept on first use): going 224 -> 336 -> 448 equals 224 -> 448,
        # and returning to the native size restores the original values bit for bit.
        key = "vision_model.embeddings.position_embedding.weight"
        if getattr(self, "_pos_native", None) is None:
            self._pos_native = (v["image_size"], self._packed["vit.pos"])
        native_size, native_pos = self._pos_native
        sd = {key: native_pos.detach().float().cpu()}
        if image_size != native_size:
            extend_position_embedding(sd, v["patch_size"], image_size)
        v["image_size"] = image_size
        self.vision_model.config.image_size = image_size
        self._packed["vit.pos"] = sd[key].to(device=self._device, dtype=torch.float32).contiguous()
        self._ws.clear()
        self._build_ctx()
        return self

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return unpack_state_dict(self._packed, self.config)

    def get_input_embeddings(self):
        return _Embedding(self._packed["llama.embed"])

    def get_output_embeddings(self):
        return _Embedding(self._packed["llama.lm_head"][: self.config.text_config["vocab_size"]])

    def resize_token_embeddings(self, new_num_tokens: Optional[int] = None):
        V = self.config.text_config["vocab_size"]
        if new_num_tokens is None or new_num_tokens == V:
            return self.get_input_embeddings()
        sd = self.state_dict()
        for key in ("text_model.model.embed_tokens.weight", "text_model.lm_head.weight"):
            w = sd[key]
            new = torch.zeros(new_num_tokens, w.shape[1])
            n = min(V, new_num_tokens)
            new[:n] = w[:n]
            if new_num_tokens > V:
                new[V:] = torch.randn(new_num_tokens - V, w.shape[1]) * self.config.initializer_range
            sd[key] = new
        self.config.text_config["vocab_size"] = new_num_tokens
        self._packed = pack_state_dict(sd, self.config, self._device, self._dtype)
        self._ws.clear()
        self._build_ctx()
        return self.get_input_embeddings()

    # ------------------------------------------------------------------ buffers
    def _buf(self, key: str, nbytes: int) -> torch.Tensor:
        t = self._ws.get(key)
        if t is None or t.numel() < nbytes:
            t = torch.empty(int(nbytes) + 512, dtype=torch.uint8, device=self._device)
            self._ws[key] = t
        return t

    def _special_ids(self):
        tk = self.tokenizer
        if tk is None or not hasattr(tk, "img_start_token_id"):
            raise ValueError("model.tokenizer with img_start_token_id / img_end_token_id / img_token_id is required "
                             "(get_model_and_tokenizer_and_processor attaches it)")
        return tk.img_start_token_id, tk.img_end_token_id, tk.img_token_id

    # ------------------------------------------------------------------ stages
    def _typed_buf(self, key: str, shape, dtype: torch.dtype) -> torch.Tensor:
        """a persistent device buffer viewed as `shape` / `dtype`: the same address on every call with the same shape, which is what
        lets the engine replay its captured vision / prefill graphs (engine.hip run_macro) instead of re-issuing ~500 launches"""
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        return self._buf(key, nbytes)[:nbytes].view(dtype).view(*shape)

    def embed_images(self, pixel_values: torch.Tensor, taps: Optional[dict] = None, _persistent: bool = False) -> torch.Tensor:
        """[B, 3, H, W] -> [B, num_query_tokens, text_hidden]: ViT + post-LN + Resampler + projection
        (the vision half; also what tgwebui's embed_images() computes, .../visualcla/visualcla.py:116-129)."""
        lib = _lib.load()
        v, r, t = self.config.vision_config, self.config.visual_resampler_config, self.config.text_config
        if pixel_values.dim() != 4 or pixel_values.shape[1] != v.get("num_channels", 3) or \
                pixel_values.shape[2] != v["image_size"] or pixel_values.shape[3] != v["image_size"]:
            raise ValueError(f"Input image size ({tuple(pixel_values.shape)}) doesn't match model "
                             f"({v['image_size']}*{v['image_size']}).")
        B = pixel_values.shape[0]
        px = pixel_values.to(device=self._device, dtype=self._dtype).contiguous()
        shape = (B, r["num_query_tokens"], t["hidden_size"])
        out = self._typed_buf("gen_img", shape, self._dtype) if _persistent else torch.empty(*shape, dtype=self._dtype, device=self._device)
        nbytes = lib.vcla_vision_workspace_bytes(self._ctx, B)
        ws = self._buf("vision", nbytes)
        vit_tap = res_tap = None
        if taps is not None:
            N = (v["image_size"] // v["patch_size"]) ** 2 + 1
            vit_tap = torch.empty(v["num_hidden_layers"] + 2, B, N, v["hidden_size"], dtype=self._dtype, device=self._device)
            res_tap = torch.empty(r["num_hidden_layers"], B, r["num_query_tokens"], r["hidden_size"], dtype=self._dtype, device=self._device)
        with torch.cuda.device(self._device):
            _lib.check(lib.vcla_vision_forward(self._ctx, px.data_ptr(), out.data_ptr(), B, ws.data_ptr(), ws.numel(),
                                               _lib.ptr(vit_tap), _lib.ptr(res_tap), _lib.stream_ptr()))
        if taps is not None:
            for i in range(v["num_hidden_layers"]):
                taps[f"vit_layer{i}"] = vit_tap[i]
            taps["vit_post_ln"] = vit_tap[v["num_hidden_layers"]]
            taps["vit_embed"] = vit_tap[v["num_hidden_layers"] + 1]
            for i in range(r["num_hidden_layers"]):
                taps[f"resampler_layer{i}"] = res_tap[i]
            taps["image_embeds"] = out
        return out

    def _request_flags(self, ids: torch.Tensor, am64: Optional[torch.Tensor], lab: Optional[torch.Tensor], q_slot: int, special, need_tok: bool,
                       prefix_visible: bool):
        """-> ([bad_vocab, bad_slot, any_masked, hole, bad_label], img_pos int32 [B] or None): ONE launch (vcla_check_request) and one copy back -- rounds
        3 - 5 issued ~25 torch launches for the same five answers, 0.4 ms in front of every request."""
        lib = _lib.load()
        B, T = ids.shape
        dev = ids.device
        img_pos = torch.empty(B, dtype=torch.int32, device=dev) if q_slot > 0 else None
        flags_dev = torch.empty(5, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.vcla_check_request(ids.data_ptr(), B, T, self.config.text_config["vocab_size"], q_slot, special[0], special[1], special[2],
                                              int(need_tok), _lib.ptr(am64), am64.shape[1] if am64 is not None else 0, int(prefix_visible),
                                              _lib.ptr(lab), lab.shape[1] if lab is not None else 0, _lib.ptr(img_pos), flags_dev.data_ptr(),
                                              _lib.stream_ptr()))
        return [bool(x) for x in flags_dev.tolist()], img_pos     # the one synchronisation

    def _check_request(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor], Q: int, for_generate: bool,
                       labels: Optional[torch.Tensor] = None):
        """Every data-dependent validation of a request in ONE launch and ONE host synchronisation: ids inside the vocabulary, image slots well-formed
        (modeling_visualcla.py:296-302 / :362-367), the attention mask all ones (-> no key mask at all) or -- generate only -- free of masked positions
        between visible ones, `labels` (forward; already extended for image_at_head) inside the vocabulary or -100.
        Q = 0: no image.  Returns (img_pos int32 [B] or None, extended mask or None when nothing is masked).  Raises the reference's ValueErrors."""
        B, T = input_ids.shape
        dev = input_ids.device
        slotted = Q > 0 and not self.image_at_head
        lab = labels.to(dev, torch.int64).contiguous() if labels is not None else None
        am = am64 = None
        if attention_mask is not None:
            am = attention_mask.to(dev)
            am64 = am.to(torch.int64).contiguous()                # a no-op for the int64 masks tokenizers produce
        f, img_pos = self._request_flags(input_ids.contiguous(), am64, lab, Q if slotted else 0, self._special_ids() if slotted else (0, 0, 0),
                                         need_tok=not for_generate,                               # forward also asks for an <img_token> (:297); generate only for the <img> (:363)
                                         prefix_visible=Q > 0 and self.image_at_head)
        flags = [f[0], f[1], not f[2], f[3], f[4]]                # bad_vocab, bad_slot, all_ones, gap, bad_label
        if am is not None and not flags[2] and Q > 0 and self.image_at_head:     # the reference prepends the image columns (:308-310)
            am = torch.cat([torch.ones(B, Q, dtype=am.dtype, device=dev), am], dim=1)
        if flags[0]:
            raise ValueError("input_ids contain ids outside the vocabulary")
        if flags[1]:
            raise ValueError(f"Num of patch ({Q}) is not equal to the length of pre-filled image patch tokens.")
        if flags[4]:
            raise ValueError("labels contain ids outside the vocabulary")
        if flags[3] and for_generate:
            # RoPE positions here are absolute sequence indices.  In `forward` that IS the reference's arithmetic for every mask: it never
            # forwards position_ids (models/visualcla/modeling_visualcla.py:321-328), so HF rotates by arange positions and the mask only
            # removes keys (fixtures head_leftpad / text_hole of tests/golden/ref_edge_cases.npz).  `generate` under the transformers versions
            # the reference pins (>= 4.29) derives cumsum(attention_mask) - 1 positions instead: for padding at either END of a row that
            # is a constant shift per row (RoPE attention is invariant to it), but zeros BETWEEN visible tokens -- only reachable through
            # image_at_head=True with a left-padded text mask, :307-312 / :372-377 -- would put the text at other relative distances from
            # the image tokens: refuse instead of computing something else.
            raise ValueError("attention_mask has masked positions between visible tokens (image_at_head=True with left padding?); "
                             "generate() supports left- or right-padded masks only")
        return img_pos, (None if (am is None or flags[2]) else am)

    def _embed(self, input_ids: torch.Tensor, image_embeds: Optional[torch.Tensor], img_pos: Optional[torch.Tensor], _persistent: bool = False):
        """-> (inputs_embeds [B, T', D], number of positions the image added in front of the text mask).  Handles both placements;
        `img_pos` comes from _check_request (slot placement) and is ignored for image_at_head."""
        lib = _lib.load()
        t = self.config.text_config
        B, T = input_ids.shape
        V, D = t["vocab_size"], t["hidden_size"]
        Q, extra = 0, 0
        ids = input_ids
        if image_embeds is not None:
            Q = image_embeds.shape[1]
            if self.image_at_head:
                # reference: cat([emb[:, :2], image, emb[:, 2:]]) (modeling_visualcla.py:291) == splice after position 1
                filler = torch.zeros(B, Q, dtype=ids.dtype, device=ids.device)
                ids = torch.cat([ids[:, :2], filler, ids[:, 2:]], dim=1)
                img_pos = torch.full((B,), 1, dtype=torch.int32, device=ids.device)
                extra = Q
        else:
            img_pos = None
        ids = ids.contiguous()
        Tn = ids.shape[1]
        out = self._typed_buf("gen_embeds", (B, Tn, D), self._dtype) if _persistent else torch.empty(B, Tn, D, dtype=self._dtype, device=self._device)
        with torch.cuda.device(self._device):
            _lib.check(lib.vcla_embed_splice(ids.data_ptr(), self._packed["llama.embed"].data_ptr(),
                                             _lib.ptr(image_embeds), _lib.ptr(img_pos), out.data_ptr(), B, Tn, Q, D, V,
                                             _lib.dtype_code(self._dtype), _lib.stream_ptr()))
        return out, extra

    def _prepare_ids(self, input_ids, pixel_values):
        """shape checks that need no device data: -> int64 ids on the device"""
        if input_ids is None:
            raise ValueError("input_ids is required")
        if input_ids.dim() != 2:
            raise ValueError(f"input_ids must be [batch, seq], got {tuple(input_ids.shape)}")
        if pixel_values is not None and pixel_values.shape[0] != input_ids.shape[0]:
            raise ValueError(f"pixel_values hold {pixel_values.shape[0]} images for {input_ids.shape[0]} prompts")
        return input_ids.to(self._device).long()               # the kernel reads int64 ids

    def _new_cache(self, B: int, ctx_max: int, _persistent: bool = False) -> VclaCache:
        t = self.config.text_config
        H, d = t["num_attention_heads"], t["hidden_size"] // t["num_attention_heads"]
        shape = (t["num_hidden_layers"], 2, B, H, ctx_max, d)
        # generate()'s own cache lives in a persistent buffer (never handed to the caller); forward(use_cache=True) returns a fresh one
        kdt = torch.uint8 if getattr(self, "_kv_fp8", False) else self._dtype      # e4m3 bytes (enable_fp8_decode(kv_cache=True))
        kv = self._typed_buf("gen_kv", shape, kdt) if _persistent else torch.empty(*shape, dtype=kdt, device=self._device)
        return VclaCache(kv, 0, ctx_max)

    def _key_mask(self, am: Optional[torch.Tensor], B: int, T: int, ctx_max: int):
        """int32 [B, ctx_max] (1 = attend), or None when nothing is masked.  `am` = the validated mask of _check_request (image columns
        already prepended for image_at_head; None = all ones)."""
        if am is None:
            return None
        if am.shape[1] != T:
            raise ValueError(f"attention_mask length {am.shape[1]} does not match sequence length {T}")
        km = torch.ones(B, ctx_max, dtype=torch.int32, device=self._device)
        km[:, :T] = am.to(torch.int32)
        return km

    def _prefill(self, embeds: torch.Tensor, cache: VclaCache, key_mask, all_logits: bool, taps: Optional[dict] = None, _persistent: bool = False):
        lib = _lib.load()
        t = self.config.text_config
        B, T, D = embeds.shape
        V = t["vocab_size"]
        pos0 = cache.length
        if pos0 + T > cache.ctx_max:
            raise ValueError(f"sequence length {pos0 + T} exceeds the KV cache capacity {cache.ctx_max}")
        lshape = (B, T, V) if all_logits else (B, V)
        logits = self._typed_buf("gen_logits", lshape, torch.float32) if _persistent else torch.empty(lshape, dtype=torch.float32, device=self._device)
        nbytes = lib.vcla_llama_workspace_bytes(self._ctx, B, T)
        ws = self._buf("llama", nbytes)
        tap = None
        if taps is not None:
            tap = torch.empty(t["num_hidden_layers"] + 1, B, T, D, dtype=self._dtype, device=self._device)
        with torch.cuda.device(self._device):
            _lib.check(lib.vcla_llama_prefill(self._ctx, embeds.data_ptr(), B, T, pos0, cache.kv.data_ptr(), cache.ctx_max,
                                              _lib.ptr(key_mask), logits.data_ptr(), int(all_logits), ws.data_ptr(),
                                              ws.numel(), _lib.ptr(tap), _lib.stream_ptr()))
        cache.length = pos0 + T
        if taps is not None:
            for i in range(t["num_hidden_layers"]):
                taps[f"llama_layer{i}"] = tap[i]
            if all_logits:
                taps["final_norm"] = tap[-1]
            taps["logits"] = logits
        return logits

    def _check_decode_status(self, B: int, ws: torch.Tensor) -> None:
        """At B = 1 the decode steps of the bf16 mode are persistent launches whose workgroups wait on each other with BOUNDED spins
        (csrc/decode_engine.hip); a wait that ran out leaves a code in the workspace and the tokens are garbage -- raise instead of returning
        them.  One stream synchronisation, at a point where the caller is about to read the tokens anyway."""
        if B == 1:
            with torch.cuda.device(self._device):
                _lib.check(_lib.load().vcla_llama_decode_status(self._ctx, B, ws.data_ptr(), ws.numel(), _lib.stream_ptr()))

    # ------------------------------------------------------------------ forward (parity entry)
    def forward(self, input_ids: Optional[torch.LongTensor] = None, pixel_values: Optional[torch.Tensor] = None,
                attention_mask: Optional[torch.Tensor] = None, position_ids: Optional[torch.LongTensor] = None,
                past_key_values: Optional[VclaCache] = None, labels: Optional[torch.LongTensor] = None,
                use_cache: Optional[bool] = None, return_loss: Optional[bool] = None,
                return_dict: Optional[bool] = None, taps: Optional[dict] = None, **kwargs):
        """Same contract as the reference forward (modeling_visualcla.py:264-330): logits [B, T, V] (fp32),
        optional loss, KV cache handle.  `position_ids` is accepted and ignored, as in the reference (:269 vs :321-328)."""
        from transformers.modeling_outputs import CausalLMOutputWithPast
        input_ids = self._prepare_ids(input_ids, pixel_values)
        B = input_ids.shape[0]
        Q = self.config.visual_resampler_config["num_query_tokens"] if pixel_values is not None else 0
        n_extra = Q if self.image_at_head else 0        # positions the image adds in front of the text (modeling_visualcla.py:290-291)
        if labels is not None:
            if n_extra:                                 # the reference's placement: Q ignore-labels after position 0 (:313-315)
                labels = torch.cat([labels[:, :1], torch.full((B, n_extra), -100, dtype=labels.dtype, device=labels.device),
                                    labels[:, 1:]], dim=1)
            if tuple(labels.shape) != (B, input_ids.shape[1] + n_extra):
                raise ValueError(f"labels of shape {tuple(labels.shape)} do not match the {input_ids.shape[1] + n_extra}-position sequence")
        # one host sync (ids, slots, mask, labels); raises before any kernel runs
        img_pos, am = self._check_request(input_ids, attention_mask, Q, for_generate=False, labels=labels)
        img = self.embed_images(pixel_values, taps) if pixel_values is not None else None
        embeds, extra = self._embed(input_ids, img, img_pos)
        if taps is not None:
            taps["spliced_embeds"] = embeds
        T = embeds.shape[1]
        cache = past_key_values
        if cache is None:
            cap = T if not use_cache else min(self.config.text_config["max_position_embeddings"], (T + 512 + 63) // 64 * 64)
            cache = self._new_cache(B, cap)
        key_mask = self._key_mask(am, B, cache.length + T, cache.ctx_max)
        logits = self._prefill(embeds, cache, key_mask, all_logits=True, taps=taps)
        loss = None
        if labels is not None:
            lab = labels.to(self._device)                  # shape and vocabulary range were checked up front (_check_request)
            with torch.cuda.device(self._device):
                loss = _lib.causal_lm_loss(logits, lab)       # shifted cross-entropy, HF's ForCausalLMLoss (vcla_causal_lm_loss)
        out = CausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=cache if use_cache else None)
        if return_dict is False:
            return tuple(x for x in (loss, logits, out.past_key_values) if x is not None)
        return out

    __call__ = forward

    # ------------------------------------------------------------------ generate
    def _resolve_generation_config(self, generation_config, kwargs):
        from transformers import GenerationConfig
        import copy
        gc = copy.deepcopy(generation_config or self.generation_config or GenerationConfig())
        if generation_config is not None and self.generation_config is not None:
            # HF generate(): every field the caller's config leaves at None comes from the model's own generation config -- the text model's
            # generation_config.json, since the reference calls text_model.generate (hf:generation/utils.py _prepare_generation_config:
            # `update(**self.generation_config.to_dict(), defaults_only=True, allow_custom_entries=True)`).  Without this chat() under the
            # reference's DEFAULT_GENERATION_CONFIG (eos_token_id=None) would never stop at </s>.  Explicit keyword arguments below still
            # override (eos_token_id=None in a call disables the stop, as the benchmark does).
            for k, v in self.generation_config.to_dict().items():
                if k.startswith("_") or k == "transformers_version" or v is None:
                    continue
                if getattr(gc, k, None) is None:
                    setattr(gc, k, copy.deepcopy(v))
        # ... then transformers' global defaults for what is still None (same function; max_length / min_length stay None = "not set", which is what
        # the length rules in logits_processors.py test).  Note top_k = 50: sampling without an explicit top_k is top-50 sampling upstream.
        for k, v in _HF_GLOBAL_GENERATION_DEFAULTS.items():
            if getattr(gc, k, None) is None:
                setattr(gc, k, v)
        for k in list(kwargs.keys()):
            if hasattr(gc, k) and k not in ("input_ids", "pixel_values", "attention_mask"):
                setattr(gc, k, kwargs.pop(k))
        return gc

    @staticmethod
    def _eos_list(gc) -> List[int]:
        e = gc.eos_token_id
        if e is None:
            return []
        return [int(x) for x in e] if isinstance(e, (list, tuple)) else [int(e)]

    def _processors(self, gc, extra_processors, prompt_len: int = 0, n_new: Optional[int] = None, prefix_allowed_tokens_fn=None):
        """the request's logits processors: transformers' classes in transformers' order (visualcla/logits_processors.py)"""
        from .logits_processors import build_logits_processors
        return build_logits_processors(gc, self._eos_list(gc), self._device, prompt_len=prompt_len, n_new=n_new, extra=extra_processors,
                                       prefix_allowed_tokens_fn=prefix_allowed_tokens_fn)

    def _device_sampling(self, gc, n_new: int, prompt_len: int = 0):
        """kwargs for _lib.sample_args when the generation config maps onto the on-device sampler (next row N2), else None"""
        from .logits_processors import min_token_floor, needs_host_processors
        if needs_host_processors(gc):
            return None
        eos = self._eos_list(gc)
        mnt = min_token_floor(gc, prompt_len) if eos else 0       # min_new_tokens, or min_length less the prompt: the same eos mask
        if len(eos) > _lib.SAMPLE_MAX_EOS and mnt:
            return None
        if n_new > _lib.SAMPLE_MAX_HIST or self.config.text_config["vocab_size"] > _lib.SAMPLE_MAX_VOCAB:
            return None
        kw = dict(repetition_penalty=gc.repetition_penalty if gc.repetition_penalty is not None else 1.0,
                  no_repeat_ngram_size=gc.no_repeat_ngram_size or 0, min_new_tokens=mnt, eos_ids=eos if mnt else ())
        if gc.do_sample:
            k = gc.top_k if gc.top_k is not None else 0
            if not 1 <= k <= _lib.SAMPLE_MAX_TOP_K:
                return None                                   # top_k off / huge: full-vocabulary sort, host path
            kw.update(temperature=gc.temperature if gc.temperature is not None else 1.0, top_k=k,
                      top_p=gc.top_p if gc.top_p is not None else 1.0)
        return kw

    @torch.no_grad()
    def generate(self, input_ids=None, pixel_values=None, attention_mask=None, generation_config=None,
                 logits_processor=None, stopping_criteria=None, prefix_allowed_tokens_fn=None, synced_gpus=False,
                 use_graph: Optional[bool] = None, device_sampling: Optional[bool] = None, **kwargs):
        """Same contract as the reference generate (modeling_visualcla.py:334-392): returns the NEW tokens only,
        LongTensor [B, n_new] (what HF generate returns when driven by inputs_embeds).  Greedy decoding without
        callbacks runs entirely on the device (argmax feeds the next step; optional hipGraph replay); so does sampling /
        greedy with HF's standard processors (repetition penalty, no-repeat-ngram, min-new-tokens, temperature, top-k <= 256,
        top-p) through vcla_sample, drawing from torch.rand(n_new, B) of the device generator.  Custom logits processors,
        stopping criteria (streaming), `prefix_allowed_tokens_fn` or top_k = 0 take the host-driven path (HF processors +
        torch.multinomial)."""
        from .logits_processors import refuse_unsupported
        gc = self._resolve_generation_config(generation_config, kwargs)
        refuse_unsupported(gc, kwargs)                            # nothing the caller switched on is dropped silently
        nb = int(gc.num_beams or 1)
        if nb > 1 and gc.do_sample:
            raise ValueError("beam search is implemented for do_sample=False (beam SAMPLING draws without replacement from an implementation-defined "
                             "stream upstream); pass do_sample=False with num_beams > 1")
        if nb > 1 and device_sampling:
            raise ValueError("device_sampling=True cannot be combined with num_beams > 1: beam search runs HF's bookkeeping on host-driven decode steps")
        if nb > 1 and getattr(gc, "max_time", None) is not None:
            # HF applies MaxTimeCriteria inside beam search too; the host bookkeeping here has no early-exit hook for it -- refuse by name rather than drop it
            raise ValueError("max_time is not implemented for num_beams > 1 (beam search here runs to max_new_tokens or until every beam is finished); "
                             "use num_beams=1 or leave max_time unset")
        if nb == 1 and (gc.num_return_sequences or 1) != 1 and not gc.do_sample:
            # HF's wording (generation/configuration_utils.py validate): several returned sequences need beams or sampling
            raise ValueError(f"Greedy methods without beam search do not support `num_return_sequences` different than 1 (got {gc.num_return_sequences}).")
        # prefix_allowed_tokens_fn: the reference forwards it to HF generate (modeling_visualcla.py:382-391), which turns it into a processor placed
        # among the configured ones (it sees the NEW tokens only, as every processor does when HF is driven by inputs_embeds).  Host-driven step path.
        t = self.config.text_config
        input_ids = self._prepare_ids(input_ids, pixel_values)
        B = input_ids.shape[0]
        if use_graph is None:
            use_graph = os.environ.get("VCLA_DECODE_GRAPH", "1") != "0"
        # hipGraph capture (decode loop; vision stack and prefill inside the engine) is illegal on the legacy default stream: the
        # whole request hops onto a side stream, and its stage buffers are persistent so that every call presents the same addresses
        cur_stream = torch.cuda.current_stream(self._device)
        side = None
        if use_graph and cur_stream.cuda_stream == 0:
            if getattr(self, "_side_stream", None) is None:
                self._side_stream = torch.cuda.Stream(device=self._device)
            side = self._side_stream
            side.wait_stream(cur_stream)
        with torch.cuda.device(self._device), torch.cuda.stream(side if side is not None else cur_stream):
            toks = self._generate_on_stream(gc, input_ids, pixel_values, attention_mask, logits_processor, stopping_criteria, use_graph,
                                            device_sampling, prefix_allowed_tokens_fn)
        if side is not None:
            cur_stream.wait_stream(side)
        return toks

    def _beam_generate(self, gc, embeds, am, T, n_new, ctx_max, eos, logits_processor, stopping_criteria, prefix_fn=None):
        """num_beams > 1 (the reference forwards it to HF generate, modeling_visualcla.py:382-391).  HF repeats every prompt num_beams times and prefills all
        B * num_beams rows; the beams of a prompt are identical until the first step, so here the vision stack and the PREFILL run once per prompt (B rows)
        and the prompt's K / V rows and first logits are broadcast to its beams (one copy of T positions).  Decode steps run on the B * num_beams rows through
        the same kernels as any batch of that size (host-driven steps).  Between steps the cache rows follow the surviving beams (HF's `reorder_cache`): a
        beam's parent is always a beam of the SAME prompt (beam_search.py: beam_rows = group offset + index within the group), whose first T positions are
        that prompt's -- only the generated positions [T, pos) are gathered, O(generated) bytes per step instead of O(context).  Bookkeeping:
        visualcla/beam_search.py.  Returns [B * num_return_sequences, n] new tokens."""
        from .beam_search import beam_search
        lib = _lib.load()
        t = self.config.text_config
        nb = int(gc.num_beams)
        B = embeds.shape[0]
        rows = B * nb
        cache1 = self._new_cache(B, ctx_max)
        first = self._prefill(embeds, cache1, self._key_mask(am, B, T, ctx_max), all_logits=False).repeat_interleave(nb, dim=0)
        cache = self._new_cache(rows, ctx_max)
        cache.kv.view(cache.kv.shape[0], 2, B, nb, *cache.kv.shape[3:])[:, :, :, :, :, :T].copy_(cache1.kv[:, :, :, None, :, :T])
        del cache1
        key_mask = self._key_mask(None if am is None else am.repeat_interleave(nb, dim=0), rows, T, ctx_max)
        ws = self._buf("llama", lib.vcla_llama_workspace_bytes(self._ctx, rows, 1))
        step_logits = torch.empty(rows, t["vocab_size"], dtype=torch.float32, device=self._device)
        ident = torch.arange(rows, device=self._device)
        state = {"pos": T}

        def step(tokens, beam_rows):
            pos = state["pos"]
            if pos > T and not torch.equal(beam_rows, ident):        # the generated positions of every cache row follow its beam
                filled = cache.kv[:, :, :, :, T:pos, :]
                filled.copy_(filled.index_select(2, beam_rows))
            with torch.cuda.device(self._device):
                _lib.check(lib.vcla_llama_decode_step(self._ctx, tokens.contiguous().data_ptr(), rows, pos, None, 0, cache.kv.data_ptr(), ctx_max,
                                                      _lib.ptr(key_mask), step_logits.data_ptr(), None, ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
            state["pos"] = pos + 1
            return step_logits
        procs = self._processors(gc, logits_processor, T, n_new, prefix_fn)       # the configured + the caller's processors, applied to log-probs as HF does
        return beam_search(first, step, B, nb, n_new, eos_ids=eos, pad_token_id=gc.pad_token_id, length_penalty=float(gc.length_penalty if gc.length_penalty is not None else 1.0),
                           early_stopping=gc.early_stopping if gc.early_stopping is not None else False,
                           num_return_sequences=int(gc.num_return_sequences or 1), processors=procs,
                           stopping_criteria=list(stopping_criteria) if stopping_criteria else ())

    def _generate_on_stream(self, gc, input_ids, pixel_values, attention_mask, logits_processor, stopping_criteria, use_graph, device_sampling,
                            prefix_fn=None):
        lib = _lib.load()
        t = self.config.text_config
        B = input_ids.shape[0]
        persistent = bool(use_graph)
        Q = self.config.visual_resampler_config["num_query_tokens"] if pixel_values is not None else 0
        img_pos, am = self._check_request(input_ids, attention_mask, Q, for_generate=True)       # one host sync; raises before any kernel runs
        img = self.embed_images(pixel_values, _persistent=persistent) if pixel_values is not None else None
        embeds, extra = self._embed(input_ids, img, img_pos, _persistent=persistent)
        nrs = int(gc.num_return_sequences or 1)
        if nrs > 1 and int(gc.num_beams or 1) == 1:
            # sampling with several returned sequences: as HF does with inputs_embeds, every prompt's spliced embeddings are repeated num_return_sequences
            # times AFTER the vision stack ran once per image (rows b * nrs .. + nrs - 1 = prompt b) and each row draws on its own.  Fresh buffers per call:
            # the graph-replayed stages are keyed on persistent addresses.
            embeds = embeds.repeat_interleave(nrs, dim=0).contiguous()
            am = None if am is None else am.repeat_interleave(nrs, dim=0)
            B, persistent, use_graph = B * nrs, False, False
        T = embeds.shape[1]
        max_pos = t["max_position_embeddings"]
        from .logits_processors import new_token_budget
        n_new = min(new_token_budget(gc, T), max_pos - T)         # max_new_tokens, else max_length less the prompt, else 20 (HF's rules for inputs_embeds)
        if n_new <= 0:
            raise ValueError(f"prompt of {T} tokens leaves no room under max_position_embeddings={max_pos}")
        ctx_max = min(max_pos, (T + n_new + 63) // 64 * 64)
        eos = self._eos_list(gc)
        if int(gc.num_beams or 1) > 1:
            return self._beam_generate(gc, embeds, am, T, n_new, ctx_max, eos, logits_processor, stopping_criteria, prefix_fn)
        cache = self._new_cache(B, ctx_max, _persistent=persistent)
        key_mask = self._key_mask(am, B, T, ctx_max)
        logits = self._prefill(embeds, cache, key_mask, all_logits=False, _persistent=persistent)

        pad_id = gc.pad_token_id if gc.pad_token_id is not None else (eos[0] if eos else 0)
        procs = self._processors(gc, logits_processor, T, n_new, prefix_fn)
        criteria = list(stopping_criteria) if stopping_criteria else []
        if getattr(gc, "max_time", None) is not None:
            from transformers.generation.stopping_criteria import MaxTimeCriteria
            criteria.append(MaxTimeCriteria(max_time=gc.max_time))
        ws = self._buf("llama", lib.vcla_llama_workspace_bytes(self._ctx, B, 1))
        stream = _lib.stream_ptr()

        plain_greedy = not procs and not gc.do_sample
        samp_kw = None
        if not plain_greedy and not logits_processor and prefix_fn is None and device_sampling is not False:
            samp_kw = self._device_sampling(gc, n_new, T)
        if device_sampling and samp_kw is None and not plain_greedy:
            raise ValueError("device_sampling=True but the generation config needs HF's processors on the host-driven path "
                             "(custom logits_processor, or top_k outside [1, %d])" % _lib.SAMPLE_MAX_TOP_K)
        fast = not criteria and (plain_greedy or samp_kw is not None)
        if fast:
            # ---- device-resident loop: argmax, or the on-device sampler, feeds the next step
            out = self._typed_buf("gen_out", (n_new, B), torch.int64) if persistent else torch.empty(n_new, B, dtype=torch.int64, device=self._device)
            samp = None
            if samp_kw is not None:
                self._uniforms = torch.rand(n_new, B, device=self._device) if gc.do_sample else None
                samp = _lib.sample_args(uniforms=self._uniforms, history=out, **samp_kw)
                with torch.cuda.device(self._device):
                    first = _lib.sample(logits, samp, n_hist=0)
            else:
                with torch.cuda.device(self._device):
                    first = _lib.argmax(logits)
            out[0] = first
            done_at = n_new
            step, chunk = 1, (n_new if not eos else 32)
            self._pos_dev.zero_()
            eos_t = torch.tensor(eos, device=self._device) if eos else None
            # position of decode step i's input token = T + *pos_dev; the counter runs 0,1,2,... across chunks so
            # the captured graph (keyed on buffers + pos0) is reused for the whole generate() call
            while step < n_new:
                k = min(chunk, n_new - step)
                _lib.check(lib.vcla_llama_decode_loop_sampled(
                    self._ctx, out[step - 1].data_ptr(), B, T, self._pos_dev.data_ptr(), k, cache.kv.data_ptr(), ctx_max,
                    _lib.ptr(key_mask), out[1:].data_ptr(), ws.data_ptr(), ws.numel(), int(use_graph),
                    C.byref(samp) if samp is not None else None, 1, _lib.stream_ptr()))
                step += k
                if eos_t is not None and bool(torch.isin(out[:step], eos_t).any(dim=0).all()):
                    done_at = step
                    break
            toks = out[:min(step, done_at)].t().contiguous()
            self._check_decode_status(B, ws)
            if eos:
                is_eos = torch.isin(toks, torch.tensor(eos, device=self._device))
                after = (is_eos.cumsum(dim=1) - is_eos.int()) > 0
                toks = torch.where(after, torch.full_like(toks, pad_id), toks)
                keep = int((~after).any(dim=0).sum())
                toks = toks[:, :max(keep, 1)]
            return toks

        # ---- general path: host-driven, one decode step per token (stopping criteria / streaming callbacks see every token).
        # Token selection is still ONE kernel per step (vcla_argmax / vcla_sample) unless the caller brought its own
        # logits processors or a config the device sampler does not cover: then HF's processor classes + torch.multinomial.
        hist = torch.empty(n_new, B, dtype=torch.int64, device=self._device)       # step-major, what the sampler reads
        generated = hist[:0].t()
        done = torch.zeros(B, dtype=torch.bool, device=self._device)
        eos_t = torch.tensor(eos, device=self._device) if eos else None
        step_logits = torch.empty(B, t["vocab_size"], dtype=torch.float32, device=self._device)
        dev_select = not logits_processor and prefix_fn is None and device_sampling is not False and (plain_greedy or samp_kw is not None)
        samp = None
        if dev_select and not plain_greedy:
            self._uniforms = torch.rand(n_new, B, device=self._device) if gc.do_sample else None
            samp = _lib.sample_args(uniforms=self._uniforms, history=hist, **samp_kw)
        for step in range(n_new):
            scores = logits
            if dev_select:
                with torch.cuda.device(self._device):
                    nxt = _lib.argmax(logits) if plain_greedy else _lib.sample(logits, samp, n_hist=step)
            else:
                for p in procs:
                    scores = p(generated, scores)
                if gc.do_sample:
                    probs = torch.softmax(scores, dim=-1)
                    nxt = torch.multinomial(probs, num_samples=1)[:, 0]
                else:
                    nxt = scores.argmax(dim=-1)
            nxt = torch.where(done, torch.full_like(nxt, pad_id), nxt)
            hist[step] = nxt
            generated = hist[:step + 1].t()
            if eos_t is not None:
                done = done | torch.isin(nxt, eos_t)
            stop = False
            for crit in criteria:
                r = crit(generated, scores)
                if isinstance(r, torch.Tensor):
                    done = done | r.to(done.device).bool()
                elif r:
                    stop = True
            if stop or bool(done.all()) or step == n_new - 1:
                break
            with torch.cuda.device(self._device):
                _lib.check(lib.vcla_llama_decode_step(self._ctx, nxt.contiguous().data_ptr(), B, T + step, None, 0,
                                                      cache.kv.data_ptr(), ctx_max, _lib.ptr(key_mask),
                                                      step_logits.data_ptr(), None, ws.data_ptr(), ws.numel(), stream))
            logits = step_logits
        self._check_decode_status(B, ws)
        return generated.contiguous()
