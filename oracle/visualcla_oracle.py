"""CPU oracle for the VisualCLA multimodal forward / generate path.

TEST INFRASTRUCTURE ONLY.  This module is a from-formula restatement (plain
torch-CPU tensor arithmetic, fp32 by default) of the algorithm the reference
runs on its hot path.  It exists so the HIP path can be checked against it:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product package never does.

Parity pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 8c), so the oracle is pinned against OUTPUTS OF THE REFERENCE
ITSELF: ``oracle/make_golden.py`` imports the reference's own model code from
``/root/reference`` (with a 4-item in-memory shim for transformers 5.x), runs it
on seeded inputs and commits per-stage tensors under ``tests/golden/``.
``tests/test_oracle_vs_golden.py`` checks every function below against them.

What each function follows (paths relative to /root/reference unless they
start with ``hf:`` = transformers/models/... of the pinned third-party
dependency ``transformers`` (setup.py:13 ``>= 4.29.0``; 5.15.0 installed)):

  clip_embeddings        hf:clip/modeling_clip.py:202-219 (+ pre_layrnorm :642)
  clip_encoder_layer     hf:clip/modeling_clip.py:353-384, attn :259-335, mlp :338-350
  vision_tower           models/visualcla/modeling_visualcla.py:283-284 / :349-350
  resampler_forward      models/visualcla/modeling_visual_resampler.py:609-737
  resampler_layer        ... :360-416, :280-328, :132-263, :266-277, :331-357
  image_projection       models/visualcla/modeling_visualcla.py:102,288,354
  embed_and_splice       models/visualcla/modeling_visualcla.py:280,292-305 / :346,358-370
  llama_rmsnorm          hf:llama/modeling_llama.py:53-71
  llama_rope_tables      hf:llama/modeling_llama.py:73-127
  apply_rope             hf:llama/modeling_llama.py:130-160
  llama_layer            hf:llama/modeling_llama.py:163-176, 191-214, 217-325
  llama_forward          hf:llama/modeling_llama.py:347-418
  lm_head                hf:llama/modeling_llama.py:478-480
  visualcla_forward      models/visualcla/modeling_visualcla.py:264-330
  causal_lm_loss         hf:loss/loss_utils.py ForCausalLMLoss (labels branch of :321-328)
  visualcla_generate     models/visualcla/modeling_visualcla.py:334-392 +
                         hf:generation/utils.py greedy loop (argmax of fp32 last-token logits)

Weights are a flat ``dict[str, Tensor]`` keyed by the reference's state_dict
names (``text_model.`` / ``vision_model.`` / ``visual_resampler.`` /
``image_projection_layer.`` prefixes, scripts/merge_llama_with_visualcla_lora.py:95-96).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------
@dataclass
class VisionCfg:
    hidden_size: int = 1024
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    intermediate_size: int = 4096
    patch_size: int = 14
    image_size: int = 224
    num_channels: int = 3
    layer_norm_eps: float = 1e-5
    hidden_act: str = "quick_gelu"

    @property
    def num_patches(self) -> int:
        return (self.image_size // self.patch_size) ** 2

    @property
    def num_positions(self) -> int:
        return self.num_patches + 1


@dataclass
class ResamplerCfg:
    hidden_size: int = 1024
    num_hidden_layers: int = 6
    num_attention_heads: int = 16
    intermediate_size: int = 4096
    num_query_tokens: int = 64
    layer_norm_eps: float = 1e-12
    hidden_act: str = "gelu"


@dataclass
class TextCfg:
    hidden_size: int = 4096
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    intermediate_size: int = 11008
    vocab_size: int = 49958
    rms_norm_eps: float = 1e-6
    rope_theta: float = 10000.0
    max_position_embeddings: int = 2048

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


@dataclass
class OracleCfg:
    vision: VisionCfg = field(default_factory=VisionCfg)
    resampler: ResamplerCfg = field(default_factory=ResamplerCfg)
    text: TextCfg = field(default_factory=TextCfg)
    img_start_token_id: int = 49954
    img_end_token_id: int = 49955
    img_token_id: int = 49957


def cfg_7b() -> OracleCfg:
    """VisualCLA-7B shapes ('7B shapes')."""
    return OracleCfg()


def cfg_tiny() -> OracleCfg:
    """A small config that keeps every structural feature (head_dim 64 vision /
    128 text would be too large; tiny uses 32/32) for fast CPU tests."""
    return OracleCfg(
        vision=VisionCfg(hidden_size=128, num_hidden_layers=2, num_attention_heads=4,
                         intermediate_size=256, patch_size=14, image_size=56),
        resampler=ResamplerCfg(hidden_size=128, num_hidden_layers=2, num_attention_heads=4,
                               intermediate_size=256, num_query_tokens=8),
        text=TextCfg(hidden_size=256, num_hidden_layers=2, num_attention_heads=4,
                     intermediate_size=512, vocab_size=320, max_position_embeddings=256),
        img_start_token_id=316, img_end_token_id=317, img_token_id=319,
    )


def cfg_small() -> OracleCfg:
    """Medium config with the 7B head sizes (vision d=64, text d=128) and
    non-power-of-two dims, so every kernel tile path is exercised."""
    return OracleCfg(
        vision=VisionCfg(hidden_size=256, num_hidden_layers=3, num_attention_heads=4,
                         intermediate_size=512, patch_size=14, image_size=112),
        resampler=ResamplerCfg(hidden_size=256, num_hidden_layers=2, num_attention_heads=4,
                               intermediate_size=512, num_query_tokens=16),
        text=TextCfg(hidden_size=512, num_hidden_layers=3, num_attention_heads=4,
                     intermediate_size=1408, vocab_size=1000, max_position_embeddings=512),
        img_start_token_id=996, img_end_token_id=997, img_token_id=999,
    )


# --------------------------------------------------------------------------
# deterministic synthetic weights / inputs
# --------------------------------------------------------------------------
def _round_bf16(t: Tensor) -> Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def make_weights(cfg: OracleCfg, seed: int = 0, dtype=torch.float32) -> Dict[str, Tensor]:
    """Random weights, bf16-rounded ONCE so that the oracle (fp32 upcast) and the
    HIP path (bf16 storage) hold bit-identical values.  Linear/Conv/Embedding
    ~N(0,.02) as `_init_weights` (modeling_visualcla.py:55-64,
    modeling_visual_resampler.py:545-559); biases, norm gains/biases, class /
    position / query embeddings are randomised too so that no term is degenerate
    (query_embeddding is zero-init in the reference, modeling_visual_resampler.py:587)."""
    g = torch.Generator().manual_seed(seed)
    W: Dict[str, Tensor] = {}

    def normal(name, *shape, std=0.02, mean=0.0):
        W[name] = _round_bf16(torch.randn(*shape, generator=g) * std + mean).to(dtype)

    v, r, t = cfg.vision, cfg.resampler, cfg.text
    p = "vision_model.vision_model."
    normal(p + "embeddings.class_embedding", v.hidden_size)
    normal(p + "embeddings.patch_embedding.weight", v.hidden_size, v.num_channels, v.patch_size, v.patch_size)
    normal(p + "embeddings.position_embedding.weight", v.num_positions, v.hidden_size)
    for ln in ("pre_layrnorm", "post_layernorm"):
        normal(p + ln + ".weight", v.hidden_size, std=0.1, mean=1.0)
        normal(p + ln + ".bias", v.hidden_size)
    for i in range(v.num_hidden_layers):
        q = f"{p}encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            normal(q + f"self_attn.{nm}.weight", v.hidden_size, v.hidden_size)
            normal(q + f"self_attn.{nm}.bias", v.hidden_size)
        for ln in ("layer_norm1", "layer_norm2"):
            normal(q + ln + ".weight", v.hidden_size, std=0.1, mean=1.0)
            normal(q + ln + ".bias", v.hidden_size)
        normal(q + "mlp.fc1.weight", v.intermediate_size, v.hidden_size)
        normal(q + "mlp.fc1.bias", v.intermediate_size)
        normal(q + "mlp.fc2.weight", v.hidden_size, v.intermediate_size)
        normal(q + "mlp.fc2.bias", v.hidden_size)

    p = "visual_resampler."
    W[p + "query_embeddding"] = _round_bf16(
        torch.randn(1, r.num_query_tokens, r.hidden_size, generator=g) * 0.02).to(dtype)
    for i in range(r.num_hidden_layers):
        q = f"{p}encoder.layer.{i}."
        for nm in ("query", "key", "value"):
            normal(q + f"crossattention.self.{nm}.weight", r.hidden_size, r.hidden_size)
            normal(q + f"crossattention.self.{nm}.bias", r.hidden_size)
        normal(q + "crossattention.output.dense.weight", r.hidden_size, r.hidden_size)
        normal(q + "crossattention.output.dense.bias", r.hidden_size)
        normal(q + "crossattention.output.LayerNorm.weight", r.hidden_size, std=0.1, mean=1.0)
        normal(q + "crossattention.output.LayerNorm.bias", r.hidden_size)
        normal(q + "intermediate.dense.weight", r.intermediate_size, r.hidden_size)
        normal(q + "intermediate.dense.bias", r.intermediate_size)
        normal(q + "output.dense.weight", r.hidden_size, r.intermediate_size)
        normal(q + "output.dense.bias", r.hidden_size)
        normal(q + "output.LayerNorm.weight", r.hidden_size, std=0.1, mean=1.0)
        normal(q + "output.LayerNorm.bias", r.hidden_size)
    # dead pooler (computed and discarded, modeling_visual_resampler.py:725) -- kept so
    # that the strict state-dict load at modeling_visualcla.py:173 succeeds
    normal(p + "pooler.dense.weight", r.hidden_size, r.hidden_size)
    normal(p + "pooler.dense.bias", r.hidden_size)

    normal("image_projection_layer.weight", t.hidden_size, v.hidden_size)
    normal("image_projection_layer.bias", t.hidden_size)

    p = "text_model."
    normal(p + "model.embed_tokens.weight", t.vocab_size, t.hidden_size)
    for i in range(t.num_hidden_layers):
        q = f"{p}model.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
            normal(q + f"self_attn.{nm}.weight", t.hidden_size, t.hidden_size)
        normal(q + "mlp.gate_proj.weight", t.intermediate_size, t.hidden_size)
        normal(q + "mlp.up_proj.weight", t.intermediate_size, t.hidden_size)
        normal(q + "mlp.down_proj.weight", t.hidden_size, t.intermediate_size)
        normal(q + "input_layernorm.weight", t.hidden_size, std=0.1, mean=1.0)
        normal(q + "post_attention_layernorm.weight", t.hidden_size, std=0.1, mean=1.0)
    normal(p + "model.norm.weight", t.hidden_size, std=0.1, mean=1.0)
    normal(p + "lm_head.weight", t.vocab_size, t.hidden_size)
    return W


def make_inputs(cfg: OracleCfg, batch: int, seq_len: int, n_prefix: Optional[int] = None,
                seed_pixels: int = 1, seed_ids: int = 2) -> Tuple[Tensor, Tensor, Tensor]:
    """Synthetic request batch: pixel_values ~N(0,1);
    input_ids = BOS + n_prefix random + <img> + Q x <img_token> + </img> + random tail;
    attention_mask all ones."""
    Q = cfg.resampler.num_query_tokens
    if n_prefix is None:
        n_prefix = max(0, min(23, seq_len - (Q + 3) - 1))
    n_tail = seq_len - (1 + n_prefix + 1 + Q + 1)
    if n_tail < 0:
        raise ValueError(f"seq_len {seq_len} too short for {Q} image tokens")
    g1 = torch.Generator().manual_seed(seed_pixels)
    g2 = torch.Generator().manual_seed(seed_ids)
    v = cfg.vision
    pixel_values = _round_bf16(torch.randn(batch, v.num_channels, v.image_size, v.image_size, generator=g1))
    hi = min(cfg.img_start_token_id, cfg.img_end_token_id, cfg.img_token_id)  # ordinary ids < special ids
    rows = []
    for _ in range(batch):
        pre = torch.randint(3, hi, (n_prefix,), generator=g2)
        tail = torch.randint(3, hi, (n_tail,), generator=g2)
        rows.append(torch.cat([
            torch.tensor([1]), pre, torch.tensor([cfg.img_start_token_id]),
            torch.full((Q,), cfg.img_token_id), torch.tensor([cfg.img_end_token_id]), tail]))
    input_ids = torch.stack(rows).to(torch.int64)
    attention_mask = torch.ones_like(input_ids)
    return pixel_values, input_ids, attention_mask


# --------------------------------------------------------------------------
# elementary ops
# --------------------------------------------------------------------------
def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    y = x @ w.to(x.dtype).t()
    if b is not None:
        y = y + b.to(x.dtype)
    return y


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    # torch.nn.LayerNorm: biased variance, statistics in fp32
    xf = x.float()
    mu = xf.mean(-1, keepdim=True)
    var = ((xf - mu) ** 2).mean(-1, keepdim=True)
    y = (xf - mu) * torch.rsqrt(var + eps) * w.float() + b.float()
    return y.to(x.dtype)


def quick_gelu(x: Tensor) -> Tensor:
    # hf:activations.py QuickGELUActivation: x * sigmoid(1.702 x)
    return x * torch.sigmoid(1.702 * x)


def gelu_erf(x: Tensor) -> Tensor:
    # ACT2FN["gelu"] = exact erf GELU (modeling_visual_resampler.py:336,342)
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _mha(q: Tensor, k: Tensor, v: Tensor, n_heads: int, scale: float,
         add_mask: Optional[Tensor] = None, softmax_fp32: bool = True) -> Tensor:
    """q [B,Tq,D], k/v [B,Tk,D] -> [B,Tq,D].  softmax(scale*QK^T + mask) V."""
    B, Tq, D = q.shape
    Tk = k.shape[1]
    d = D // n_heads
    qh = q.view(B, Tq, n_heads, d).transpose(1, 2)
    kh = k.view(B, Tk, n_heads, d).transpose(1, 2)
    vh = v.view(B, Tk, n_heads, d).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * scale
    if add_mask is not None:
        s = s + add_mask
    if softmax_fp32:
        p = torch.softmax(s.float(), dim=-1).to(q.dtype)
    else:
        p = torch.softmax(s, dim=-1)
    o = p @ vh
    return o.transpose(1, 2).reshape(B, Tq, D)


# --------------------------------------------------------------------------
# CLIP ViT (third-party arithmetic: transformers CLIPVisionModel)
# --------------------------------------------------------------------------
def clip_embeddings(pixel_values: Tensor, W: Dict[str, Tensor], cfg: VisionCfg) -> Tensor:
    """conv(patch, stride patch, no bias) -> flatten -> prepend class emb -> + pos emb
    (hf:clip/modeling_clip.py:202-219), then pre_layrnorm (:642)."""
    p = "vision_model.vision_model."
    B = pixel_values.shape[0]
    w = W[p + "embeddings.patch_embedding.weight"].to(pixel_values.dtype)
    x = F.conv2d(pixel_values, w, stride=cfg.patch_size)          # [B, D, g, g]
    x = x.flatten(2).transpose(1, 2)                               # [B, g*g, D]
    cls = W[p + "embeddings.class_embedding"].to(x.dtype).expand(B, 1, -1)
    x = torch.cat([cls, x], dim=1)
    x = x + W[p + "embeddings.position_embedding.weight"].to(x.dtype)[None]
    return layer_norm(x, W[p + "pre_layrnorm.weight"], W[p + "pre_layrnorm.bias"], cfg.layer_norm_eps)


def clip_encoder_layer(x: Tensor, W: Dict[str, Tensor], cfg: VisionCfg, i: int) -> Tensor:
    q = f"vision_model.vision_model.encoder.layers.{i}."
    h = layer_norm(x, W[q + "layer_norm1.weight"], W[q + "layer_norm1.bias"], cfg.layer_norm_eps)
    qq = linear(h, W[q + "self_attn.q_proj.weight"], W[q + "self_attn.q_proj.bias"])
    kk = linear(h, W[q + "self_attn.k_proj.weight"], W[q + "self_attn.k_proj.bias"])
    vv = linear(h, W[q + "self_attn.v_proj.weight"], W[q + "self_attn.v_proj.bias"])
    d = cfg.hidden_size // cfg.num_attention_heads
    a = _mha(qq, kk, vv, cfg.num_attention_heads, d ** -0.5)
    a = linear(a, W[q + "self_attn.out_proj.weight"], W[q + "self_attn.out_proj.bias"])
    x = x + a
    h = layer_norm(x, W[q + "layer_norm2.weight"], W[q + "layer_norm2.bias"], cfg.layer_norm_eps)
    h = linear(h, W[q + "mlp.fc1.weight"], W[q + "mlp.fc1.bias"])
    h = quick_gelu(h)
    h = linear(h, W[q + "mlp.fc2.weight"], W[q + "mlp.fc2.bias"])
    return x + h


def vision_tower(pixel_values: Tensor, W: Dict[str, Tensor], cfg: VisionCfg,
                 taps: Optional[dict] = None) -> Tensor:
    """ViT last_hidden_state then post_layernorm on ALL tokens
    (modeling_visualcla.py:283-284; HF itself only normalises the CLS token)."""
    x = clip_embeddings(pixel_values, W, cfg)
    if taps is not None:
        taps["vit_embed"] = x
    for i in range(cfg.num_hidden_layers):
        x = clip_encoder_layer(x, W, cfg, i)
        if taps is not None:
            taps[f"vit_layer{i}"] = x
    p = "vision_model.vision_model.post_layernorm."
    x = layer_norm(x, W[p + "weight"], W[p + "bias"], cfg.layer_norm_eps)
    if taps is not None:
        taps["vit_post_ln"] = x
    return x


# --------------------------------------------------------------------------
# Resampler (in-repo arithmetic)
# --------------------------------------------------------------------------
def resampler_layer(lat: Tensor, img: Tensor, W: Dict[str, Tensor], cfg: ResamplerCfg, i: int) -> Tensor:
    """One VisualResamplerLayer: K/V source = cat([latents, image]) (:315); Q/K/V Linear+bias
    (:174,:187-188); QK^T / sqrt(d) (:213,:237) + all-zero masks (:240); softmax in the input
    dtype (:243); PV (:253); dense + residual + LayerNorm (:274-276); dense -> erf-GELU
    (:341-342); dense + residual + LayerNorm (:354-356)."""
    q = f"visual_resampler.encoder.layer.{i}."
    src = torch.cat([lat, img], dim=1)
    qq = linear(lat, W[q + "crossattention.self.query.weight"], W[q + "crossattention.self.query.bias"])
    kk = linear(src, W[q + "crossattention.self.key.weight"], W[q + "crossattention.self.key.bias"])
    vv = linear(src, W[q + "crossattention.self.value.weight"], W[q + "crossattention.self.value.bias"])
    d = cfg.hidden_size // cfg.num_attention_heads
    a = _mha(qq, kk, vv, cfg.num_attention_heads, 1.0 / math.sqrt(d), softmax_fp32=False)
    a = linear(a, W[q + "crossattention.output.dense.weight"], W[q + "crossattention.output.dense.bias"])
    h = layer_norm(a + lat, W[q + "crossattention.output.LayerNorm.weight"],
                   W[q + "crossattention.output.LayerNorm.bias"], cfg.layer_norm_eps)
    f = linear(h, W[q + "intermediate.dense.weight"], W[q + "intermediate.dense.bias"])
    f = gelu_erf(f)
    f = linear(f, W[q + "output.dense.weight"], W[q + "output.dense.bias"])
    return layer_norm(f + h, W[q + "output.LayerNorm.weight"], W[q + "output.LayerNorm.bias"], cfg.layer_norm_eps)


def resampler_forward(img: Tensor, W: Dict[str, Tensor], cfg: ResamplerCfg,
                      taps: Optional[dict] = None) -> Tensor:
    """query_embeddding expanded over the batch (:661), N layers; the tanh pooler (:725) is
    computed by the reference and discarded by its only caller -> not restated."""
    lat = W["visual_resampler.query_embeddding"].to(img.dtype).expand(img.shape[0], -1, -1)
    for i in range(cfg.num_hidden_layers):
        lat = resampler_layer(lat, img, W, cfg, i)
        if taps is not None:
            taps[f"resampler_layer{i}"] = lat
    return lat


def image_projection(x: Tensor, W: Dict[str, Tensor]) -> Tensor:
    return linear(x, W["image_projection_layer.weight"], W["image_projection_layer.bias"])


def image_embeds(pixel_values: Tensor, W: Dict[str, Tensor], cfg: OracleCfg,
                 taps: Optional[dict] = None) -> Tensor:
    """pixel_values -> [B, Q, text_hidden]: ViT + post-LN(all) + Resampler + projection
    (modeling_visualcla.py:283-288; second caller tgwebui embed_images visualcla.py:116-129)."""
    x = vision_tower(pixel_values, W, cfg.vision, taps)
    x = resampler_forward(x, W, cfg.resampler, taps)
    x = image_projection(x, W)
    if taps is not None:
        taps["image_embeds"] = x
    return x


# --------------------------------------------------------------------------
# embed + splice
# --------------------------------------------------------------------------
def embed_and_splice(input_ids: Tensor, img_emb: Optional[Tensor], W: Dict[str, Tensor],
                     cfg: OracleCfg, dtype=torch.float32, need_img_token: bool = True) -> Tensor:
    """embed_tokens gather, then (image_at_head=False branch) overwrite the Q rows after the
    <img> token with the image embeds; ValueError if the id at p0+Q+1 is not </img>
    (modeling_visualcla.py:292-305).  forward() leaves a row alone unless it holds BOTH an <img> and an <img_token>
    (:297); generate() only asks for the <img> (:363): `need_img_token=False`."""
    emb = W["text_model.model.embed_tokens.weight"].to(dtype)[input_ids]
    if img_emb is None:
        return emb
    out = []
    for ids, e, im in zip(input_ids, emb, img_emb):
        Q = im.shape[0]
        pos = torch.where(ids == cfg.img_start_token_id)[0]
        if len(pos) == 0 or (need_img_token and not bool((ids == cfg.img_token_id).any())):
            out.append(e)
            continue
        p0 = int(pos[0])
        if p0 + Q + 1 >= len(ids) or int(ids[p0 + Q + 1]) != cfg.img_end_token_id:
            raise ValueError(f"Num of patch ({Q}) is not equal to the length of pre-filled image patch tokens.")
        out.append(torch.cat([e[:p0 + 1], im.to(dtype), e[p0 + Q + 1:]], dim=0))
    return torch.stack(out, dim=0)


# --------------------------------------------------------------------------
# LLaMA decoder (third-party arithmetic: transformers LlamaForCausalLM)
# --------------------------------------------------------------------------
def llama_rmsnorm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    """fp32 internally, cast back to the input dtype BEFORE the weight multiply."""
    xf = x.float()
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return w.to(x.dtype) * xf.to(x.dtype)


def llama_rope_tables(positions: Tensor, head_dim: int, theta: float, dtype) -> Tuple[Tensor, Tensor]:
    """inv_freq = theta^(-2i/d); emb = cat(freqs, freqs); cos/sin in fp32 then cast."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    freqs = positions.float()[:, None] * inv_freq[None, :]
    emb = torch.cat([freqs, freqs], dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def _rotate_half(x: Tensor) -> Tensor:
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def apply_rope(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """x [B,H,T,d]; cos/sin [T,d]."""
    return x * cos[None, None] + _rotate_half(x) * sin[None, None]


def llama_layer(x: Tensor, W: Dict[str, Tensor], cfg: TextCfg, i: int, positions: Tensor,
                add_mask: Tensor, cache: Optional[List[Tuple[Tensor, Tensor]]]) -> Tensor:
    q = f"text_model.model.layers.{i}."
    B, T, D = x.shape
    H, d = cfg.num_attention_heads, cfg.head_dim
    h = llama_rmsnorm(x, W[q + "input_layernorm.weight"], cfg.rms_norm_eps)
    qq = linear(h, W[q + "self_attn.q_proj.weight"]).view(B, T, H, d).transpose(1, 2)
    kk = linear(h, W[q + "self_attn.k_proj.weight"]).view(B, T, H, d).transpose(1, 2)
    vv = linear(h, W[q + "self_attn.v_proj.weight"]).view(B, T, H, d).transpose(1, 2)
    cos, sin = llama_rope_tables(positions, d, cfg.rope_theta, x.dtype)
    qq, kk = apply_rope(qq, cos, sin), apply_rope(kk, cos, sin)
    if cache is not None:
        if cache[i] is not None:
            kk = torch.cat([cache[i][0], kk], dim=2)
            vv = torch.cat([cache[i][1], vv], dim=2)
        cache[i] = (kk, vv)
    s = (qq @ kk.transpose(2, 3)) * (d ** -0.5) + add_mask
    p = torch.softmax(s.float(), dim=-1).to(x.dtype)
    a = (p @ vv).transpose(1, 2).reshape(B, T, D)
    x = x + linear(a, W[q + "self_attn.o_proj.weight"])
    h = llama_rmsnorm(x, W[q + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
    g = linear(h, W[q + "mlp.gate_proj.weight"])
    u = linear(h, W[q + "mlp.up_proj.weight"])
    return x + linear(F.silu(g) * u, W[q + "mlp.down_proj.weight"])


def _causal_add_mask(T: int, past: int, key_mask: Optional[Tensor], dtype) -> Tensor:
    """additive mask [B or 1, 1, T, past+T]: causal + key padding (hf create_causal_mask)."""
    ctx = past + T
    qpos = torch.arange(past, past + T)[:, None]
    kpos = torch.arange(ctx)[None, :]
    allowed = (kpos <= qpos)[None, None]                       # [1,1,T,ctx]
    if key_mask is not None:
        allowed = allowed & key_mask.bool()[:, None, None, :ctx]
    return torch.zeros(allowed.shape, dtype=dtype).masked_fill(~allowed, torch.finfo(dtype).min)


def llama_forward(inputs_embeds: Tensor, W: Dict[str, Tensor], cfg: TextCfg,
                  attention_mask: Optional[Tensor] = None,
                  cache: Optional[List] = None, past_len: int = 0,
                  taps: Optional[dict] = None) -> Tensor:
    """LlamaModel.forward: position_ids = arange(T) + past (NOT mask-derived,
    modeling_llama.py:386-389), causal+padding mask, N layers, final RMSNorm."""
    B, T, _ = inputs_embeds.shape
    positions = torch.arange(past_len, past_len + T)
    add_mask = _causal_add_mask(T, past_len, attention_mask, inputs_embeds.dtype)
    x = inputs_embeds
    for i in range(cfg.num_hidden_layers):
        x = llama_layer(x, W, cfg, i, positions, add_mask, cache)
        if taps is not None:
            taps[f"llama_layer{i}"] = x
    return llama_rmsnorm(x, W["text_model.model.norm.weight"], cfg.rms_norm_eps)


def lm_head(h: Tensor, W: Dict[str, Tensor]) -> Tensor:
    return linear(h, W["text_model.lm_head.weight"])


# --------------------------------------------------------------------------
# composite model
# --------------------------------------------------------------------------
def _cast_weights(W: Dict[str, Tensor], dtype) -> Dict[str, Tensor]:
    return W if dtype == torch.float32 else {k: v.to(dtype) for k, v in W.items()}


def causal_lm_loss(logits: Tensor, labels: Tensor) -> Tensor:
    """hf:loss/loss_utils.py ForCausalLMLoss (what LlamaForCausalLM.forward(labels=...) returns, reached from
    modeling_visualcla.py:321-328): fp32 logits, labels shifted left by one (the last position gets -100), mean
    cross-entropy over the positions whose label is not -100."""
    lg = logits.float()
    shifted = F.pad(labels, (0, 1), value=-100)[..., 1:].contiguous()
    return F.cross_entropy(lg.reshape(-1, lg.shape[-1]), shifted.reshape(-1), ignore_index=-100, reduction="mean")


def visualcla_forward(input_ids: Tensor, pixel_values: Optional[Tensor], attention_mask: Optional[Tensor],
                      W: Dict[str, Tensor], cfg: OracleCfg, dtype=torch.float32,
                      taps: Optional[dict] = None, image_at_head: bool = False, labels: Optional[Tensor] = None,
                      cache: Optional[List] = None, past_len: int = 0):
    """VisualCLAModel.forward (modeling_visualcla.py:264-330): returns logits [B, T', V] in `dtype`, or
    (logits, loss) when `labels` is given.

    image_at_head=False (the production setting, modeling_utils.py:134): image embeds overwrite the <img_token> slots
    (:293-305), mask and labels pass through.  image_at_head=True (:290-291, :308-310, :313-315): embeds =
    emb[:, :2] ++ image ++ emb[:, 2:], mask = ones[B, Q] ++ mask (image columns FIRST), labels = labels[:, :1] ++ Q x -100 ++
    labels[:, 1:] -- the reference inserts the ignore labels one position EARLIER than the image embeds; restated as written.
    `cache` (a list of per-layer (k, v) or None entries, filled in place) + `past_len` = the past_key_values pass-through
    (:321-328): a later call with the next ids and the full-length mask continues the sequence."""
    W = _cast_weights(W, dtype)
    img = None
    if pixel_values is not None:
        img = image_embeds(pixel_values.to(dtype), W, cfg, taps)
    if img is not None and image_at_head:
        emb = W["text_model.model.embed_tokens.weight"].to(dtype)[input_ids]
        x = torch.cat([emb[:, :2], img.to(dtype), emb[:, 2:]], dim=1)
        B, Q = img.shape[:2]
        if attention_mask is not None:
            attention_mask = torch.cat([torch.ones(B, Q, dtype=attention_mask.dtype), attention_mask], dim=1)
        if labels is not None:
            labels = torch.cat([labels[:, :1], torch.full((B, Q), -100, dtype=labels.dtype), labels[:, 1:]], dim=1)
    else:
        x = embed_and_splice(input_ids, img, W, cfg, dtype)
    if taps is not None:
        taps["spliced_embeds"] = x
    h = llama_forward(x, W, cfg.text, attention_mask, cache, past_len, taps)
    if taps is not None:
        taps["final_norm"] = h
    logits = lm_head(h, W)
    if taps is not None:
        taps["logits"] = logits
    if labels is not None:
        return logits, causal_lm_loss(logits, labels)
    return logits


def visualcla_generate(input_ids: Tensor, pixel_values: Optional[Tensor], attention_mask: Tensor,
                       W: Dict[str, Tensor], cfg: OracleCfg, max_new_tokens: int,
                       eos_token_id: Optional[int] = None, dtype=torch.float32,
                       return_logits: bool = False, select_fn=None, image_at_head: bool = False):
    """`select_fn(logits [B, V], generated [B, step]) -> next ids [B]` replaces the argmax (sampling oracle, next row N2).

    VisualCLAModel.generate, greedy (do_sample=False): prefill over the spliced embeds with a
    KV cache, then one token per step = argmax of the fp32 last-position logits.  Returns the NEW
    tokens only, as HF does when called with inputs_embeds (modeling_utils.py:173-174)."""
    W = _cast_weights(W, dtype)
    img = image_embeds(pixel_values.to(dtype), W, cfg) if pixel_values is not None else None
    mask = attention_mask.clone()
    if img is not None and image_at_head:           # modeling_visualcla.py:356, :373-375
        emb = W["text_model.model.embed_tokens.weight"].to(dtype)[input_ids]
        x = torch.cat([emb[:, :2], img.to(dtype), emb[:, 2:]], dim=1)
        mask = torch.cat([torch.ones(img.shape[0], img.shape[1], dtype=mask.dtype), mask], dim=1)
    else:
        x = embed_and_splice(input_ids, img, W, cfg, dtype, need_img_token=False)
    B, T, _ = x.shape
    cache: List = [None] * cfg.text.num_hidden_layers
    h = llama_forward(x, W, cfg.text, mask, cache, 0)
    logits = lm_head(h[:, -1:, :], W)[:, 0].float()
    out, all_logits = [], [logits]
    done = torch.zeros(B, dtype=torch.bool)
    pad = eos_token_id if eos_token_id is not None else 0
    past = T
    for step in range(max_new_tokens):
        if select_fn is not None:
            nxt = select_fn(logits, torch.stack(out, dim=1) if out else torch.zeros(B, 0, dtype=torch.int64))
        else:
            nxt = logits.argmax(dim=-1)
        if eos_token_id is not None:
            nxt = torch.where(done, torch.full_like(nxt, pad), nxt)
            done = done | (nxt == eos_token_id)
        out.append(nxt)
        if step == max_new_tokens - 1 or bool(done.all()):
            break
        e = W["text_model.model.embed_tokens.weight"][nxt][:, None, :].to(dtype)
        mask = torch.cat([mask, torch.ones(B, 1, dtype=mask.dtype)], dim=1)
        h = llama_forward(e, W, cfg.text, mask, cache, past)
        past += 1
        logits = lm_head(h, W)[:, 0].float()
        all_logits.append(logits)
    toks = torch.stack(out, dim=1)
    return (toks, all_logits) if return_logits else toks
