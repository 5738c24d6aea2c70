"""Weight packing: reference state_dict names -> the HBM layout the HIP kernels read.

Layout (see DESIGN.md "Data layout in HBM"):
  * every Linear weight W [N, K] is stored bf16, K contiguous, rows zero-padded to a multiple of 128
    (one GEMM workgroup tile) so kernels never bounds-check the weight operand;
  * q/k/v projections are concatenated row-wise into one [3D, D] matrix (one GEMM instead of three);
  * the resampler's key/value projections are concatenated into [2D, D];
  * LLaMA gate/up projections are interleaved in blocks of 16 rows (16 gate, 16 up, ...) so the SwiGLU
    product is formed inside the GEMM epilogue from two accumulator tiles of the same lane;
  * the patch-embedding conv weight [D, 3, P, P] is flattened to [D, 3*P*P] and zero-padded in K to a
    multiple of 64 (588 -> 640) to match the im2col operand;
  * biases, norm gains, class/position embeddings and the RoPE cos/sin tables are fp32.

State-dict key names follow the reference (`text_model.` / `vision_model.` / `visual_resampler.` /
`image_projection_layer.`; scripts/merge_llama_with_visualcla_lora.py:95-96, models/visualcla/modeling_visualcla.py:172-179).
Both CLIP layouts are accepted: `vision_model.vision_model.*` (transformers 4.x, what the reference saves)
and the flat `vision_model.*` of transformers 5.x.
"""
from __future__ import annotations

import os
from typing import Callable, Dict

import torch


def pad_to(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _pack_w(w: torch.Tensor, device) -> torch.Tensor:
    """[N, K] -> bf16 [pad128(N), K] on device."""
    n, k = w.shape
    out = torch.zeros(pad_to(n, 128), k, dtype=torch.bfloat16, device=device)
    out[:n] = w.to(device=device, dtype=torch.bfloat16)
    return out


def _f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def interleave_gate_up(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """[I, K] x2 -> [2I, K] with rows (16 gate, 16 up) repeating."""
    i, k = gate.shape
    assert i % 16 == 0, "intermediate size must be a multiple of 16"
    return torch.stack([gate.reshape(i // 16, 16, k), up.reshape(i // 16, 16, k)], dim=1).reshape(2 * i, k)


def to_fragment_major(w_packed: torch.Tensor) -> torch.Tensor:
    """[N_pad, K] (row-major, N_pad % 16 == 0, K % 32 == 0) -> [N_pad/16, K/32, 64, 8]: every MFMA operand fragment
    (16 rows x 32 k of v_mfma_f32_16x16x32_bf16; lane = (k%32)//8 * 16 + n%16, 8 consecutive k per lane) is one contiguous
    1 KiB block and the fragments of one 16-row tile are contiguous along K.  A wave that streams a tile's weights for
    batch decode then reads one fully contiguous K/32-KiB region with plain 16-byte-per-lane loads (HBM page friendly),
    straight into MFMA operand registers -- no LDS transpose."""
    n, k = w_packed.shape
    assert n % 16 == 0 and k % 32 == 0
    return w_packed.view(n // 16, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(n // 16, k // 32, 64, 8)


def from_fragment_major(wf: torch.Tensor) -> torch.Tensor:
    nt, ks = wf.shape[0], wf.shape[1]
    return wf.view(nt, ks, 4, 16, 8).permute(0, 3, 1, 2, 4).contiguous().view(nt * 16, ks * 32)


def to_slab_major(x: torch.Tensor) -> torch.Tensor:
    """[rows, K] -> [K/64, rows, 64] (vcla_gemm_args.A_slab / W_slab / W_q8_slab): the K slab of any 8 consecutive rows (fp8: 16) is one
    contiguous 1 KiB -- what a single LDS-DMA wave instruction moves at full rate (profiles/r05_l2_intake.txt: ~50 B/clk/CU against ~19 for the
    8-row x 128-byte gather out of a row-major matrix).  Any element type."""
    r, k = x.shape
    assert k % 64 == 0
    return x.view(r, k // 64, 64).permute(1, 0, 2).contiguous()


def from_slab_major(xs: torch.Tensor) -> torch.Tensor:
    ks, r, _ = xs.shape
    return xs.permute(1, 0, 2).contiguous().view(r, ks * 64)


def quantize_fp8_rows(w_packed: torch.Tensor):
    """bf16 [N_pad, K] -> (uint8 [N_pad, K] holding OCP e4m3fn bits, fp32 per-row scale [N_pad]); W ~= q * scale[row]."""
    wf = w_packed.float()
    scale = (wf.abs().amax(dim=1) / 448.0).clamp_min(1e-20)
    q = (wf / scale[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).contiguous(), scale.float().contiguous()


def dequantize_fp8_rows(q: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    return q.view(torch.float8_e4m3fn).float() * scale[:, None].float()


def to_fragment_pair_major_fp8(q: torch.Tensor) -> torch.Tensor:
    """uint8 [N_pad, K] -> [N_pad/16, K/64, 64, 16]: lane (k%32)//8*16 + n%16 holds 8 elements of k-step 2j then 8 of
    k-step 2j+1 -> one 16-byte load per lane feeds two MFMA k-steps; a 16-row tile is contiguous along K."""
    n, k = q.shape
    assert n % 16 == 0 and k % 64 == 0
    return q.view(n // 16, 16, k // 64, 2, 4, 8).permute(0, 2, 4, 1, 3, 5).contiguous().view(n // 16, k // 64, 64, 16)


def add_fp8_copies(packed: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """fp8 twins of the LLaMA decode matrices: `<name>.q8` (row-major), `<name>.q8f` (fragment-pair-major), `<name>.s8`."""
    for name in list(packed.keys()):
        if (name.startswith("llama.l") and name.rsplit(".", 1)[-1] in FRAG_KEYS) or name == "llama.lm_head":
            q, sc = quantize_fp8_rows(packed[name])
            packed[name + ".q8"], packed[name + ".s8"] = q, sc
            packed[name + ".q8f"] = to_fragment_pair_major_fp8(q)
    return packed


def add_resampler_kv_all(packed: Dict[str, torch.Tensor], n_layers: int) -> Dict[str, torch.Tensor]:
    """`res.wkv_all` [L * 2D, D] / `res.bkv_all` [L * 2D]: the K/V projections of ALL resampler layers stacked, so that the
    image-token rows -- which do not depend on the latents (modeling_visual_resampler.py:315-316 projects cat([latents, image])
    with the same Linear in every layer) -- go through ONE GEMM [B*N, D] x [D, L*2D] before the layer loop instead of one per
    layer.  The per-layer tensors become views of the stacked one (no second copy).  Needs 2D % 128 == 0 (no row padding
    between the layers); otherwise nothing is added and the engine keeps the per-layer GEMMs."""
    ws = [packed[f"res.l{i}.wkv"] for i in range(n_layers)]
    bs = [packed[f"res.l{i}.bkv"] for i in range(n_layers)]
    if n_layers == 0 or ws[0].shape[0] != bs[0].shape[0]:      # rows were padded to a multiple of 128
        return packed
    wall, ball = torch.cat(ws, 0).contiguous(), torch.cat(bs, 0).contiguous()
    n = ws[0].shape[0]
    packed["res.wkv_all"], packed["res.bkv_all"] = wall, ball
    for i in range(n_layers):
        packed[f"res.l{i}.wkv"], packed[f"res.l{i}.bkv"] = wall[i * n:(i + 1) * n], ball[i * n:(i + 1) * n]
    return packed


FRAG_KEYS = ("wqkv", "wo", "wgu", "wd")     # LLaMA decode matrices that get a fragment-major twin (".f")


def add_fragment_copies(packed: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Adds `llama.l{i}.{wqkv,wo,wgu,wd}.f` and `llama.lm_head.f` (second copy, ~13 GB at 7B: HBM is 288 GB)."""
    for name in list(packed.keys()):
        if name.startswith("llama.l") and name.rsplit(".", 1)[-1] in FRAG_KEYS or name == "llama.lm_head":
            packed[name + ".f"] = to_fragment_major(packed[name])
    return packed


# ---- the persistent B = 1 decode engine's weight stream (csrc/decode_engine.hip) ------------------------------------------------------------
ENGINE_NCU, ENGINE_D, ENGINE_H, ENGINE_SLOT = 256, 4096, 32, 16384      # one workgroup per CU of an MI355X; LLaMA-7B hidden = 32 heads x 128


def engine_geometry(hidden: int, heads: int, inter: int, vocab: int, n_layers: int):
    """Slot counts of the engine stream (mirrors vcla_engine_geometry, csrc/decode_engine.hip); None when the model cannot run on the engine."""
    if hidden != ENGINE_D or heads != ENGINE_H or inter <= 0 or inter % ENGINE_NCU or not 0 < n_layers <= 120 or vocab <= 0:
        return None
    upc = inter // ENGINE_NCU
    gpc = (upc + 1) // 2
    if ENGINE_NCU * gpc > 5632:
        return None
    s_lm = (vocab + 2 * ENGINE_NCU - 1) // (2 * ENGINE_NCU)
    if min(upc, gpc, s_lm) < 8:          # every operator must hold the kernel's unconditional register preloads (2 x EG_PRE_MAX slots)
        return None
    slots_layer = 24 + 8 + upc + gpc
    return dict(upc=upc, gpc=gpc, s_lm=s_lm, slots_layer=slots_layer, slots_total=n_layers * slots_layer + s_lm)


def engine_down_kmap(inter: int) -> torch.Tensor:
    """down_proj's K dimension in GRANULE order: producer CU c publishes gpc granules of two activations each (units upc*c + 2j, + 2j + 1; the
    odd last one alone), and the consumers use the gathered mailbox as the input vector as it is.  kmap[k'] = real k, or -1 (zero column)."""
    upc = inter // ENGINE_NCU
    gpc = (upc + 1) // 2
    c = torch.arange(ENGINE_NCU).view(-1, 1, 1)
    j = torch.arange(gpc).view(1, -1, 1)
    e = torch.arange(2).view(1, 1, -1)
    local = 2 * j + e
    k = upc * c + local
    return torch.where(local < upc, k, torch.full_like(k, -1)).reshape(-1)


def add_engine_stream(packed: Dict[str, torch.Tensor], hidden: int, heads: int, inter: int, vocab: int, n_layers: int) -> Dict[str, torch.Tensor]:
    """`llama.engine.w` [slots_total][256 CUs][8192] bf16 + `llama.engine.g` [2 L + 1][4096] fp32: a THIRD copy of the LLaMA matrices (13.4 GB at
    7B; HBM is 288 GB) cut into 16-KiB slots in the order the persistent decode step consumes them -- CU c's loader wave reads slot g at
    (g * 256 + c) * 16 KiB, front to back, without knowing what an operator is.  SLOT-major: the 256 loaders advance together, so at any moment the chip
    reads ONE moving window of a few MiB, the access pattern of a plain copy.  (The first form of the round kept every CU's run contiguous -- 256 concurrent
    sequential streams 52 MB apart: 5 % slower on average and anywhere between 2.24 and 2.40 ms per step depending on WHERE the allocation landed in HBM,
    profiles/r06_engine_placement.txt; slot-major: 2.156 - 2.170 wherever it lands.)  Per layer and CU c (head h = c // 8, s = c % 8): 24 slots of wqkv
    (2 rows each: q, k, v rows h*128 + s*16 + 2j), 8 of wo (K-major: rows 16c .. 16c + 15 x 512 k), upc of wgu (gate row + up row of unit upc*c + j),
    gpc of wd (K-major: 16 rows x 512 k', k' in granule order, engine_down_kmap); after the layers s_lm slots of lm_head (rows 2 s_lm c + 2j).  Nothing is
    added when the geometry does not fit (the launch path serves the model)."""
    g = engine_geometry(hidden, heads, inter, vocab, n_layers)
    if g is None or "llama.lm_head" not in packed:
        return packed
    D, N, upc, gpc, s_lm, SL = ENGINE_D, ENGINE_NCU, g["upc"], g["gpc"], g["s_lm"], g["slots_layer"]
    dev = packed["llama.lm_head"].device
    stream = torch.zeros(N, g["slots_total"], 2 * D, dtype=torch.bfloat16, device=dev)
    kmap = engine_down_kmap(inter).to(dev)
    ksrc = kmap.clamp_min(0)
    kzero = (kmap < 0)
    for l in range(n_layers):
        d = f"llama.l{l}."
        blk = stream[:, l * SL:(l + 1) * SL]
        # wqkv [3D, D] -> [part, head, s, j, 2, D] -> [head, s, part, j, 2 D]
        blk[:, 0:24] = packed[d + "wqkv"][:3 * D].view(3, ENGINE_H, 8, 8, 2, D).permute(1, 2, 0, 3, 4, 5).reshape(N, 24, 2 * D)
        blk[:, 24:32] = packed[d + "wo"][:D].view(N, 16, 8, 512).permute(0, 2, 1, 3).reshape(N, 8, 16 * 512)     # K-major: slot j = 16 rows x k in [512 j, 512 j + 512)
        # wgu rows come in blocks (16 gate, 16 up): unit u = gate row u and up row u
        blk[:, 32:32 + upc] = packed[d + "wgu"][:2 * inter].view(inter // 16, 2, 16, D).permute(0, 2, 1, 3).reshape(N, upc, 2 * D)
        wd = packed[d + "wd"][:D].index_select(1, ksrc)
        wd[:, kzero] = 0
        blk[:, 32 + upc:32 + upc + gpc] = wd.view(N, 16, gpc, 512).permute(0, 2, 1, 3).reshape(N, gpc, 16 * 512)
    lm = packed["llama.lm_head"]
    rows = N * 2 * s_lm
    lmp = torch.zeros(rows, D, dtype=torch.bfloat16, device=dev)
    n = min(rows, lm.shape[0])
    lmp[:n] = lm[:n]
    lmp[vocab:] = 0
    stream[:, n_layers * SL:] = lmp.view(N, s_lm, 2 * D)
    gam = [packed[f"llama.l{l}.ln{i}.g"] for l in range(n_layers) for i in (1, 2)] + [packed["llama.norm.g"]]
    if os.environ.get("VCLA_ENGINE_LAYOUT", "slot") == "cu":        # A/B only (tools/debug/engine_variance.py): the first form of round 6, every CU's run contiguous
        packed["llama.engine.w.cu"] = stream
    else:
        packed["llama.engine.w"] = stream.permute(1, 0, 2).contiguous()
    packed["llama.engine.g"] = torch.stack(gam, 0).to(torch.float32).contiguous()
    return packed


def extend_position_embedding(state_dict: Dict[str, torch.Tensor], patch_size: int, after: int) -> Dict[str, torch.Tensor]:
    """Grow the CLIP position embedding for a larger input resolution (e.g. 224 -> 336 px, BASELINE configs[4]): the
    class-token row is kept, the g x g patch grid is bicubically interpolated to (after // patch_size)^2, position_ids are
    rebuilt.  In place.  This is the INTENDED behaviour of the reference helper of the same name
    (models/visualcla/modeling_visualcla.py:13-43); that helper is dead code and cannot run as written (it reshapes with the
    patch count instead of the grid side, :29/:33)."""
    pe_key = next(k for k in state_dict if k.endswith("vision_model.embeddings.position_embedding.weight"))
    pe = state_dict[pe_key]
    n_before, dim = pe.shape
    g0 = int(round((n_before - 1) ** 0.5))
    g1 = after // patch_size
    grid = pe[1:].reshape(g0, g0, dim).permute(2, 0, 1).unsqueeze(0).float()
    grid = torch.nn.functional.interpolate(grid, size=(g1, g1), mode="bicubic")
    new = torch.cat([pe[0:1].float(), grid.squeeze(0).permute(1, 2, 0).reshape(g1 * g1, dim)], dim=0).to(pe.dtype)
    state_dict[pe_key] = new
    for k in list(state_dict):
        if k.endswith("vision_model.embeddings.position_ids"):
            state_dict[k] = torch.arange(g1 * g1 + 1).unsqueeze(0)
    return state_dict


def fold_lora(state_dict: Dict[str, torch.Tensor], adapter: Dict[str, torch.Tensor], adapter_config: dict) -> Dict[str, torch.Tensor]:
    """Fold a peft LoRA adapter (the un-merged release layout: adapter_config.json + adapter_model.bin, README_EN.md:122-133)
    into a reference-named state dict, in place: what `PeftModel.from_pretrained(...).merge_and_unload()` does in
    scripts/merge_llama_with_visualcla_lora.py:78-85, without peft.

      * `<module>.lora_A.weight` [r, in] / `<module>.lora_B.weight` [out, r]  ->  W += (lora_alpha / r) * B @ A
        (transposed when fan_in_fan_out), accumulated in fp32;
      * every other adapter tensor is a `modules_to_save` module saved whole (embed_tokens / lm_head grown to the new
        vocabulary, visual_resampler.*, image_projection_layer.*): it REPLACES the base tensor.
    Adapter keys carry peft's `base_model.model.` prefix and, depending on the peft version, a `.default` adapter name or a
    `modules_to_save.default.` infix; all are accepted."""
    r = int(adapter_config["r"])
    scale = float(adapter_config["lora_alpha"]) / r
    fan_in_fan_out = bool(adapter_config.get("fan_in_fan_out", False))

    def clean(k: str) -> str:
        while k.startswith("base_model.model."):
            k = k[len("base_model.model."):]
        return k.replace(".modules_to_save.default.", ".").replace(".original_module.", ".").replace(".default.", ".")

    def resolve(k: str) -> str:
        # transformers 5.x CLIP is flat, the reference (4.x) nests vision_model.vision_model.
        if k in state_dict:
            return k
        for a, b in (("vision_model.vision_model.", "vision_model."), ("vision_model.", "vision_model.vision_model.")):
            if k.startswith(a) and b + k[len(a):] in state_dict:
                return b + k[len(a):]
        return k

    pairs: Dict[str, Dict[str, torch.Tensor]] = {}
    for k, v in adapter.items():
        k = clean(k)
        for tag in ("lora_A", "lora_B"):
            if f".{tag}." in k:
                pairs.setdefault(k.split(f".{tag}.")[0], {})[tag] = v
                break
        else:
            state_dict[resolve(k)] = v.detach().clone()
    for mod, ab in pairs.items():
        if "lora_A" not in ab or "lora_B" not in ab:
            raise KeyError(f"LoRA adapter holds only one of lora_A / lora_B for '{mod}'")
        name = resolve(mod + ".weight")
        if name not in state_dict:
            raise KeyError(f"LoRA adapter targets '{mod}', which the base checkpoint does not have")
        A, B = ab["lora_A"].float(), ab["lora_B"].float()
        if A.shape[0] != r or B.shape[1] != r:
            raise ValueError(f"LoRA rank mismatch for '{mod}': A {tuple(A.shape)}, B {tuple(B.shape)}, r={r}")
        delta = (B @ A) * scale
        if fan_in_fan_out:
            delta = delta.t()
        w = state_dict[name]
        if delta.shape != w.shape:
            raise ValueError(f"LoRA delta {tuple(delta.shape)} does not fit '{name}' {tuple(w.shape)}")
        state_dict[name] = (w.float() + delta).to(w.dtype if w.dtype != torch.float16 else torch.float32)
    return state_dict


def rope_tables(max_pos: int, head_dim: int, theta: float):
    """fp32 cos/sin [max_pos, head_dim/2], computed exactly as hf:llama/modeling_llama.py:98-127 does (CPU, fp32)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    freqs = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv_freq[None, :]
    return freqs.cos().contiguous(), freqs.sin().contiguous()


def pack_state_dict(sd: Dict[str, torch.Tensor], cfg, device, act_dtype: torch.dtype) -> Dict[str, torch.Tensor]:
    """cfg: visualcla.VisualCLAConfig (dict sub-configs).  Returns {engine tensor name: device tensor}."""
    v, r, t = cfg.vision_config, cfg.visual_resampler_config, cfg.text_config
    out: Dict[str, torch.Tensor] = {}

    vp = "vision_model.vision_model." if any(k.startswith("vision_model.vision_model.") for k in sd) else "vision_model."

    def g(name):
        if name not in sd:
            raise KeyError(f"state_dict is missing '{name}'")
        return sd[name]

    D = v["hidden_size"]
    pw = g(vp + "embeddings.patch_embedding.weight").reshape(D, -1).float()
    kpad = pad_to(pw.shape[1], 64)
    pw = torch.nn.functional.pad(pw, (0, kpad - pw.shape[1]))
    out["vit.patch_w"] = _pack_w(pw, device)
    out["vit.cls"] = _f32(g(vp + "embeddings.class_embedding").reshape(-1), device)
    out["vit.pos"] = _f32(g(vp + "embeddings.position_embedding.weight"), device)
    out["vit.pre_ln.g"] = _f32(g(vp + "pre_layrnorm.weight"), device)
    out["vit.pre_ln.b"] = _f32(g(vp + "pre_layrnorm.bias"), device)
    out["vit.post_ln.g"] = _f32(g(vp + "post_layernorm.weight"), device)
    out["vit.post_ln.b"] = _f32(g(vp + "post_layernorm.bias"), device)
    for i in range(v["num_hidden_layers"]):
        s, d = f"{vp}encoder.layers.{i}.", f"vit.l{i}."
        out[d + "ln1.g"] = _f32(g(s + "layer_norm1.weight"), device)
        out[d + "ln1.b"] = _f32(g(s + "layer_norm1.bias"), device)
        out[d + "wqkv"] = _pack_w(torch.cat([g(s + f"self_attn.{n}_proj.weight") for n in "qkv"], 0), device)
        out[d + "bqkv"] = _f32(torch.cat([g(s + f"self_attn.{n}_proj.bias") for n in "qkv"], 0), device)
        out[d + "wo"] = _pack_w(g(s + "self_attn.out_proj.weight"), device)
        out[d + "bo"] = _f32(g(s + "self_attn.out_proj.bias"), device)
        out[d + "ln2.g"] = _f32(g(s + "layer_norm2.weight"), device)
        out[d + "ln2.b"] = _f32(g(s + "layer_norm2.bias"), device)
        out[d + "w1"] = _pack_w(g(s + "mlp.fc1.weight"), device)
        out[d + "b1"] = _f32(g(s + "mlp.fc1.bias"), device)
        out[d + "w2"] = _pack_w(g(s + "mlp.fc2.weight"), device)
        out[d + "b2"] = _f32(g(s + "mlp.fc2.bias"), device)

    rp = "visual_resampler."
    # bf16-round the learned queries once, like every other weight, then hold them in the activation dtype
    out["res.query"] = g(rp + "query_embeddding").reshape(r["num_query_tokens"], r["hidden_size"]).to(
        device=device, dtype=torch.bfloat16).to(act_dtype).contiguous()
    for i in range(r["num_hidden_layers"]):
        s, d = f"{rp}encoder.layer.{i}.", f"res.l{i}."
        out[d + "wq"] = _pack_w(g(s + "crossattention.self.query.weight"), device)
        out[d + "bq"] = _f32(g(s + "crossattention.self.query.bias"), device)
        out[d + "wkv"] = _pack_w(torch.cat([g(s + "crossattention.self.key.weight"),
                                            g(s + "crossattention.self.value.weight")], 0), device)
        out[d + "bkv"] = _f32(torch.cat([g(s + "crossattention.self.key.bias"),
                                         g(s + "crossattention.self.value.bias")], 0), device)
        out[d + "wo"] = _pack_w(g(s + "crossattention.output.dense.weight"), device)
        out[d + "bo"] = _f32(g(s + "crossattention.output.dense.bias"), device)
        out[d + "ln1.g"] = _f32(g(s + "crossattention.output.LayerNorm.weight"), device)
        out[d + "ln1.b"] = _f32(g(s + "crossattention.output.LayerNorm.bias"), device)
        out[d + "w1"] = _pack_w(g(s + "intermediate.dense.weight"), device)
        out[d + "b1"] = _f32(g(s + "intermediate.dense.bias"), device)
        out[d + "w2"] = _pack_w(g(s + "output.dense.weight"), device)
        out[d + "b2"] = _f32(g(s + "output.dense.bias"), device)
        out[d + "ln2.g"] = _f32(g(s + "output.LayerNorm.weight"), device)
        out[d + "ln2.b"] = _f32(g(s + "output.LayerNorm.bias"), device)
    out["proj.w"] = _pack_w(g("image_projection_layer.weight"), device)
    out["proj.b"] = _f32(g("image_projection_layer.bias"), device)
    add_resampler_kv_all(out, r["num_hidden_layers"])

    if t.get("num_hidden_layers", 0) == 0:       # vision-only context (tgwebui pipeline): no decoder tensors
        return out
    tp = "text_model."
    out["llama.embed"] = g(tp + "model.embed_tokens.weight").to(device=device, dtype=torch.bfloat16).contiguous()
    for i in range(t["num_hidden_layers"]):
        s, d = f"{tp}model.layers.{i}.", f"llama.l{i}."
        out[d + "ln1.g"] = _f32(g(s + "input_layernorm.weight"), device)
        out[d + "ln2.g"] = _f32(g(s + "post_attention_layernorm.weight"), device)
        out[d + "wqkv"] = _pack_w(torch.cat([g(s + f"self_attn.{n}_proj.weight") for n in "qkv"], 0), device)
        out[d + "wo"] = _pack_w(g(s + "self_attn.o_proj.weight"), device)
        out[d + "wgu"] = _pack_w(interleave_gate_up(g(s + "mlp.gate_proj.weight"), g(s + "mlp.up_proj.weight")), device)
        out[d + "wd"] = _pack_w(g(s + "mlp.down_proj.weight"), device)
    out["llama.norm.g"] = _f32(g(tp + "model.norm.weight"), device)
    out["llama.lm_head"] = _pack_w(g(tp + "lm_head.weight"), device)
    hd = t["hidden_size"] // t["num_attention_heads"]
    theta = float(t.get("rope_theta") or (t.get("rope_parameters") or {}).get("rope_theta") or 10000.0)
    cos, sin = rope_tables(t["max_position_embeddings"], hd, theta)
    out["llama.rope_cos"], out["llama.rope_sin"] = cos.to(device), sin.to(device)
    if act_dtype == torch.bfloat16:
        add_fragment_copies(out)
        add_engine_stream(out, t["hidden_size"], t["num_attention_heads"], t["intermediate_size"], t["vocab_size"], t["num_hidden_layers"])
    return out


def random_packed(cfg, device, act_dtype: torch.dtype, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random-init weights generated directly in the packed device layout (benchmarks: there are no
    checkpoints to load, and a 7B state_dict does not need to round-trip through the host).
    Same distributions as the oracle's make_weights (N(0, .02) matrices, gains 1 + N(0, .1))."""
    v, r, t = cfg.vision_config, cfg.visual_resampler_config, cfg.text_config
    gen = torch.Generator(device=device).manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}

    def w(n, k, k_real=None):
        m = torch.zeros(pad_to(n, 128), k, dtype=torch.bfloat16, device=device)
        kk = k_real or k
        m[:n, :kk] = (torch.randn(n, kk, generator=gen, device=device, dtype=torch.float32) * 0.02).to(torch.bfloat16)
        return m

    def vec(n, mean=0.0, std=0.02):
        x = torch.randn(n, generator=gen, device=device, dtype=torch.float32) * std + mean
        return x.to(torch.bfloat16).float()

    D, I = v["hidden_size"], v["intermediate_size"]
    kreal = v.get("num_channels", 3) * v["patch_size"] ** 2
    npos = (v["image_size"] // v["patch_size"]) ** 2 + 1
    out["vit.patch_w"] = w(D, pad_to(kreal, 64), kreal)
    out["vit.cls"] = vec(D)
    out["vit.pos"] = vec(npos * D).view(npos, D)
    for n in ("pre_ln", "post_ln"):
        out[f"vit.{n}.g"], out[f"vit.{n}.b"] = vec(D, 1.0, 0.1), vec(D)
    for i in range(v["num_hidden_layers"]):
        d = f"vit.l{i}."
        out[d + "ln1.g"], out[d + "ln1.b"] = vec(D, 1.0, 0.1), vec(D)
        out[d + "wqkv"], out[d + "bqkv"] = w(3 * D, D), vec(3 * D)
        out[d + "wo"], out[d + "bo"] = w(D, D), vec(D)
        out[d + "ln2.g"], out[d + "ln2.b"] = vec(D, 1.0, 0.1), vec(D)
        out[d + "w1"], out[d + "b1"] = w(I, D), vec(I)
        out[d + "w2"], out[d + "b2"] = w(D, I), vec(D)
    Dr, Ir, Q = r["hidden_size"], r["intermediate_size"], r["num_query_tokens"]
    out["res.query"] = vec(Q * Dr).view(Q, Dr).to(act_dtype)
    for i in range(r["num_hidden_layers"]):
        d = f"res.l{i}."
        out[d + "wq"], out[d + "bq"] = w(Dr, Dr), vec(Dr)
        out[d + "wkv"], out[d + "bkv"] = w(2 * Dr, Dr), vec(2 * Dr)
        out[d + "wo"], out[d + "bo"] = w(Dr, Dr), vec(Dr)
        out[d + "ln1.g"], out[d + "ln1.b"] = vec(Dr, 1.0, 0.1), vec(Dr)
        out[d + "w1"], out[d + "b1"] = w(Ir, Dr), vec(Ir)
        out[d + "w2"], out[d + "b2"] = w(Dr, Ir), vec(Dr)
        out[d + "ln2.g"], out[d + "ln2.b"] = vec(Dr, 1.0, 0.1), vec(Dr)
    Dt, It, V = t["hidden_size"], t["intermediate_size"], t["vocab_size"]
    out["proj.w"], out["proj.b"] = w(Dt, Dr), vec(Dt)
    add_resampler_kv_all(out, r["num_hidden_layers"])
    out["llama.embed"] = (torch.randn(V, Dt, generator=gen, device=device, dtype=torch.float32) * 0.02).to(torch.bfloat16)
    for i in range(t["num_hidden_layers"]):
        d = f"llama.l{i}."
        out[d + "ln1.g"], out[d + "ln2.g"] = vec(Dt, 1.0, 0.1), vec(Dt, 1.0, 0.1)
        out[d + "wqkv"], out[d + "wo"] = w(3 * Dt, Dt), w(Dt, Dt)
        out[d + "wgu"], out[d + "wd"] = w(2 * It, Dt), w(Dt, It)
    out["llama.norm.g"] = vec(Dt, 1.0, 0.1)
    out["llama.lm_head"] = w(V, Dt)
    hd = Dt // t["num_attention_heads"]
    theta = float(t.get("rope_theta") or (t.get("rope_parameters") or {}).get("rope_theta") or 10000.0)
    cos, sin = rope_tables(t["max_position_embeddings"], hd, theta)
    out["llama.rope_cos"], out["llama.rope_sin"] = cos.to(device), sin.to(device)
    if act_dtype == torch.bfloat16:
        add_fragment_copies(out)
        add_engine_stream(out, Dt, t["num_attention_heads"], It, V, t["num_hidden_layers"])
    return out


def unpack_state_dict(packed: Dict[str, torch.Tensor], cfg) -> Dict[str, torch.Tensor]:
    """Inverse of pack_state_dict (CPU fp32, reference names): lets the CPU oracle run on weights that were
    generated in packed form, and backs `VisualCLAModel.state_dict()`."""
    v, r, t = cfg.vision_config, cfg.visual_resampler_config, cfg.text_config
    sd: Dict[str, torch.Tensor] = {}
    c = lambda x: x.detach().float().cpu()
    D = v["hidden_size"]
    C_, P = v.get("num_channels", 3), v["patch_size"]
    vp = "vision_model.vision_model."
    sd[vp + "embeddings.patch_embedding.weight"] = c(packed["vit.patch_w"][:D, :C_ * P * P]).reshape(D, C_, P, P)
    sd[vp + "embeddings.class_embedding"] = c(packed["vit.cls"])
    sd[vp + "embeddings.position_embedding.weight"] = c(packed["vit.pos"])
    for a, b in (("pre_layrnorm", "pre_ln"), ("post_layernorm", "post_ln")):
        sd[vp + a + ".weight"], sd[vp + a + ".bias"] = c(packed[f"vit.{b}.g"]), c(packed[f"vit.{b}.b"])
    I = v["intermediate_size"]
    for i in range(v["num_hidden_layers"]):
        s, d = f"{vp}encoder.layers.{i}.", f"vit.l{i}."
        wqkv, bqkv = c(packed[d + "wqkv"][:3 * D]), c(packed[d + "bqkv"])
        for j, n in enumerate("qkv"):
            sd[s + f"self_attn.{n}_proj.weight"] = wqkv[j * D:(j + 1) * D]
            sd[s + f"self_attn.{n}_proj.bias"] = bqkv[j * D:(j + 1) * D]
        sd[s + "self_attn.out_proj.weight"], sd[s + "self_attn.out_proj.bias"] = c(packed[d + "wo"][:D]), c(packed[d + "bo"])
        sd[s + "layer_norm1.weight"], sd[s + "layer_norm1.bias"] = c(packed[d + "ln1.g"]), c(packed[d + "ln1.b"])
        sd[s + "layer_norm2.weight"], sd[s + "layer_norm2.bias"] = c(packed[d + "ln2.g"]), c(packed[d + "ln2.b"])
        sd[s + "mlp.fc1.weight"], sd[s + "mlp.fc1.bias"] = c(packed[d + "w1"][:I]), c(packed[d + "b1"])
        sd[s + "mlp.fc2.weight"], sd[s + "mlp.fc2.bias"] = c(packed[d + "w2"][:D]), c(packed[d + "b2"])
    rp = "visual_resampler."
    Dr, Ir = r["hidden_size"], r["intermediate_size"]
    sd[rp + "query_embeddding"] = c(packed["res.query"]).reshape(1, r["num_query_tokens"], Dr)
    for i in range(r["num_hidden_layers"]):
        s, d = f"{rp}encoder.layer.{i}.", f"res.l{i}."
        sd[s + "crossattention.self.query.weight"], sd[s + "crossattention.self.query.bias"] = c(packed[d + "wq"][:Dr]), c(packed[d + "bq"])
        wkv, bkv = c(packed[d + "wkv"][:2 * Dr]), c(packed[d + "bkv"])
        sd[s + "crossattention.self.key.weight"], sd[s + "crossattention.self.value.weight"] = wkv[:Dr], wkv[Dr:]
        sd[s + "crossattention.self.key.bias"], sd[s + "crossattention.self.value.bias"] = bkv[:Dr], bkv[Dr:]
        sd[s + "crossattention.output.dense.weight"], sd[s + "crossattention.output.dense.bias"] = c(packed[d + "wo"][:Dr]), c(packed[d + "bo"])
        sd[s + "crossattention.output.LayerNorm.weight"], sd[s + "crossattention.output.LayerNorm.bias"] = c(packed[d + "ln1.g"]), c(packed[d + "ln1.b"])
        sd[s + "intermediate.dense.weight"], sd[s + "intermediate.dense.bias"] = c(packed[d + "w1"][:Ir]), c(packed[d + "b1"])
        sd[s + "output.dense.weight"], sd[s + "output.dense.bias"] = c(packed[d + "w2"][:Dr]), c(packed[d + "b2"])
        sd[s + "output.LayerNorm.weight"], sd[s + "output.LayerNorm.bias"] = c(packed[d + "ln2.g"]), c(packed[d + "ln2.b"])
    Dt, It, V = t["hidden_size"], t["intermediate_size"], t["vocab_size"]
    sd["image_projection_layer.weight"], sd["image_projection_layer.bias"] = c(packed["proj.w"][:Dt]), c(packed["proj.b"])
    if "llama.embed" not in packed:              # vision-only context
        return sd
    tp = "text_model."
    sd[tp + "model.embed_tokens.weight"] = c(packed["llama.embed"])
    for i in range(t["num_hidden_layers"]):
        s, d = f"{tp}model.layers.{i}.", f"llama.l{i}."
        wqkv = c(packed[d + "wqkv"][:3 * Dt])
        for j, n in enumerate("qkv"):
            sd[s + f"self_attn.{n}_proj.weight"] = wqkv[j * Dt:(j + 1) * Dt]
        sd[s + "self_attn.o_proj.weight"] = c(packed[d + "wo"][:Dt])
        gu = c(packed[d + "wgu"][:2 * It]).reshape(It // 16, 2, 16, Dt)
        sd[s + "mlp.gate_proj.weight"], sd[s + "mlp.up_proj.weight"] = gu[:, 0].reshape(It, Dt), gu[:, 1].reshape(It, Dt)
        sd[s + "mlp.down_proj.weight"] = c(packed[d + "wd"][:Dt])
        sd[s + "input_layernorm.weight"], sd[s + "post_attention_layernorm.weight"] = c(packed[d + "ln1.g"]), c(packed[d + "ln2.g"])
    sd[tp + "model.norm.weight"] = c(packed["llama.norm.g"])
    sd[tp + "lm_head.weight"] = c(packed["llama.lm_head"][:V])
    return sd
