"""CPU baseline of BASELINE configs[0]: "the reference CPU HuggingFace path" timed on this host.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/visualcla_oracle.py header): executed as a subprocess by the `cpu_baseline` leg of
bench.py and by nothing else.  It runs what the reference's `VisualCLAModel.generate` runs on a CPU (models/visualcla/modeling_visualcla.py:
334-392, scripts/inference/inference.py:78-79 `.float()` on CPU): transformers' own `CLIPVisionModel` (+ post_layernorm over all tokens,
:349-350) and `LlamaForCausalLM.generate(inputs_embeds=...)` (:382-391), fp32, eager attention, with this repo's restatement of the
reference's resampler + projection + splice between them (oracle.visualcla_oracle: the reference's own modeling_visual_resampler.py does
not import under transformers 5.x).  kind = "hf+port-resampler".

Weights are random values of the 7B architecture (timing only: a 4M-element N(0, 0.02) block tiled through every matrix, so that filling
27 GB takes seconds instead of a minute of serial RNG; RMSNorm / LayerNorm gains 1).  NUMA: the process pins itself to the cores of ONE
memory node before torch starts, so every page is first-touched and read locally; the decode step is then swept over thread counts and
the best one is used for the timed request.  Prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time


def node0_cores(limit: int):
    """logical CPUs of memory node 0, one per physical core first (SMT siblings last)"""
    def parse(lst):
        out = []
        for part in lst.strip().split(","):
            if "-" in part:
                a, b = part.split("-")
                out += list(range(int(a), int(b) + 1))
            elif part:
                out.append(int(part))
        return out
    try:
        cpus = parse(open("/sys/devices/system/node/node0/cpulist").read())
    except OSError:
        cpus = sorted(os.sched_getaffinity(0))
    allowed = os.sched_getaffinity(0)
    cpus = [c for c in cpus if c in allowed] or sorted(allowed)
    primary, seen = [], set()
    for c in cpus:
        try:
            sib = tuple(parse(open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read()))
        except OSError:
            sib = (c,)
        if sib[0] not in seen:
            seen.add(sib[0])
            primary.append(c)
    rest = [c for c in cpus if c not in primary]
    return (primary + rest)[:limit], len(primary)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompt-len", type=int, default=128)
    ap.add_argument("--tokens", type=int, default=32)
    ap.add_argument("--sweep", default="8,16,32,64")
    ap.add_argument("--geometry", default="7b", choices=["7b", "small"])
    args = ap.parse_args()
    sweep = [int(x) for x in args.sweep.split(",")]
    cores, n_phys = node0_cores(max(sweep))
    os.sched_setaffinity(0, cores)
    os.environ.setdefault("OMP_NUM_THREADS", str(len(cores)))
    os.environ.setdefault("OMP_PROC_BIND", "close")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

    import torch
    from transformers import CLIPVisionConfig, CLIPVisionModel, GenerationConfig, LlamaConfig, LlamaForCausalLM
    from oracle import visualcla_oracle as O

    torch.set_num_threads(len(cores))
    cfg = O.cfg_7b() if args.geometry == "7b" else O.cfg_small()
    v, r, t = cfg.vision, cfg.resampler, cfg.text
    t0 = time.time()
    with torch.device("meta"):
        llama = LlamaForCausalLM(LlamaConfig(
            vocab_size=t.vocab_size, hidden_size=t.hidden_size, intermediate_size=t.intermediate_size, num_hidden_layers=t.num_hidden_layers,
            num_attention_heads=t.num_attention_heads, num_key_value_heads=t.num_attention_heads, rms_norm_eps=t.rms_norm_eps,
            max_position_embeddings=t.max_position_embeddings, rope_theta=t.rope_theta, tie_word_embeddings=False,
            bos_token_id=1, eos_token_id=2, attn_implementation="eager"))
        clip = CLIPVisionModel(CLIPVisionConfig(
            hidden_size=v.hidden_size, intermediate_size=v.intermediate_size, num_hidden_layers=v.num_hidden_layers,
            num_attention_heads=v.num_attention_heads, image_size=v.image_size, patch_size=v.patch_size, hidden_act=v.hidden_act,
            layer_norm_eps=v.layer_norm_eps, attn_implementation="eager"))
    g = torch.Generator().manual_seed(0)
    blk = torch.randn(1 << 22, generator=g) * 0.02

    def fill(module):
        module.to_empty(device="cpu")
        for name, p in module.named_parameters():
            flat = p.data.view(-1)
            if p.dim() == 1 and ("norm" in name.lower() or "layrnorm" in name) and name.endswith("weight"):
                flat.fill_(1.0)
                continue
            n, m = flat.numel(), blk.numel()
            reps = n // m
            if reps:
                flat[: reps * m].view(reps, m).copy_(blk)                 # parallel, bandwidth-bound; first touch on this node
            flat[reps * m:].copy_(blk[: n - reps * m])
        for name, b in module.named_buffers():                            # rotary inv_freq / position ids: recompute
            if "inv_freq" in name:
                d = b.numel() * 2
                b.data.copy_(1.0 / (t.rope_theta ** (torch.arange(0, d, 2, dtype=torch.float32) / d)))
            elif "position_ids" in name:
                b.data.copy_(torch.arange(b.shape[-1]).expand_as(b))
        return module.eval().float()
    llama, clip = fill(llama), fill(clip)
    W = {k: val for k, val in O.make_weights(O.OracleCfg(vision=O.VisionCfg(num_hidden_layers=0, hidden_size=v.hidden_size, image_size=v.image_size),
                                                        resampler=r, text=O.TextCfg(num_hidden_layers=0, hidden_size=t.hidden_size, vocab_size=8,
                                                                                    intermediate_size=16)), seed=0).items()
         if k.startswith(("visual_resampler.", "image_projection_layer."))}
    t_init = time.time() - t0

    px, ids, mask = O.make_inputs(cfg, 1, args.prompt_len)
    emb_w = llama.get_input_embeddings().weight
    vis = clip.vision_model if hasattr(clip, "vision_model") else clip       # transformers 4.x nests the tower, 5.x is flat
    gen = GenerationConfig(max_new_tokens=args.tokens, min_new_tokens=args.tokens, do_sample=False, num_beams=1, bos_token_id=1,
                           eos_token_id=None, pad_token_id=0)

    def vision(px_):
        with torch.no_grad():
            h = vis.post_layernorm(clip(pixel_values=px_).last_hidden_state)            # modeling_visualcla.py:349-350: ALL tokens
            lat = O.resampler_forward(h, W, r)                                          # :351-353 (restated, see the header)
            return O.image_projection(lat, W)                                           # :354

    def request(n_tokens):
        with torch.no_grad():
            tb = time.time()
            img = vision(px)
            tv = time.time() - tb
            x = emb_w[ids].clone()
            p0 = int((ids[0] == cfg.img_start_token_id).nonzero()[0])
            Q = img.shape[1]
            x[0, p0 + 1: p0 + 1 + Q] = img[0]                                           # :358-370
            gen.max_new_tokens = gen.min_new_tokens = n_tokens
            t1 = time.time()
            out = llama.generate(inputs_embeds=x, attention_mask=mask, generation_config=gen)    # :382-391
            return out, tv, time.time() - t1, time.time() - tb

    # thread sweep on a short request (prefill + 3 decode steps), best decode rate wins
    sweep_res = {}
    for n in [s_ for s_ in sweep if s_ <= len(cores)]:
        torch.set_num_threads(n)
        with torch.no_grad():
            x1 = emb_w[ids[:, :1]].clone()
            past = llama(inputs_embeds=emb_w[ids].clone(), use_cache=True).past_key_values
            t1 = time.time()
            for _ in range(3):
                o = llama(inputs_embeds=x1, past_key_values=past, use_cache=True)
                past = o.past_key_values
            sweep_res[n] = (time.time() - t1) / 3
    best = min(sweep_res, key=sweep_res.get)
    torch.set_num_threads(best)
    out, tv, tgen, total = request(args.tokens)
    n_out = int(out.shape[1])
    # split prefill / decode: one more prefill-only call
    with torch.no_grad():
        t1 = time.time()
        llama(inputs_embeds=emb_w[ids].clone(), use_cache=True)
        t_pre = time.time() - t1
    t_dec = max(tgen - t_pre, 1e-9)
    cpu_model = ""
    try:
        cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        pass
    print(json.dumps({
        "value": round(n_out / total, 3), "unit": "tokens/s", "cores": best, "kind": "hf+port-resampler",
        "images_per_sec": round(1.0 / total, 4),
        "decode_s_per_token": round(t_dec / max(n_out - 1, 1), 4),
        "thread_sweep_decode_s_per_token": {str(k): round(val, 4) for k, val in sweep_res.items()},
        "sample": (f"BASELINE configs[0] in full: 1 image, T={args.prompt_len} prompt, {n_out} greedy tokens in {total:.1f}s (vision stack {tv:.2f}s, "
                   f"prefill ~{t_pre:.2f}s, {n_out - 1} decode steps ~{t_dec:.1f}s = {t_dec / max(n_out - 1, 1):.3f}s/token) on transformers' CLIPVisionModel + "
                   f"LlamaForCausalLM.generate (fp32, eager attention) with the restated resampler between them; {best} threads pinned to memory node 0 "
                   f"({n_phys} physical cores on that node, {os.cpu_count()} logical CPUs on the host; {cpu_model}); random tiled weights filled in {t_init:.1f}s (not timed)"),
    }))


if __name__ == "__main__":
    main()
