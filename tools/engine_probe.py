#!/usr/bin/env python3
"""A/B of the persistent B = 1 decode step (csrc/decode_engine.hip, VCLA_ENGINE=1) against the launch path (VCLA_ENGINE=0) on ONE box:
the same model, cache and tokens through both; logits compared step by step, then both forms timed in hipGraph-replayed decode loops.

    python tools/engine_probe.py [--layers 4] [--steps 6] [--ctx 160] [--time 64]

Runs a reduced-depth decoder at the LLaMA-7B widths by default (the engine's geometry is fixed; depth and vocabulary are free), the full
7B with --layers 32.  Every wait inside the engine is bounded, so a protocol bug shows as a status code, not a hang; run it under
`timeout` anyway."""
import argparse
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "visual-chinese-llama-alpaca_amd"))
import visualcla  # noqa: E402
from visualcla import _lib  # noqa: E402


def build(layers: int, vocab: int, inter: int):
    cfg = visualcla.visualcla_7b_config()
    cfg.text_config.update(num_hidden_layers=layers, vocab_size=vocab, intermediate_size=inter)
    # a small vision tower: the probe never runs it
    cfg.vision_config.update(num_hidden_layers=1, hidden_size=256, intermediate_size=512, num_attention_heads=4)
    cfg.visual_resampler_config.update(num_hidden_layers=1, hidden_size=256, intermediate_size=512, num_attention_heads=4)
    return visualcla.VisualCLAModel.from_random(cfg, device="cuda:0", torch_dtype=torch.bfloat16, seed=3)


PHASES = ["stage x (gather + RMSNorm)", "qkv slots", "attention (attention CUs only)", "o_proj (slices from the mailbox) + finish", "gather x1 + RMSNorm",
          "gate/up slots", "down (slices from the mailbox) + finish"]
STAMPS = [0, 1, 2, 3, 5, 6, 7, 9]


def timeline(m, lib, embeds, T, ctx_max, a):
    """One engine step with debug stamps (VCLA_ENGINE_TL): leader-wave wall-clock (100 MHz) at the phase boundaries of every layer and CU."""
    dev = m._device
    L = a.layers
    os.environ["VCLA_ENGINE"] = "1"
    tl = torch.zeros(256, 2048, dtype=torch.int64, device=dev)
    os.environ["VCLA_ENGINE_TL"] = hex(tl.data_ptr())
    cache = m._new_cache(1, ctx_max)
    m._prefill(embeds, cache, None, all_logits=False)
    ws = m._buf("llama", lib.vcla_llama_workspace_bytes(m._ctx, 1, 1))
    tok = torch.tensor([17], device=dev)
    lg = torch.empty(1, m.config.text_config["vocab_size"], dtype=torch.float32, device=dev)
    for s in range(3):      # the last step's stamps are the ones read
        _lib.check(lib.vcla_llama_decode_step(m._ctx, tok.data_ptr(), 1, T + s, None, 0, cache.kv.data_ptr(), ctx_max, None, lg.data_ptr(), None,
                                              ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
        _lib.check(lib.vcla_llama_decode_status(m._ctx, 1, ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
    del os.environ["VCLA_ENGINE_TL"]
    t = tl.cpu().double() * 0.01      # us
    st = t[:, :L * 16].view(256, L, 16)[:, :, STAMPS]
    cu = torch.arange(256)
    attn = (cu & 7) == ((cu >> 3) & 7)
    d = st[:, :, 1:] - st[:, :, :-1]           # [256, L, 9] phase durations
    lay = slice(1, L) if L > 1 else slice(0, 1)
    print(f"engine timeline, one step, layers {lay.start}..{L - 1}, us (median over CUs | max over CUs; attention CUs / other CUs):")
    for k, name in enumerate(PHASES):
        da, do = d[attn][:, lay, k].flatten(), d[~attn][:, lay, k].flatten()
        print(f"  {name:34s} attn-CU {da.median():6.2f} | {da.max():6.2f}    other {do.median():6.2f} | {do.max():6.2f}")
    full = t[:, :L * 16].view(256, L, 16)
    sa = full[attn][:, lay]
    print(f"  inside the attention (attention CUs, leader wave): qkv done -> q/k/v granules complete {(sa[..., 10] - sa[..., 2]).median():.2f} | RoPE, cache append, LDS "
          f"{(sa[..., 11] - sa[..., 10]).median():.2f} | scores + PV over the cache {(sa[..., 12] - sa[..., 11]).median():.2f} (the key batches {(sa[..., 13] - sa[..., 11]).median():.2f}, "
          f"the new token + the wave's 4 lane groups {(sa[..., 14] - sa[..., 13]).median():.2f}, share to LDS {(sa[..., 12] - sa[..., 14]).median():.2f}) | wait for the other waves, merge, publish "
          f"{(sa[..., 3] - sa[..., 12]).median():.2f}")
    per_layer = (st[:, lay, -1] - st[:, lay, 0])
    print(f"  layer, stamp 0 -> 9: median {per_layer.median():.2f} us, max {per_layer.max():.2f} us; whole step (loader begin -> end): "
          f"{(t[:, 2041] - t[:, 2040]).median():.1f} us")
    stall, ns = t[:, 2042], tl.cpu()[:, 2043].double()
    print(f"  loader: ring-full stalls {ns.median():.0f} per CU, {stall.median():.1f} us stalled per CU (median) of the step; "
          f"slots {a.layers * (32 + a.inter // 256 + (a.inter // 256 + 1) // 2)} + lm_head")
    SL = 32 + a.inter // 256 + (a.inter // 256 + 1) // 2
    by_slot = t[:, 1024:1024 + SL].median(dim=0).values / L          # us per layer, median over CUs
    names = [(0, 24, "qkv"), (24, 32, "o_proj"), (32, 32 + a.inter // 256, "gate/up"), (32 + a.inter // 256, SL, "down")]
    print("  loader stall by the slot it could not issue (us per layer, median over CUs): " +
          "; ".join(f"{n} {by_slot[lo:hi].sum():.2f} (worst slot {lo + int(by_slot[lo:hi].argmax())}: {by_slot[lo:hi].max():.2f})" for lo, hi, n in names))
    skew = st[:, lay, 1] - st[:, lay, 1].min(dim=0, keepdim=True).values
    print(f"  skew of 'x staged' over CUs: median {skew.median():.2f} us, max {skew.max():.2f} us")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--vocab", type=int, default=5003)
    ap.add_argument("--inter", type=int, default=11008)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--ctx", type=int, default=160, help="prompt length")
    ap.add_argument("--time", type=int, default=64, help="decode steps per timed loop (0: skip)")
    ap.add_argument("--mask", action="store_true", help="pass an all-ones key mask (the MASK instantiation)")
    ap.add_argument("--timeline", action="store_true", help="print the per-phase wall-clock breakdown of one engine step (debug stamps)")
    a = ap.parse_args()
    lib = _lib.load()
    m = build(a.layers, a.vocab, a.inter)
    assert "llama.engine.w" in m._packed, "engine stream was not built"
    dev = m._device
    t = m.config.text_config
    V, T = t["vocab_size"], a.ctx
    g = torch.Generator(device="cpu").manual_seed(5)
    ids = torch.randint(3, V - 8, (1, T), generator=g).to(dev)
    n_new = max(a.steps, a.time) + 2
    ctx_max = (T + n_new + 63) // 64 * 64
    embeds, _ = m._embed(ids, None, None)
    results = {}
    for mode in ("0", "1"):
        os.environ["VCLA_ENGINE"] = mode
        cache = m._new_cache(1, ctx_max)
        am = torch.ones(1, T, dtype=torch.int64, device=dev) if a.mask else None
        key_mask = m._key_mask(am, 1, T, ctx_max)
        logits = m._prefill(embeds, cache, key_mask, all_logits=False)
        ws = m._buf("llama", lib.vcla_llama_workspace_bytes(m._ctx, 1, 1))
        step_logits = torch.empty(1, V, dtype=torch.float32, device=dev)
        tok = logits.argmax(-1)
        toks, lgs = [int(tok)], []
        forced = results.get("0", (None, None))[0]
        for s in range(a.steps):
            if forced is not None:                       # teacher-forced on the launch path's tokens: the logits stay comparable
                tok = torch.tensor([forced[s]], device=dev)
            _lib.check(lib.vcla_llama_decode_step(m._ctx, tok.contiguous().data_ptr(), 1, T + s, None, 0, cache.kv.data_ptr(), ctx_max,
                                                  _lib.ptr(key_mask), step_logits.data_ptr(), None, ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
            _lib.check(lib.vcla_llama_decode_status(m._ctx, 1, ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
            lgs.append(step_logits.clone())
            tok = step_logits.argmax(-1)
            toks.append(int(tok))
        torch.cuda.synchronize()
        results[mode] = (toks, lgs, cache.kv.clone())
        print(f"VCLA_ENGINE={mode}: tokens {toks}", flush=True)
    l0, l1 = results["0"][1], results["1"][1]
    for s in range(a.steps):
        d = (l0[s] - l1[s]).abs()
        print(f"step {s}: logits std {l0[s].std().item():.3f}  |engine - launches| max {d.max().item():.4f} mean {d.mean().item():.5f}  "
              f"argmax {int(l0[s].argmax())} / {int(l1[s].argmax())}  finite {bool(torch.isfinite(l1[s]).all())}", flush=True)
    kd = (results["0"][2][..., :T + a.steps, :].float() - results["1"][2][..., :T + a.steps, :].float()).abs()
    print(f"K/V cache: max |diff| {kd.max().item():.4f} over the {T + a.steps} rows both paths wrote", flush=True)
    if a.timeline:
        timeline(m, lib, embeds, T, ctx_max, a)
    if a.time > 0:
        for mode in ("0", "1", "0", "1"):
            os.environ["VCLA_ENGINE"] = mode
            cache = m._new_cache(1, ctx_max, _persistent=True)
            key_mask = m._key_mask(None, 1, T, ctx_max)
            m._prefill(embeds, cache, key_mask, all_logits=False, _persistent=True)
            ws = m._buf("llama", lib.vcla_llama_workspace_bytes(m._ctx, 1, 1))
            out = m._typed_buf("gen_out", (a.time + 1, 1), torch.int64)
            out[0] = 17
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                for rep in range(3):
                    m._pos_dev.zero_()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    _lib.check(lib.vcla_llama_decode_loop(m._ctx, out[0].data_ptr(), 1, T, m._pos_dev.data_ptr(), a.time, cache.kv.data_ptr(), ctx_max,
                                                          _lib.ptr(key_mask), out[1:].data_ptr(), ws.data_ptr(), ws.numel(), 1, _lib.stream_ptr()))
                    _lib.check(lib.vcla_llama_decode_status(m._ctx, 1, ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
                    dt = time.perf_counter() - t0
            per_layer = dt / a.time * 1e6 / a.layers
            print(f"VCLA_ENGINE={mode}: {dt / a.time * 1e3:.4f} ms/step over {a.time} graph-replayed steps  (~{per_layer:.1f} us per layer incl. lm_head share)  "
                  f"tokens {out[1:6, 0].tolist()}", flush=True)


if __name__ == "__main__":
    main()
