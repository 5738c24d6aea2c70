#!/usr/bin/env python3
"""Follow-up: ONE model, the engine stream (13.4 GB) copied into several allocations; the context re-pointed at each copy and the step timed.
Allocation order / size tricks tried: plain torch allocations, and sub-ranges of one large allocation at different offsets."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "visual-chinese-llama-alpaca_amd"))
import visualcla
from visualcla import _lib

lib = _lib.load()
cfg = visualcla.visualcla_7b_config()
cfg.vision_config.update(num_hidden_layers=1, hidden_size=256, intermediate_size=512, num_attention_heads=4)
cfg.visual_resampler_config.update(num_hidden_layers=1, hidden_size=256, intermediate_size=512, num_attention_heads=4)
m = visualcla.VisualCLAModel.from_random(cfg, device="cuda:0", torch_dtype=torch.bfloat16, seed=3)
T, N, ctx_max = 128, 64, 256
V = cfg.text_config["vocab_size"]
ids = torch.randint(3, V - 8, (1, T), generator=torch.Generator().manual_seed(5)).to("cuda:0")
out = torch.zeros(N + 1, 1, dtype=torch.int64, device="cuda:0"); out[0] = 17
orig = m._packed["llama.engine.w"]
nbytes = orig.numel() * 2

def time_with(stream_tensor):
    m._packed["llama.engine.w"] = stream_tensor
    m._build_ctx()
    embeds, _ = m._embed(ids, None, None)
    cache = m._new_cache(1, ctx_max)
    m._prefill(embeds, cache, None, all_logits=False)
    ws = m._buf("llama", lib.vcla_llama_workspace_bytes(m._ctx, 1, 1))
    for rep in range(3):
        m._pos_dev.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(lib.vcla_llama_decode_loop(m._ctx, out[0].data_ptr(), 1, T, m._pos_dev.data_ptr(), N, cache.kv.data_ptr(), ctx_max, None, out[1:].data_ptr(),
                                              ws.data_ptr(), ws.numel(), 0, _lib.stream_ptr()))
        _lib.check(lib.vcla_llama_decode_status(m._ctx, 1, ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
        dt = time.perf_counter() - t0
    return dt / N * 1e3

print(f"original at {orig.data_ptr():#x}: {time_with(orig):.4f} ms/step", flush=True)
copies = []
for i in range(4):
    c = orig.clone()
    copies.append(c)
    print(f"clone {i} at {c.data_ptr():#x}: {time_with(c):.4f} ms/step", flush=True)
big = torch.empty(nbytes + (64 << 20), dtype=torch.uint8, device="cuda:0")
for off in (0, 4096, 1 << 20, 2 << 20, 17 << 20, 33 << 20 | 8192):
    v = big[off:off + nbytes].view(torch.bfloat16).view(orig.shape)
    v.copy_(orig)
    print(f"big+{off:#x} at {v.data_ptr():#x}: {time_with(v):.4f} ms/step", flush=True)
print(f"original again: {time_with(orig):.4f} ms/step; clone 0 again: {time_with(copies[0]):.4f}", flush=True)
