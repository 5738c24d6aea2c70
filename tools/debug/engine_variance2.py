#!/usr/bin/env python3
"""Follow-up to engine_variance.py: ONE model; the workspace (mailboxes), the K/V cache and the token buffers moved to other addresses one at a time."""
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
embeds, _ = m._embed(ids, None, None)
nws = lib.vcla_llama_workspace_bytes(m._ctx, 1, 1)
big = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda:0")
out = torch.zeros(N + 1, 1, dtype=torch.int64, device="cuda:0"); out[0] = 17

def run(ws, cache):
    best = None
    for rep in range(3):
        m._pos_dev.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(lib.vcla_llama_decode_loop(m._ctx, out[0].data_ptr(), 1, T, m._pos_dev.data_ptr(), N, cache.kv.data_ptr(), ctx_max, None, out[1:].data_ptr(),
                                              ws.data_ptr(), ws.numel(), 0, _lib.stream_ptr()))
        _lib.check(lib.vcla_llama_decode_status(m._ctx, 1, ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
        dt = time.perf_counter() - t0
    return dt / N * 1e3

caches = []
for i in range(3):
    c = m._new_cache(1, ctx_max)
    m._prefill(embeds, c, None, all_logits=False)
    caches.append(c)
print("workspace offsets inside one 1-GiB allocation (cache 0):")
for off in (0, 4096, 65536, 1 << 20, 2 << 20, 33 << 20, 64 << 20, 100 << 20 | 12288, 512 << 20):
    ws = big[off:off + nws]
    print(f"  ws at +{off:#x} ({ws.data_ptr():#x}): {run(ws, caches[0]):.4f} ms/step", flush=True)
print("caches (workspace +0):")
for i, c in enumerate(caches):
    print(f"  cache {i} at {c.kv.data_ptr():#x}: {run(big[:nws], c):.4f} ms/step", flush=True)
print("again, workspace offsets:")
for off in (0, 4096, 1 << 20, 33 << 20):
    ws = big[off:off + nws]
    print(f"  ws at +{off:#x}: {run(ws, caches[0]):.4f} ms/step", flush=True)
