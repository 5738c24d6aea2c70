#!/usr/bin/env python
"""Workload for the PMC / kernel-trace passes of the persistent B = 1 decode step (decode_engine_kernel): the full 7B model, a 128-token prompt, 12 eager
decode steps (each ONE engine launch streaming 13.36 GB of weights + the K/V rows of the context)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-chinese-llama-alpaca_amd"))
import torch
import visualcla
from visualcla import _lib

cfg = visualcla.visualcla_7b_config()
cfg.vision_config.update(num_hidden_layers=1)          # the vision tower is not run here
cfg.visual_resampler_config.update(num_hidden_layers=1)
m = visualcla.VisualCLAModel.from_random(cfg, device="cuda:0", torch_dtype=torch.bfloat16, seed=0)
lib = _lib.load()
T, n = 128, 12
V = cfg.text_config["vocab_size"]
ids = torch.randint(3, V - 8, (1, T), generator=torch.Generator().manual_seed(5)).cuda()
ctx_max = 192
embeds, _ = m._embed(ids, None, None)
cache = m._new_cache(1, ctx_max)
logits = m._prefill(embeds, cache, None, all_logits=False)
ws = m._buf("llama", lib.vcla_llama_workspace_bytes(m._ctx, 1, 1))
lg = torch.empty(1, V, dtype=torch.float32, device="cuda:0")
tok = logits.argmax(-1)
for s in range(n):
    _lib.check(lib.vcla_llama_decode_step(m._ctx, tok.contiguous().data_ptr(), 1, T + s, None, 0, cache.kv.data_ptr(), ctx_max, None, lg.data_ptr(), None,
                                          ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
    tok = lg.argmax(-1)
_lib.check(lib.vcla_llama_decode_status(m._ctx, 1, ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
print("done", int(tok))
