#!/usr/bin/env python3
"""Where does the process-to-process spread of the persistent decode step come from (2.29 - 2.40 ms on one box)?  Several models in ONE process, each with its own
weight / workspace / cache allocations, timed alternately: a spread BETWEEN models that is stable across rounds = placement of the buffers, not the process."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "visual-chinese-llama-alpaca_amd"))
import visualcla
from visualcla import _lib

n_models = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lib = _lib.load()
cfg = visualcla.visualcla_7b_config()
cfg.vision_config.update(num_hidden_layers=1, hidden_size=256, intermediate_size=512, num_attention_heads=4)
cfg.visual_resampler_config.update(num_hidden_layers=1, hidden_size=256, intermediate_size=512, num_attention_heads=4)
models = []
for i in range(n_models):
    m = visualcla.VisualCLAModel.from_random(cfg, device="cuda:0", torch_dtype=torch.bfloat16, seed=3)
    models.append(m)
    print(f"model {i}: engine.w at {m._packed['llama.engine.w'].data_ptr():#x}", flush=True)
T, N = 128, 64
V = cfg.text_config["vocab_size"]
ids = torch.randint(3, V - 8, (1, T), generator=torch.Generator().manual_seed(5)).to("cuda:0")
ctx_max = 256
state = []
for m in models:
    embeds, _ = m._embed(ids, None, None)
    cache = m._new_cache(1, ctx_max, _persistent=True)
    m._prefill(embeds, cache, None, all_logits=False, _persistent=True)
    ws = m._buf("llama", lib.vcla_llama_workspace_bytes(m._ctx, 1, 1))
    out = m._typed_buf("gen_out", (N + 1, 1), torch.int64)
    out[0] = 17
    state.append((cache, ws, out))
for rnd in range(4):
    line = f"round {rnd}: "
    for i, m in enumerate(models):
        cache, ws, out = state[i]
        for rep in range(2):
            m._pos_dev.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _lib.check(lib.vcla_llama_decode_loop(m._ctx, out[0].data_ptr(), 1, T, m._pos_dev.data_ptr(), N, cache.kv.data_ptr(), ctx_max, None, out[1:].data_ptr(),
                                                  ws.data_ptr(), ws.numel(), 1, _lib.stream_ptr()))
            _lib.check(lib.vcla_llama_decode_status(m._ctx, 1, ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
            dt = time.perf_counter() - t0
        line += f"model {i} {dt / N * 1e3:.4f} ms/step | "
    print(line, flush=True)
