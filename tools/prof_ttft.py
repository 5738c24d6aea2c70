#!/usr/bin/env python
"""Time to first token at B = 1 (one 224 px image, T = 128 prompt with 64 image tokens: BASELINE configs[1]'s request): vision stack + prefill, no decode
steps.  The workload behind `rocprofv3 --kernel-trace -- python tools/prof_ttft.py [reps]` (tools/prof_by_grid.py prints the per-kernel averages)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-chinese-llama-alpaca_amd"))
import torch
import visualcla
from visualcla.synthetic import make_inputs, stub_tokenizer

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = visualcla.visualcla_7b_config()
m = visualcla.VisualCLAModel.from_random(cfg, device="cuda:0", torch_dtype=torch.bfloat16, seed=0)
m.tokenizer = stub_tokenizer()
m.image_at_head = False
px, ids, mask = make_inputs(m.config, B, 128)
px, ids, mask = px.to(m.device, torch.bfloat16), ids.to(m.device), mask.to(m.device)
kw = dict(input_ids=ids, pixel_values=px, attention_mask=mask, max_new_tokens=1, do_sample=False, eos_token_id=None)
for name, fn in (("vision stack", lambda: m.embed_images(px)), ("generate(max_new_tokens=1)", lambda: m.generate(**kw))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(f"B={B} {name}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms")
