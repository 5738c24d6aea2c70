#!/usr/bin/env python
"""Run the vision stack alone (ViT + resampler + projection, B images) a few times: the workload behind
`rocprofv3 --kernel-trace -- python tools/prof_vision.py [B] [reps]` (tools/prof_stats.py prints the per-kernel totals)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-chinese-llama-alpaca_amd"))
import torch
import visualcla

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = visualcla.visualcla_7b_config()
cfg.text_config = dict(cfg.text_config, num_hidden_layers=1)     # the decoder is not exercised here: keep its weights small
m = visualcla.VisualCLAModel.from_random(cfg, device="cuda:0", torch_dtype=torch.bfloat16, seed=0)
px = torch.randn(B, 3, 224, 224, device="cuda:0").to(torch.bfloat16)
m.embed_images(px)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    m.embed_images(px)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"vision stack B={B}: {dt*1e3:.2f} ms  {B/dt:.0f} img/s  {B*179.2e9/dt/1e12:.0f} TF/s ({B*179.2e9/dt/2.5e15*100:.1f}% of bf16 MFMA peak)")
