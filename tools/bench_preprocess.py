#!/usr/bin/env python
"""Next row N1 throughput: CLIP preprocessing of decoded uint8 RGB frames (640 x 480 -> 224 x 224, bicubic, bit-exact with
Pillow / CLIPImageProcessor) on the GPU -- batched entry point fed from ONE pinned staging buffer -- next to the per-image entry
and to transformers' CLIPImageProcessor on the host.  The vision stack consumes ~3 k img/s at B = 64: preprocessing must not be
the bottleneck."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-chinese-llama-alpaca_amd"))
import numpy as np
import torch
from visualcla.preprocess import GpuClipImageProcessor

N, H, W = int(os.environ.get("N", "256")), 480, 640
rng = np.random.default_rng(0)
frames = torch.from_numpy((rng.random((N, H, W, 3)) * 255).astype(np.uint8))
pinned = frames.pin_memory()
proc = GpuClipImageProcessor(size=224, dtype=torch.bfloat16)
out = torch.empty(N, 3, 224, 224, dtype=torch.bfloat16, device="cuda:0")


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


t_batch = timed(lambda: proc.preprocess_batch(pinned, out=out))
dev = pinned.cuda()
t_kern = timed(lambda: proc.preprocess_batch(dev, out=out))
n1 = 32
t_single = timed(lambda: [proc.preprocess_into(frames[i], out[i]) for i in range(n1)], reps=3)
print(f"GPU batched, pinned host -> device -> 2 launches   N={N}: {t_batch*1e3:8.2f} ms  {N/t_batch:10.0f} img/s  ({N*H*W*3/t_batch/1e9:.1f} GB/s over PCIe)")
print(f"GPU batched, frames already on the device           N={N}: {t_kern*1e3:8.2f} ms  {N/t_kern:10.0f} img/s")
print(f"GPU one image per call (pageable copy + 2 launches) N={n1}: {t_single*1e3:8.2f} ms  {n1/t_single:10.0f} img/s")
try:
    from transformers import CLIPImageProcessor
    hf = CLIPImageProcessor()
    k = 8
    t0 = time.perf_counter()
    hf([frames[i].numpy() for i in range(k)], return_tensors="pt")
    t_hf = time.perf_counter() - t0
    print(f"transformers CLIPImageProcessor on the host (1 thread) N={k}: {t_hf*1e3:8.2f} ms  {k/t_hf:10.0f} img/s")
except Exception as e:
    print("CLIPImageProcessor baseline unavailable:", e)
