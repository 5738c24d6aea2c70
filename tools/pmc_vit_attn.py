#!/usr/bin/env python
"""Workload for the PMC passes of the whole-sequence ViT attention (attn_vit_dma_kernel): B = 64 images x 16 heads x 257 tokens, d = 64, through the
strided fused-qkv layout the engine uses, 20 launches, random data.  One counter per rocprofv3 run (tools/round_end.sh / the block below)."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-chinese-llama-alpaca_amd"))
import torch
from visualcla import _lib

B, H, T, D = 64, 16, 257, 64
q, k, v = ((torch.randn(B, H, T, D, device="cuda:0")).to(torch.bfloat16) for _ in range(3))
out = torch.empty(B, T, H * D, dtype=torch.bfloat16, device="cuda:0")
for _ in range(20):
    _lib.attention(q, k, v, 1 / math.sqrt(D), causal=False, out=out, force_kernel=3)
torch.cuda.synchronize()
print("done")
