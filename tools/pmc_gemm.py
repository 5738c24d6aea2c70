#!/usr/bin/env python
"""Workload for the MFMA-utilisation PMC pass: the 256x256 direct-to-LDS GEMM on the LLaMA prefill gate/up shape
(M=8192, N=22016, K=4096, SwiGLU epilogue), 5 launches, random data."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-chinese-llama-alpaca_amd"))
import torch
from visualcla import _lib

dev = "cuda:0"
vit = os.environ.get("VCLA_PMC_SHAPE", "") == "vit"       # ViT fc1: M = 16384 (the whole rounds), N = 4096, K = 1024, bias + quick-GELU
M, N, K = (16384, 4096, 1024) if vit else (8192, 22016, 4096)
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = torch.zeros((N + 127) // 128 * 128, K, dtype=torch.bfloat16, device=dev)
w.normal_(0, 0.02)
out = torch.empty(M, N if vit else N // 2, dtype=torch.bfloat16, device=dev)
bias = torch.randn(N, device=dev) if vit else None
for _ in range(20 if vit else 5):
    _lib.gemm(a, w, N, bias=bias, epilogue=1 if vit else _lib.EPI_SWIGLU, out=out, force_kernel=4)
torch.cuda.synchronize()
print("done")
