#!/usr/bin/env python
"""Workload for the PMC passes: the dominant decode kernel (gate/up SwiGLU GEMV + fused RMSNorm, 180.4 MB of weights per
launch) over 32 distinct weight matrices (5.8 GB footprint, nothing Infinity-Cache resident), 3 rounds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-chinese-llama-alpaca_amd"))
import torch
from visualcla import _lib

D, I = 4096, 11008
dev = "cuda:0"
ws = []
for _ in range(32):
    w = torch.zeros(2 * I, D, dtype=torch.bfloat16, device=dev)
    w.normal_(0, 0.02)
    ws.append(w)
x = torch.randn(1, D, device=dev).to(torch.bfloat16)
gamma = torch.ones(D, device=dev)
out = torch.empty(1, I, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    for w in ws:
        _lib.gemm(x, w, 2 * I, out=out, epilogue=_lib.EPI_SWIGLU, force_kernel=2, norm_gamma=gamma, norm_eps=1e-6)
torch.cuda.synchronize()
print("done")
