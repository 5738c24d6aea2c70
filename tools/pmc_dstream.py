#!/usr/bin/env python
"""Workload for the PMC passes of the batch-decode (B = 64) dominant kernel: the gate/up SwiGLU streaming GEMM (kernel 9,
gemm_dstream_kernel<SWIGLU, MT=4>), 180.4 MB of fragment-major weights per launch, over 32 distinct weight matrices (5.8 GB
footprint, nothing Infinity-Cache resident), 3 rounds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-chinese-llama-alpaca_amd"))
import torch
from visualcla import _lib
from visualcla.weights import to_fragment_major

D, I, M = 4096, 11008, 64
dev = "cuda:0"
ws = []
for _ in range(32):
    w = torch.zeros(2 * I, D, dtype=torch.bfloat16, device=dev)
    w.normal_(0, 0.02)
    ws.append((w, to_fragment_major(w)))
af = _lib.to_frag(torch.randn(M, D, device=dev).to(torch.bfloat16))
cf = torch.zeros(I // 32, M // 16, 64, 8, dtype=torch.bfloat16, device=dev)
out = torch.empty(M, I, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    for w, wf in ws:
        _lib.gemm(None, w, 2 * I, out=out, epilogue=_lib.EPI_SWIGLU, force_kernel=9, a_frag=af, m=M, w_frag=wf, c_frag=cf)
torch.cuda.synchronize()
print("done")
