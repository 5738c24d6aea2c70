#!/usr/bin/env python
"""Workload for the round-5 PMC passes: the ring kernel's gate/up SwiGLU GEMM at M = 256 (180.4 MB of bf16 weights per launch, 2 MB activation panel
read by every workgroup out of L2) over 16 distinct weight matrices, 3 rounds; then the 577-token ViT attention at B = 32 (16 heads), 8 launches."""
import ctypes as C
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-chinese-llama-alpaca_amd"))
import torch
from visualcla import _lib

D, I, M = 4096, 11008, 256
dev = "cuda:0"
ws = []
for _ in range(16):
    w = torch.zeros(2 * I, D, dtype=torch.bfloat16, device=dev)
    w.normal_(0, 0.02)
    ws.append(w)
x = torch.randn(M, D, device=dev).to(torch.bfloat16)
out = torch.empty(M, I, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    for w in ws:
        _lib.gemm(x, w, 2 * I, out=out, epilogue=_lib.EPI_SWIGLU, force_kernel=11)
B, H, T, Dh = 32, 16, 577, 64
qkv = torch.randn(B, T, 3 * H * Dh, device=dev).to(torch.bfloat16)
o = torch.empty(B, T, H * Dh, dtype=torch.bfloat16, device=dev)
a = _lib.AttnArgs()
base = qkv.data_ptr()
a.q, a.k, a.v, a.o = base, base + H * Dh * 2, base + 2 * H * Dh * 2, o.data_ptr()
a.q_bs = a.k_bs = a.v_bs = T * 3 * H * Dh
a.q_hs = a.k_hs = a.v_hs = Dh
a.q_rs = a.k_rs = a.v_rs = 3 * H * Dh
a.o_bs, a.o_hs, a.o_rs = T * H * Dh, Dh, H * Dh
a.B, a.H, a.Tq, a.Tk, a.D = B, H, T, T, Dh
a.scale, a.causal, a.force_kernel = 1 / math.sqrt(Dh), 0, 3
for _ in range(8):
    _lib.check(_lib.load().vcla_attention(C.byref(a), _lib.dtype_code(torch.bfloat16), _lib.stream_ptr()))
torch.cuda.synchronize()
print("done")
