#!/usr/bin/env python
"""Workload for PMC passes of the batch-decode attention kernel: B = 64, H = 32, d = 128, context 192 of a 256-row cache
(algorithmic bytes = 2 * B * H * pos * d * 2 = 201 MB per launch), 8 distinct caches (layers) x 3 rounds."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-chinese-llama-alpaca_amd"))
import torch
from visualcla import _lib
from visualcla.weights import rope_tables

pos = int(os.environ.get("VCLA_PMC_POS", "192"))
B, H, d, ctx = 64, 32, 128, (pos + 64) // 64 * 64
dev = "cuda:0"
L = _lib.load()
cos, sin = (t.to(dev) for t in rope_tables(1024, d, 10000.0))
caches = [(torch.randn(B, H, ctx, d, device=dev).to(torch.bfloat16), torch.randn(B, H, ctx, d, device=dev).to(torch.bfloat16)) for _ in range(8)]
qkv = torch.randn(B, 3 * H * d, device=dev).to(torch.bfloat16)
out = torch.empty(B, H * d, dtype=torch.bfloat16, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(3):
    if r == 2:
        e0.record()
    for kc, vc in caches:
        _lib.check(L.vcla_attn_decode_fused(qkv.data_ptr(), kc.data_ptr(), vc.data_ptr(), cos.data_ptr(), sin.data_ptr(), out.data_ptr(), B, H, d,
                                            ctx, pos, None, None, 0, 1 / math.sqrt(d), 1, 0, _lib.stream_ptr()))
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 8 * 1e-3
print(f"attn_decode B={B} pos={pos}: {t*1e6:.1f} us  {2*B*H*pos*d*2/t/1e9:.0f} GB/s algorithmic")
