#!/usr/bin/env python
"""Per-workgroup phase timeline of the whole-sequence ViT attention (attn_vit_dma_kernel) at the B = 64 shape.  Needs the debug library:
    make -C visual-chinese-llama-alpaca_amd/csrc timeline      (-> tools/libvcla_timeline.so, stamps compiled in)
    VCLA_LIB=tools/libvcla_timeline.so python tools/debug/vit_attn_timeline.py [B]
Stamps (100 MHz wall clock, thread 0 of each workgroup): 0 entry, 1 own Q landed, 2 Q in registers + tiles 2/3 requested, 3 tile 0 ready,
4..7 tiles 0..3 computed, 8 last key folded + O packed, 9 last-row phase done, 10 barrier, 11 stores issued."""
import ctypes as C
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-chinese-llama-alpaca_amd"))
import torch
from visualcla import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H, T, D = 16, 257, 64
L = C.CDLL(os.environ["VCLA_LIB"])
L.vcla_debug_set_vit_timeline.argtypes = [C.c_void_p]
dev = "cuda:0"
q, k, v = ((torch.randn(B, H, T, D, device=dev)).to(torch.bfloat16) for _ in range(3))
out = torch.empty(B, T, H * D, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    _lib.attention(q, k, v, 1 / math.sqrt(D), causal=False, out=out, force_kernel=3)
tl = torch.zeros(B * H, 16, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
L.vcla_debug_set_vit_timeline(tl.data_ptr())
_lib.attention(q, k, v, 1 / math.sqrt(D), causal=False, out=out, force_kernel=3)
torch.cuda.synchronize()
L.vcla_debug_set_vit_timeline(None)
t = tl.cpu().double() * 0.01                      # us
t0 = t[:, 0].min()
names = ["entry", "Q landed", "Q read+t2/3 req", "tile0 ready", "tile0 done", "tile1 done", "tile2 done", "tile3 done", "key+pack", "last row", "barrier", "stores"]
print(f"B={B}: {B * H} workgroups; launch span {t[:, 11].max() - t0:.1f} us")
start = t[:, 0] - t0
first = start < 3.0
if (~first).any():
    print(f"  workgroups starting within 3 us of the first: {int(first.sum())}; the others start at {start[~first].mean():.1f} us on average (min {start[~first].min():.1f}, max {start[~first].max():.1f})")
else:
    print("  all start together")
for grp, nm in ((first, "first wave of workgroups"), (~first, "later workgroups")):
    if not grp.any():
        continue
    print(f"  {nm}: mean time since the workgroup's own entry [us]")
    for i in range(1, 12):
        d = (t[grp, i] - t[grp, 0])
        print(f"    {names[i]:18s} {d.mean():7.2f}   (+{(t[grp, i] - t[grp, i - 1]).mean():5.2f})   max {d.max():6.2f}")
