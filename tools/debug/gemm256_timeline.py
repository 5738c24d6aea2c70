#!/usr/bin/env python
"""Where does a 256x256 GEMM round go?  Runs the ViT shapes (and one LLaMA prefill shape) on the timeline build of the library
(make -C visual-chinese-llama-alpaca_amd/csrc timeline; per-workgroup wall-clock stamps at entry / first K slab landed / K loop
done / epilogue issued / stores drained) and prints the phase durations averaged over the workgroups, per dispatch round."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ABL = {0: "full kernel", 1: "no fragment reads (MFMA + DMA)", 2: "no MFMAs (LDS reads + DMA)", 3: "no DMA after the first slab (MFMA + LDS reads)"}
if len(sys.argv) < 2:      # one child process per build: the library is chosen at import (VCLA_LIB)
    for abl in (0, 1, 2, 3):
        lib = os.path.join(ROOT, "tools", "libvcla_timeline.so" if abl == 0 else f"libvcla_timeline_abl{abl}.so")
        if os.path.exists(lib):
            subprocess.call([sys.executable, os.path.abspath(__file__), str(abl)], env=dict(os.environ, VCLA_LIB=lib))
    sys.exit(0)
abl = int(sys.argv[1])
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-chinese-llama-alpaca_amd"))
import torch
from visualcla import _lib
L = _lib.load()
L.vcla_debug_set_timeline.argtypes = [C.c_void_p]
DEV = "cuda:0"
def rnd(*s, scale=1.0): return (torch.randn(*s, device=DEV) * scale).to(torch.bfloat16)
def packw(n, k):
    w = torch.zeros((n + 127) // 128 * 128, k, dtype=torch.bfloat16, device=DEV); w[:n] = rnd(n, k, scale=0.02); return w
print(f"==== {ABL[abl]}; VCLA_GEMM_PF={os.environ.get('VCLA_GEMM_PF', '1 (default)')}")
for tag, M, N, K, epi in (("vit qkv", 16384, 3072, 1024, 0), ("vit fc1", 16384, 4096, 1024, 1), ("vit fc2", 16384, 1024, 4096, 0),
                          ("vit fc1, 257-row tiles", 16448, 4096, 1024, 1), ("llama qkv prefill", 8192, 12288, 4096, 0)):
    a, w = rnd(M, K), packw(N, K)
    bias = torch.randn(N, device=DEV)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    nwg = (M // 257 if M % 257 == 0 else M // 256) * ((N + 255) // 256)
    tl = torch.zeros(nwg, 8, dtype=torch.int64, device=DEV)
    for _ in range(3):
        _lib.gemm(a, w, N, bias=bias, epilogue=epi, out=out, force_kernel=4)
    torch.cuda.synchronize()
    L.vcla_debug_set_timeline(tl.data_ptr())
    _lib.gemm(a, w, N, bias=bias, epilogue=epi, out=out, force_kernel=4)
    torch.cuda.synchronize()
    L.vcla_debug_set_timeline(None)
    t = tl.cpu().double() / 100.0          # 100 MHz -> us
    t0 = t[:, 0].min()
    order = t[:, 0].argsort()
    print(f"== {tag}: M={M} N={N} K={K}, {nwg} workgroups; whole launch {t[:, 4].max() - t0:.1f} us (first entry -> last drain)")
    for r in range(min(2, (nwg + 255) // 256)):
        idx = order[r * 256:(r + 1) * 256]
        s = t[idx]
        print(f"   round {r}: entry at {s[:, 0].mean() - t0:7.1f} us (spread {s[:, 0].max() - s[:, 0].min():5.1f}) | first slab {(s[:, 1] - s[:, 0]).mean():5.1f} | "
              f"K loop {(s[:, 2] - s[:, 1]).mean():6.1f} ({(s[:, 2] - s[:, 1]).mean() / (K / 64):5.2f} per K step) | epilogue issue {(s[:, 3] - s[:, 2]).mean():5.1f} | "
              f"store drain {(s[:, 4] - s[:, 3]).mean():5.1f} | total {(s[:, 4] - s[:, 0]).mean():6.1f} us")
