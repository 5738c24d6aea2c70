import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-chinese-llama-alpaca_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from visualcla import _lib
from bench_kernels import timeit_graph, rnd, packw
DEV = "cuda:0"
ws = torch.zeros(64 << 20, dtype=torch.uint8, device=DEV)
for M in (257, 32, 64):
  for tag, N, K, res in (("qkv", 3072, 1024, False), ("o", 1024, 1024, True), ("fc1", 4096, 1024, False), ("fc2", 1024, 4096, True)):
    a = rnd(M, K); w = [packw(N, K) for _ in range(4)]; bias = torch.randn(N, device=DEV)
    r = rnd(M, N) if res else None
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ref = None
    line = f"M={M} {tag:4s} N={N} K={K}: "
    for fk, nm in ((0, "auto+ws"), (14, "ring64x64"), (13, "ring128x96"), (1, "k1 no ws"), (7, "skinny")):
        if fk == 7 and M > 128: continue
        i = [0]
        def run():
            i[0] = (i[0] + 1) % 4
            _lib.gemm(a, w[i[0]], N, bias=bias, residual=r, out=out, force_kernel=fk, splitk_ws=(ws if fk == 0 else None))
        try:
            t = timeit_graph(run, reps=50)
            _lib.gemm(a, w[0], N, bias=bias, residual=r, out=out, force_kernel=fk, splitk_ws=(ws if fk == 0 else None))
            o = out.float().clone()
            if ref is None: ref = o
            line += f"{nm} {t*1e6:6.1f} us (d {float((o-ref).abs().max()):.3f}) | "
        except Exception as e:
            line += f"{nm} ERR {str(e)[:40]} | "
    print(line, flush=True)
