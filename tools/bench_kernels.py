#!/usr/bin/env python
"""Per-kernel micro-benchmarks on the VisualCLA-7B shapes (run on the GPU box): GEMM TF/s against the 2.5 PF bf16 MFMA
peak, GEMV / decode-attention GB/s against the 8 TB/s HBM peak, attention TF/s.  Random data (never zero-filled)."""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-chinese-llama-alpaca_amd"))
import torch
from visualcla import _lib

DEV = "cuda:0"


def timeit(fn, reps=20, warm=3):
    reps = int(os.environ.get("VCLA_BENCH_REPS", reps))    # e.g. 400: a sustained run (tens of ms) instead of a ~3 ms burst
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def timeit_graph(fn, reps=20):
    """like timeit, but the launches of fn() are captured into a hipGraph once and REPLAYED: no Python / ctypes time between kernels (a
    ~20 us kernel launched from Python measures the interpreter, not the GPU)"""
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=DEV) * scale).to(torch.bfloat16)


def packw(n, k):
    w = torch.zeros((n + 127) // 128 * 128, k, dtype=torch.bfloat16, device=DEV)
    w[:n] = rnd(n, k, scale=0.02)
    return w


def bench_gemm(tag, M, N, K, epi=0, fk=1):
    a, w = rnd(M, K), packw(N, K)
    out = torch.empty(M, N // 2 if epi == 3 else N, dtype=torch.bfloat16, device=DEV)
    t = timeit(lambda: _lib.gemm(a, w, N, epilogue=epi, out=out, force_kernel=fk))
    tf = 2.0 * M * N * K / t / 1e12
    print(f"gemm  {tag:28s} M={M:6d} N={N:6d} K={K:6d} epi={epi}  {t*1e6:9.1f} us  {tf:8.1f} TF/s  ({tf/2500*100:5.1f}% of bf16 MFMA peak)")


def bench_gemv(tag, M, N, K, epi=0, fused_norm=False):
    a, w = rnd(M, K), packw(N, K)
    gamma = torch.ones(K, device=DEV) if fused_norm else None
    out = torch.empty(M, N // 2 if epi == 3 else N, dtype=torch.bfloat16, device=DEV)
    t = timeit(lambda: _lib.gemm(a, w, N, epilogue=epi, out=out, force_kernel=2, norm_gamma=gamma, norm_eps=1e-6), reps=50)
    gbs = N * K * 2 / t / 1e9
    print(f"gemv  {tag:28s} M={M:6d} N={N:6d} K={K:6d} epi={epi}  {t*1e6:9.1f} us  {gbs:8.1f} GB/s  ({gbs/8000*100:5.1f}% of HBM peak)")


def bench_gemv_tune():
    """Every streaming variant on the real decode shapes, rotating over 8 weight buffers (> 256 MB MALL) so that the
    number is an HBM number, not an Infinity-Cache one."""
    import ctypes as C
    tune_path = os.path.join(ROOT, "tools", "libvcla_tune.so")      # make -C visual-chinese-llama-alpaca_amd/csrc tune
    if not os.path.exists(tune_path):
        print("tools/libvcla_tune.so missing: run `make -C visual-chinese-llama-alpaca_amd/csrc tune` first")
        return
    lib = C.CDLL(tune_path)
    lib.vcla_gemv_tune.restype, lib.vcla_gemv_tune.argtypes = C.c_int, [C.POINTER(_lib.GemmArgs), C.c_int, C.c_void_p]
    lib.vcla_tune_last_error.restype = C.c_char_p
    shapes = [("qkv", 12288, 4096, 0, True), ("o", 4096, 4096, 0, False), ("gate-up", 22016, 4096, 3, True),
              ("down", 4096, 11008, 0, False)]
    NBUF = 8
    for tag, N, K, epi, fused in shapes:
        ws = [packw(N, K) for _ in range(NBUF)]
        x = rnd(1, K)
        gamma = torch.ones(K, device=DEV) if fused else None
        out = torch.empty(1, N // 2 if epi == 3 else N, dtype=torch.bfloat16, device=DEV)
        res = []
        def mk(w):
            a = _lib.GemmArgs()
            a.A, a.lda, a.W, a.C, a.ldc = x.data_ptr(), K, w.data_ptr(), out.data_ptr(), out.shape[1]
            a.M, a.N, a.K, a.epilogue = 1, N, K, epi
            a.norm_gamma, a.norm_eps = _lib.ptr(gamma), 1e-6
            return a
        argsl = [mk(w) for w in ws]
        # production kernel first
        def prod():
            for w in ws:
                _lib.gemm(x, w, N, epilogue=epi, out=out, force_kernel=2, norm_gamma=gamma, norm_eps=1e-6)
        t = timeit(prod, reps=5) / NBUF
        print(f"tune  {tag:8s} production gemv_kernel                 {t*1e6:8.1f} us  {N*K*2/t/1e9:8.1f} GB/s")
        for wpb in (4, 8):
            for xl, nt in ((0, 1), (1, 1), (1, 0)):
                if wpb == 8 and (xl, nt) == (1, 0):
                    continue
                for R in (1, 2, 4, 8):
                    if epi == 3 and R < 2:
                        continue
                    for U in (1, 2, 4):
                        var = R | (U << 8) | (xl << 16) | (nt << 17) | (wpb << 20)
                        def run():
                            for a in argsl:
                                rc = lib.vcla_gemv_tune(C.byref(a), var, _lib.stream_ptr())
                                if rc:
                                    raise RuntimeError(lib.vcla_tune_last_error().decode())
                        try:
                            t = timeit(run, reps=5) / NBUF
                        except Exception as e:
                            print(f"tune  {tag} var R={R} U={U} xlds={xl} nt={nt} wpb={wpb}: {e}")
                            continue
                        res.append((N * K * 2 / t / 1e9, R, U, xl, nt, wpb, t))
        res.sort(reverse=True)
        for gbs, R, U, xl, nt, wpb, t in res[:6] + res[-2:]:
            print(f"tune  {tag:8s} R={R} U={U} xlds={xl} nt={nt} wpb={wpb}            {t*1e6:8.1f} us  {gbs:8.1f} GB/s")
        del ws


def bench_attn(tag, B, H, Tq, Tk, D, causal, fk):
    q, k, v = rnd(B, H, Tq, D), rnd(B, H, Tk, D), rnd(B, H, Tk, D)
    out = torch.empty(B, Tq, H * D, dtype=torch.bfloat16, device=DEV)
    try:
        t = timeit(lambda: _lib.attention(q, k, v, 1 / math.sqrt(D), causal=causal, out=out, force_kernel=fk), reps=10)
    except Exception as e:
        print(f"attn  {tag}: {e}")
        return
    fl = 4.0 * B * H * Tq * Tk * D * (0.5 if causal and Tq == Tk else 1.0)
    print(f"attn  {tag:28s} B={B} H={H} Tq={Tq} Tk={Tk} D={D} kernel={fk}  {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s")


def main():
    which = sys.argv[1:] or ["gemm", "gemv", "attn"]
    if "gemm" in which:
        B = 64
        Mv = B * 257
        print("== GEMM (MFMA 128x128x64 tile kernel)")
        bench_gemm("vit qkv  (B=64)", Mv, 3072, 1024)
        bench_gemm("vit out  (B=64)", Mv, 1024, 1024)
        bench_gemm("vit fc1  (B=64)", Mv, 4096, 1024, epi=1)
        bench_gemm("vit fc2  (B=64)", Mv, 1024, 4096)
        bench_gemm("vit fc1  (B=1)", 257, 4096, 1024, epi=1)
        Ml = B * 128
        bench_gemm("llama qkv (B=64,T=128)", Ml, 12288, 4096)
        bench_gemm("llama o", Ml, 4096, 4096)
        bench_gemm("llama gate-up swiglu", Ml, 22016, 4096, epi=3)
        bench_gemm("llama down", Ml, 4096, 11008)
        bench_gemm("llama qkv (B=1,T=128)", 128, 12288, 4096)
        bench_gemm("llama gate-up (B=1,T=128)", 128, 22016, 4096, epi=3)
        bench_gemm("llama down (B=1,T=128)", 128, 4096, 11008)
        bench_gemm("square 4096", 4096, 4096, 4096)
        bench_gemm("square 8192", 8192, 8192, 8192)
        bench_gemm("decode M=64 qkv (tile kernel)", 64, 12288, 4096)
        bench_gemm("decode M=64 gate-up", 64, 22016, 4096, epi=3)
        for fk, nm in ((4, "256x256 glds +sched"), (5, "256x256 glds")):
            print(f"== GEMM ({nm})")
            bench_gemm("vit qkv  (B=64)", Mv, 3072, 1024, fk=fk)
            bench_gemm("vit out  (B=64)", Mv, 1024, 1024, fk=fk)
            bench_gemm("vit fc1  (B=64)", Mv, 4096, 1024, epi=1, fk=fk)
            bench_gemm("vit fc2  (B=64)", Mv, 1024, 4096, fk=fk)
            bench_gemm("llama qkv (B=64,T=128)", Ml, 12288, 4096, fk=fk)
            bench_gemm("llama o", Ml, 4096, 4096, fk=fk)
            bench_gemm("llama gate-up swiglu", Ml, 22016, 4096, epi=3, fk=fk)
            bench_gemm("llama down", Ml, 4096, 11008, fk=fk)
            bench_gemm("square 4096", 4096, 4096, 4096, fk=fk)
            bench_gemm("square 8192", 8192, 8192, 8192, fk=fk)
    if "ring" in which:
        print("== 129 - 256-row decode GEMMs (LLaMA-7B layer shapes, rotating 8 weight matrices = no Infinity-Cache reuse): the round-4 dispatch")
        print("   (k1: 128 x 128 tiles + K slices + reduce launch) vs the ring kernel (k11 auto tile; k12 / k13 / k14 = 256x96 / 128x96 / 64x64), bf16 and fp8 weights")
        from visualcla.weights import quantize_fp8_rows
        skws = torch.zeros(64 << 20, dtype=torch.uint8, device=DEV)
        for M in [int(x) for x in os.environ.get("VCLA_BENCH_MS", "256,192,129").split(",")]:
            tot = {}
            for tag, N, K, epi in (("qkv", 12288, 4096, 0), ("o", 4096, 4096, 0), ("gate-up", 22016, 4096, 3), ("down", 4096, 11008, 0)):
                a = rnd(M, K)
                ws = [packw(N, K) for _ in range(8)]
                n_out = N // 2 if epi == 3 else N
                out = torch.empty(M, n_out, dtype=torch.bfloat16, device=DEV)
                res = rnd(M, n_out) if tag in ("o", "down") else None
                q8s = [quantize_fp8_rows(w) for w in ws]
                from visualcla.weights import to_slab_major
                a_sl = to_slab_major(a)
                w_sls = [to_slab_major(w) for w in ws]
                q_sls = [to_slab_major(q) for q, _ in q8s]
                fks = [int(x) for x in os.environ.get("VCLA_BENCH_FKS", "1,11,12,13,14").split(",")]
                for fk, fp8, slab in [(f, p8, sl) for sl in (False, True) for p8 in (False, True) for f in fks]:
                    if (fk > 12 and epi == 3) or (fk == 1 and (fp8 or slab)):
                        continue
                    def run():
                        for w, (q, sc), wsl, qsl in zip(ws, q8s, w_sls, q_sls):
                            _lib.gemm(a, w, N, epilogue=epi, out=out, residual=res, force_kernel=fk, splitk_ws=skws if fk == 1 else None,
                                      w_q8=q if (fp8 and not slab) else None, w_scale=sc if fp8 else None,
                                      a_slab=a_sl if slab else None, w_slab=wsl if (slab and not fp8) else None, w_q8_slab=qsl if (slab and fp8) else None)
                    t = timeit_graph(run, reps=10) / len(ws)
                    gbs = N * K * (1 if fp8 else 2) / t / 1e9
                    tf = 2.0 * M * N * K / t / 1e12
                    print(f"ring  M={M:3d} {tag:8s} k{fk:<2d} {'fp8 ' if fp8 else 'bf16'} {'slab-major' if slab else 'row-major '} N={N:6d} K={K:6d}  {t*1e6:8.1f} us  {gbs:7.1f} GB/s of W  {tf:7.1f} TF/s")
                    key = (fk, fp8, slab)
                    tot[key] = tot.get(key, 0.0) + t
                del w_sls, q_sls
                del ws, q8s
            print(f"ring  M={M:3d} layer sums (qkv + o + gate/up + down): " + "  ".join(f"k{fk}{'/fp8' if f8 else ''}{'/slab' if sl else ''}={v*1e6:.0f}us" for (fk, f8, sl), v in tot.items() if fk in (1, 11)))
    if "ringwf" in which:
        print("== ring kernel (k11 / k14), bf16: weight pieces from the row-major matrix (8-row gathers) vs from the fragment-major twin (1 KiB contiguous), hipGraph replay over 8 matrices")
        from visualcla.weights import to_fragment_major
        for M in [int(x) for x in os.environ.get("VCLA_BENCH_MS", "256,192,129").split(",")]:
            tot = {}
            for tag, N, K, epi in (("qkv", 12288, 4096, 0), ("o", 4096, 4096, 0), ("gate-up", 22016, 4096, 3), ("down", 4096, 11008, 0)):
                a = rnd(M, K)
                ws = [packw(N, K) for _ in range(8)]
                wfs = [to_fragment_major(w) for w in ws]
                n_out = N // 2 if epi == 3 else N
                out = torch.empty(M, n_out, dtype=torch.bfloat16, device=DEV)
                res = rnd(M, n_out) if tag in ("o", "down") else None
                for fk, frag in [(f, fr) for f in (11, 12, 13, 14) for fr in (False, True)]:
                    if fk > 12 and epi == 3:
                        continue
                    def run():
                        for w, wf in zip(ws, wfs):
                            _lib.gemm(a, w, N, epilogue=epi, out=out, residual=res, force_kernel=fk, w_frag=wf if frag else None)
                    t = timeit_graph(run, reps=10) / len(ws)
                    print(f"ringwf M={M:3d} {tag:8s} k{fk} {'fragment-major' if frag else 'row-major     '} N={N:6d} K={K:6d}  {t*1e6:8.1f} us  {N*K*2/t/1e9:7.1f} GB/s of W")
                    tot[(fk, frag)] = tot.get((fk, frag), 0.0) + t
                del ws, wfs
            print(f"ringwf M={M:3d} layer sums: " + "  ".join(f"k{fk}{'/frag' if fr else ''}={v*1e6:.0f}us" for (fk, fr), v in tot.items() if fk == 11))
    if "slab256" in which:
        print("== 256 x 256 direct-to-LDS kernel (k4): row-major operands (8-row x 128-byte DMA pieces) vs slab-major A / W (1 KiB contiguous pieces), hipGraph replay")
        from visualcla.weights import to_slab_major
        shapes = [("vit qkv", 64 * 257, 3072, 1024, 0, True), ("vit out", 64 * 257, 1024, 1024, 0, True), ("vit fc1", 64 * 257, 4096, 1024, 1, True),
                  ("vit fc2", 64 * 257, 1024, 4096, 0, True), ("vit fc1 M=16384", 16384, 4096, 1024, 1, True),
                  ("llama qkv", 8192, 12288, 4096, 0, False), ("llama o", 8192, 4096, 4096, 0, False), ("llama gate-up", 8192, 22016, 4096, 3, False),
                  ("llama down", 8192, 4096, 11008, 0, False)]
        for tag, M, N, K, epi, has_bias in shapes:
            a = rnd(M, K)
            nw = 4
            ws = [packw(N, K) for _ in range(nw)]
            bias = torch.randn(N, device=DEV) if has_bias else None
            out = torch.empty(M, N // 2 if epi == 3 else N, dtype=torch.bfloat16, device=DEV)
            a_sl = to_slab_major(a)
            w_sls = [to_slab_major(w) for w in ws]
            ref = None
            for nm, use_a, use_w in (("row-major A, W", False, False), ("slab A, slab W", True, True), ("slab W only  ", False, True), ("slab A only  ", True, False)):
                def run():
                    for w, wsl in zip(ws, w_sls):
                        _lib.gemm(None if use_a else a, w, N, bias=bias, epilogue=epi, out=out, force_kernel=4, a_slab=a_sl if use_a else None,
                                  w_slab=wsl if use_w else None, m=M)
                t = timeit_graph(run, reps=10) / nw
                tf = 2.0 * M * N * K / t / 1e12
                if ref is None:
                    ref = out.clone()
                same = torch.equal(out, ref)
                print(f"slab256 {tag:16s} {nm} M={M:6d} N={N:6d} K={K:6d}  {t*1e6:8.1f} us  {tf:7.1f} TF/s ({tf/25:5.1f}% of peak)  {'== row-major result' if same else 'DIFFERS'}")
            del ws, w_sls
    if "vit" in which:
        print("== ViT GEMMs at B=64 through AUTO dispatch (256x256 whole rounds + 64-row tail) vs forced kernels")
        Mv = 64 * 257
        skws = torch.zeros(32 << 20, dtype=torch.uint8, device=DEV)
        for tag, N, K, epi in (("qkv", 3072, 1024, 0), ("out", 1024, 1024, 0), ("fc1", 4096, 1024, 1), ("fc2", 1024, 4096, 0)):
            a, w = rnd(Mv, K), packw(N, K)
            bias = torch.randn(N, device=DEV)              # every CLIP linear carries a bias
            out = torch.empty(Mv, N, dtype=torch.bfloat16, device=DEV)
            for fk, nm in ((0, "auto"), (4, "256 "), (1, "128 ")):
                t = timeit(lambda: _lib.gemm(a, w, N, bias=bias, epilogue=epi, out=out, force_kernel=fk, splitk_ws=skws))
                tf = 2.0 * Mv * N * K / t / 1e12
                print(f"vit   {tag:4s} {nm} M={Mv} N={N:5d} K={K:5d}  {t*1e6:8.1f} us  {tf:7.1f} TF/s ({tf/25:.1f}% of peak)")
    if "vit1" in which:
        print("== ONE image through the ViT / the resampler (257 / 64 rows; bias, residual on out / fc2): the K-slice dispatch of rounds 3 - 5 (auto + workspace: 128 x 128 tiles,")
        print("   K slices, reduce launch) vs the ring tiles over the full K (k14 = 64 x 64, k13 = 128 x 96), graph-replayed, rotating 4 weight matrices")
        skws = torch.zeros(64 << 20, dtype=torch.uint8, device=DEV)
        for M in (257, 64):
            for tag, N, K, res in (("qkv", 3072, 1024, False), ("out", 1024, 1024, True), ("fc1", 4096, 1024, False), ("fc2", 1024, 4096, True)):
                a, ws, bias = rnd(M, K), [packw(N, K) for _ in range(4)], torch.randn(N, device=DEV)
                r = rnd(M, N) if res else None
                out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
                line = f"vit1 M={M:3d} {tag:4s} N={N:5d} K={K:5d}: "
                for fk, nm, env in ((0, "K slices + reduce", "0"), (14, "ring 64x64", "1"), (13, "ring 128x96", "1")):
                    os.environ["VCLA_RING_VIT"] = env
                    i = [0]
                    def run():
                        i[0] = (i[0] + 1) % 4
                        _lib.gemm(a, ws[i[0]], N, bias=bias, residual=r, out=out, force_kernel=fk, splitk_ws=(skws if fk == 0 else None))
                    line += f"{nm} {timeit_graph(run, reps=50) * 1e6:6.1f} us | "
                os.environ.pop("VCLA_RING_VIT", None)
                print(line, flush=True)
    if "vittail" in which:
        print("== the 64-row ragged-M tail of the ViT GEMMs at B=64: split-K panel + reduce (8) vs skinny, one launch (7)")
        skws = torch.zeros(32 << 20, dtype=torch.uint8, device=DEV)
        for tag, N, K, epi in (("qkv", 3072, 1024, 0), ("out", 1024, 1024, 0), ("fc1", 4096, 1024, 1), ("fc2", 1024, 4096, 0)):
            a, w = rnd(64, K), packw(N, K)
            bias = torch.randn(N, device=DEV)
            out = torch.empty(64, N, dtype=torch.bfloat16, device=DEV)
            res = rnd(64, N) if tag in ("out", "fc2") else None
            for fk in (8, 7):
                t = timeit(lambda: _lib.gemm(a, w, N, bias=bias, epilogue=epi, out=out, residual=res, force_kernel=fk, splitk_ws=skws), reps=50)
                print(f"vittail {tag:4s} kernel {fk} M=64 N={N:5d} K={K:5d}  {t*1e6:8.1f} us")
    if "skinny" in which:
        print("== skinny (k7) / panel split-K (k8) MFMA GEMM, W streamed once, rotating 4 weight buffers")
        skws = torch.zeros(32 << 20, dtype=torch.uint8, device=DEV)
        for M in (2, 16, 32, 64, 128):
            for tag, N, K, epi in (("qkv", 12288, 4096, 0), ("o", 4096, 4096, 0), ("gate-up", 22016, 4096, 3), ("down", 4096, 11008, 0)):
                a = rnd(M, K)
                ws = [packw(N, K) for _ in range(4)]
                out = torch.empty(M, N // 2 if epi == 3 else N, dtype=torch.bfloat16, device=DEV)
                from visualcla.weights import to_fragment_major
                wfs = [to_fragment_major(w) for w in ws]
                for fk, frag in ((7, False), (8, False), (8, True)):
                    def run():
                        for w, wf in zip(ws, wfs):
                            _lib.gemm(a, w, N, epilogue=epi, out=out, force_kernel=fk, splitk_ws=skws, w_frag=wf if frag else None)
                    t = timeit(run, reps=10) / 4
                    print(f"k{fk}{'f' if frag else ' '} {tag:8s} M={M:4d} N={N:6d} K={K:6d}  {t*1e6:8.1f} us  {N*K*2/t/1e9:8.1f} GB/s  {2.0*M*N*K/t/1e12:7.1f} TF/s")
                del ws, wfs
    if "gemv1" in which:
        # B = 1 decode GEMVs, rotating over 6 weight buffers (HBM figures); VCLA_GEMV1X=0 selects the runtime-K kernel
        print("== M = 1 decode GEMVs; env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("VCLA_")))
        tot = 0.0
        for tag, N, K, epi, norm, f32 in (("qkv", 12288, 4096, 0, True, False), ("o", 4096, 4096, 0, False, False), ("gate-up", 22016, 4096, 3, True, False),
                                          ("down", 4096, 11008, 0, False, False), ("lm_head", 49958, 4096, 0, True, True)):
            a = rnd(1, K)
            ws = [packw(N, K) for _ in range(6)]
            gamma = torch.ones(K, device=DEV) if norm else None
            res = None if (epi == 3 or f32) else rnd(1, N)
            out = torch.empty(1, N // 2 if epi == 3 else N, dtype=torch.float32 if f32 else torch.bfloat16, device=DEV)
            def run():
                for w in ws:
                    _lib.gemm(a, w, N, epilogue=epi, out=out, out_f32=f32, residual=res, norm_gamma=gamma, norm_eps=1e-6)
            t = timeit(run, reps=20) / len(ws)
            if tag != "lm_head":
                tot += t
            print(f"gemv1 {tag:8s} N={N:6d} K={K:6d}  {t*1e6:7.2f} us  {N*K*2/t/1e9:7.0f} GB/s")
            del ws
        print(f"gemv1 layer GEMVs: {tot*1e6:.1f} us ({404.8e6/tot/1e9:.0f} GB/s over 404.8 MB)")
    if "dstream" in which:
        # batch-decode GEMMs: split-K panel kernel (8, + its reduce launches) vs the streaming kernel (9), bf16 and fp8 weights,
        # rotating over 4 weight buffers so the figures are HBM figures
        from visualcla.weights import to_fragment_major, quantize_fp8_rows, to_fragment_pair_major_fp8
        print("== batch-decode GEMMs: panel split-K (k8) vs streaming (k9); env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("VCLA_")))
        skws = torch.zeros(64 << 20, dtype=torch.uint8, device=DEV)
        for M in [int(x) for x in os.environ.get("VCLA_BENCH_MS", "64,32,16").split(",")]:
            tot = {"k8": 0.0, "k9": 0.0, "k8q": 0.0, "k9q": 0.0}
            for tag, N, K, epi in (("qkv", 12288, 4096, 0), ("o", 4096, 4096, 0), ("gate-up", 22016, 4096, 3), ("down", 4096, 11008, 0),
                                   ("lm_head", 49958, 4096, 0)):
                n_out = N // 2 if epi == 3 else N
                a = rnd(M, K)
                af = _lib.to_frag(a)
                nb = 2 if tag == "lm_head" else 4
                nb = int(os.environ.get("VCLA_BENCH_NB", nb))     # 1: the same matrix every launch (<= 256 MB: served from the Infinity Cache)
                ws = [packw(N, K) for _ in range(nb)]
                wfs = [to_fragment_major(w) for w in ws]
                out = torch.empty(M, n_out, dtype=torch.float32 if tag == "lm_head" else torch.bfloat16, device=DEV)
                res = rnd(M, n_out) if tag in ("o", "down") else None
                gam = torch.ones(n_out, device=DEV)
                hn = torch.empty(M, n_out, dtype=torch.bfloat16, device=DEV)
                f32 = tag == "lm_head"
                def run8():
                    for w, wf in zip(ws, wfs):
                        if res is not None:    # as the engine runs it: fused reduce + next RMSNorm
                            _lib.gemm(a, w, N, epilogue=epi, out=out, residual=res, force_kernel=8, splitk_ws=skws, w_frag=wf, post_norm_gamma=gam, post_norm_eps=1e-6, post_norm_out=hn)
                        else:
                            _lib.gemm(a, w, N, epilogue=epi, out=out, out_f32=f32, force_kernel=8, splitk_ws=skws, w_frag=wf)
                def run9():
                    for w, wf in zip(ws, wfs):
                        _lib.gemm(None, w, N, epilogue=epi, out=out, out_f32=f32, residual=res, force_kernel=9, a_frag=af, m=M, w_frag=wf)
                t8, t9 = timeit(run8, reps=10) / nb, timeit(run9, reps=10) / nb
                if tag == "qkv" and M > 32:     # as the engine runs it at M > 32: two K slices, raw fp32 partials (the attention sums them), no reduce launch
                    def run9r():
                        for w, wf in zip(ws, wfs):
                            _lib.gemm(None, w, N, out=out, force_kernel=9, a_frag=af, m=M, w_frag=wf, splitk_ws=skws, ds_splitk=2, ds_raw_partials=True)
                    print(f"M={M:3d} qkv      k9 split in 2 K slices, raw partials (wide kernel, no reduce launch): {timeit(run9r, reps=10) / nb * 1e6:6.1f} us   (unsplit k9 {t9*1e6:.1f} us)")
                if res is not None:     # o / down: the split-K form of the streaming kernel (+ its reduce launch), as the engine runs it
                    cfr = torch.zeros(n_out // 32, (M + 15) // 16, 64, 8, dtype=torch.bfloat16, device=DEV)
                    ssq = torch.zeros(M, n_out // 16, dtype=torch.float32, device=DEV)
                    line = []
                    for S_ in (1, 2, 4, 8):
                        def run9s():
                            for w, wf in zip(ws, wfs):
                                _lib.gemm(None, w, N, out=out, residual=res, force_kernel=9, a_frag=af, m=M, w_frag=wf, splitk_ws=skws, ds_splitk=S_,
                                          c_frag=cfr, c_frag_gamma=gam, c_row_ssq=ssq)
                        line.append(f"S={S_} {timeit(run9s, reps=10) / nb * 1e6:6.1f} us")
                    print(f"M={M:3d} {tag:8s} k9 with deferred-norm outputs, split-K: " + " | ".join(line))
                qs = [quantize_fp8_rows(w) for w in ws]
                qfs = [to_fragment_pair_major_fp8(q) for q, _ in qs]
                def run8q():
                    for w, (q, sc), qf in zip(ws, qs, qfs):
                        _lib.gemm(a, w, N, epilogue=epi, out=out, out_f32=f32, residual=res, force_kernel=8, splitk_ws=skws, w_q8=q, w_q8_frag=qf, w_scale=sc)
                def run9q():
                    for w, (q, sc), qf in zip(ws, qs, qfs):
                        _lib.gemm(None, w, N, epilogue=epi, out=out, out_f32=f32, residual=res, force_kernel=9, a_frag=af, m=M, w_q8_frag=qf, w_scale=sc)
                t8q, t9q = timeit(run8q, reps=10) / nb, timeit(run9q, reps=10) / nb
                by = ws[0].shape[0] * K * 2
                print(f"M={M:3d} {tag:8s} N={N:6d} K={K:6d}  k8 {t8*1e6:7.1f} us {by/t8/1e9:6.0f} GB/s | k9 {t9*1e6:7.1f} us {by/t9/1e9:6.0f} GB/s"
                      f" || fp8: k8 {t8q*1e6:7.1f} us {by/2/t8q/1e9:6.0f} GB/s | k9 {t9q*1e6:7.1f} us {by/2/t9q/1e9:6.0f} GB/s")
                if tag != "lm_head":
                    for k_, t_ in (("k8", t8), ("k9", t9), ("k8q", t8q), ("k9q", t9q)):
                        tot[k_] += t_
                del ws, wfs, qs, qfs
            xn = rnd(M, 4096)
            gam = torch.ones(4096, device=DEV)
            pk = _lib.rmsnorm_pack(xn, gam, 1e-6)
            tp = timeit(lambda: _lib.rmsnorm_pack(xn, gam, 1e-6, out=pk), reps=50)
            print(f"M={M:3d} layer GEMMs: k8 (incl. reduce+norm launches) {tot['k8']*1e6:.1f} us | k9 {tot['k9']*1e6:.1f} us + 2 x rmsnorm_pack {tp*1e6:.1f} us"
                  f" = {(tot['k9'] + 2 * tp)*1e6:.1f} us ({404e6/(tot['k9'] + 2 * tp)/1e9:.0f} GB/s over 404 MB) || fp8: k8 {tot['k8q']*1e6:.1f} | k9 {tot['k9q']*1e6:.1f} us")
    if "dec256" in which:
        # decode batches of 65 - 256 rows: the row-major MFMA tile kernels through the product dispatch (what engine.hip's llama_layer
        # issues), rotating over 4 weight buffers.  VCLA_MFMA128_S forces the K-slice count of the 128-tile kernel, VCLA_BENCH_FK the kernel.
        print("== decode GEMMs at 65 <= M <= 256 (product dispatch); env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("VCLA_")))
        skws = torch.zeros(64 << 20, dtype=torch.uint8, device=DEV)
        fk = int(os.environ.get("VCLA_BENCH_FK", "0"))
        for M in [int(x) for x in os.environ.get("VCLA_BENCH_MS", "256,192,128").split(",")]:
            tot = 0.0
            for tag, N, K, epi in (("qkv", 12288, 4096, 0), ("o", 4096, 4096, 0), ("gate-up", 22016, 4096, 3), ("down", 4096, 11008, 0), ("lm_head", 49958, 4096, 0)):
                n_out = N // 2 if epi == 3 else N
                a = rnd(M, K)
                nb = 2 if tag == "lm_head" else 4
                ws = [packw(N, K) for _ in range(nb)]
                f32 = tag == "lm_head"
                out = torch.empty(M, n_out, dtype=torch.float32 if f32 else torch.bfloat16, device=DEV)
                res = rnd(M, n_out) if tag in ("o", "down") else None
                gam = torch.ones(n_out, device=DEV)
                hn = torch.empty(M, n_out, dtype=torch.bfloat16, device=DEV)
                def run():
                    for w in ws:
                        if res is not None:
                            _lib.gemm(a, w, N, epilogue=epi, out=out, residual=res, force_kernel=fk, splitk_ws=skws, post_norm_gamma=gam, post_norm_eps=1e-6, post_norm_out=hn)
                        else:
                            _lib.gemm(a, w, N, epilogue=epi, out=out, out_f32=f32, force_kernel=fk, splitk_ws=skws)
                t = timeit(run, reps=10) / nb
                by = ws[0].shape[0] * K * 2
                print(f"M={M:3d} {tag:8s} N={N:6d} K={K:6d}  {t*1e6:7.1f} us  {by/t/1e9:6.0f} GB/s  {2.0*M*N*K/t/1e12:7.1f} TF/s")
                if tag != "lm_head":
                    tot += t
                del ws
            print(f"M={M:3d} layer GEMMs (incl. reduce / norm launches): {tot*1e6:.1f} us")
    if "fp8mfma" in which:
        # BASELINE configs[4]: the prefill / ViT GEMM shapes on the bf16 MFMA kernel (4) vs fp8 x fp8 on the fp8 MFMA pipe (10)
        from visualcla.weights import quantize_fp8_rows
        print("== bf16 256x256x64 (k4) vs fp8 256x256x128 (k10); TF/s against 2.5 PF (bf16) and 5 PF (fp8) dense peaks")
        for tag, M, N, K, epi in (("llama qkv  B=64,T=128", 8192, 12288, 4096, 0), ("llama o", 8192, 4096, 4096, 0), ("llama gate-up", 8192, 22016, 4096, 3),
                                  ("llama down", 8192, 4096, 11008, 0), ("llama qkv  B=32,T=128", 4096, 12288, 4096, 0),
                                  ("vit fc1 B=64", 16448, 4096, 1024, 1), ("vit fc2 B=64", 16448, 1024, 4096, 0), ("vit qkv B=32 336px", 32 * 577, 3072, 1024, 0)):
            a, w = rnd(M, K), packw(N, K)
            out = torch.empty(M, N // 2 if epi == 3 else N, dtype=torch.bfloat16, device=DEV)
            t4 = timeit(lambda: _lib.gemm(a, w, N, epilogue=epi, out=out, force_kernel=4), reps=10)
            wq, ws_ = quantize_fp8_rows(w)
            aq, as_ = _lib.quant_fp8_rows(a)
            t10 = timeit(lambda: _lib.gemm(None, w, N, epilogue=epi, out=out, force_kernel=10, a_q8=aq, a_scale=as_, w_q8=wq, w_scale=ws_), reps=10)
            tq = timeit(lambda: _lib.quant_fp8_rows(a), reps=10)
            fl = 2.0 * M * N * K
            print(f"{tag:24s} M={M:6d} N={N:6d} K={K:6d}  k4 {t4*1e6:8.1f} us {fl/t4/1e12:7.1f} TF/s ({fl/t4/2.5e13:.1f}%) | k10 {t10*1e6:8.1f} us {fl/t10/1e12:7.1f} TF/s "
                  f"({fl/t10/5e13:.1f}% of 5 PF) | quantise A {tq*1e6:6.1f} us")
    if "panel" in which:
        print(f"== panel split-K kernel (fragment-major W), M=64, env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("VCLA_PANEL")))
        skws = torch.zeros(64 << 20, dtype=torch.uint8, device=DEV)
        from visualcla.weights import to_fragment_major
        tot = tot8 = 0.0
        for M in (int(os.environ.get("VCLA_BENCH_M", "64")),):
            for tag, N, K, epi in (("qkv", 12288, 4096, 0), ("o", 4096, 4096, 0), ("gate-up", 22016, 4096, 3), ("down", 4096, 11008, 0)):
                a = rnd(M, K)
                ws = [packw(N, K) for _ in range(4)]
                wfs = [to_fragment_major(w) for w in ws]
                out = torch.empty(M, N // 2 if epi == 3 else N, dtype=torch.bfloat16, device=DEV)
                def run():
                    for w, wf in zip(ws, wfs):
                        _lib.gemm(a, w, N, epilogue=epi, out=out, force_kernel=8, splitk_ws=skws, w_frag=wf)
                t = timeit(run, reps=10) / 4
                tot += t
                print(f"k8f {tag:8s} M={M:4d} N={N:6d} K={K:6d}  {t*1e6:8.1f} us  {N*K*2/t/1e9:8.1f} GB/s")
                def run1():   # no workspace -> one K slice per column tile, epilogue in the kernel, no reduce launch
                    for w, wf in zip(ws, wfs):
                        _lib.gemm(a, w, N, epilogue=epi, out=out, force_kernel=8, splitk_ws=None, w_frag=wf)
                t1 = timeit(run1, reps=10) / 4
                print(f"k8f {tag:8s} (no split-K)               {t1*1e6:8.1f} us  {N*K*2/t1/1e9:8.1f} GB/s")
                from visualcla.weights import quantize_fp8_rows, to_fragment_pair_major_fp8
                qs = [quantize_fp8_rows(w) for w in ws]
                qfs = [to_fragment_pair_major_fp8(q) for q, _ in qs]
                def run8():
                    for w, (q, sc), qf in zip(ws, qs, qfs):
                        _lib.gemm(a, w, N, epilogue=epi, out=out, force_kernel=8, splitk_ws=skws, w_q8=q, w_q8_frag=qf, w_scale=sc)
                t8 = timeit(run8, reps=10) / 4
                tot8 += t8
                print(f"k8q {tag:8s} M={M:4d} N={N:6d} K={K:6d}  {t8*1e6:8.1f} us  {N*K/t8/1e9:8.1f} GB/s (fp8 bytes)")
                del ws, wfs, qs, qfs
        print(f"layer total {tot*1e6:.1f} us  ({404e6/tot/1e9:.0f} GB/s over the 404 MB of layer weights); fp8 copies: {tot8*1e6:.1f} us")
    if "gemv1" in which:
        print("== M=1 GEMV on the decode shapes, rotating over 8 weight buffers (HBM-resident); env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("VCLA_GEMV1")))
        tot = 0.0
        for tag, N, K, epi, fused in (("qkv", 12288, 4096, 0, True), ("o", 4096, 4096, 0, False), ("gate-up", 22016, 4096, 3, True), ("down", 4096, 11008, 0, False)):
            ws = [packw(N, K) for _ in range(8)]
            x = rnd(1, K)
            gamma = torch.ones(K, device=DEV) if fused else None
            out = torch.empty(1, N // 2 if epi == 3 else N, dtype=torch.bfloat16, device=DEV)
            def run():
                for w in ws:
                    _lib.gemm(x, w, N, epilogue=epi, out=out, force_kernel=2, norm_gamma=gamma, norm_eps=1e-6)
            t = timeit(run, reps=10) / 8
            tot += t
            print(f"gemv1 {tag:8s} N={N:6d} K={K:6d}  {t*1e6:8.2f} us  {N*K*2/t/1e9:8.1f} GB/s")
            del ws
        print(f"layer total {tot*1e6:.1f} us  ({404e6/tot/1e9:.0f} GB/s)")
    if "gemv" in which:
        print("== GEMV (decode, weight streaming)")
        for M in (1, 4, 8):
            bench_gemv("llama qkv", M, 12288, 4096, fused_norm=True)
            bench_gemv("llama o", M, 4096, 4096)
            bench_gemv("llama gate-up swiglu", M, 22016, 4096, epi=3, fused_norm=True)
            bench_gemv("llama down", M, 4096, 11008)
            bench_gemv("lm_head", M, 49958, 4096, fused_norm=True)
    if "tune" in which:
        print("== GEMV tuning (rotating buffers, HBM-resident)")
        bench_gemv_tune()
    if "attndec" in which:
        # fused decode attention (RoPE + append + attend), rotating over cache buffers so K / V rows come from HBM as in the model
        from visualcla.weights import rope_tables
        print("== decode attention; env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("VCLA_")))
        H, d = 32, 128
        cos, sin = (t.to(DEV) for t in rope_tables(1024, d, 10000.0))
        for B, ctx_max, poss in ((64, 256, (128, 192, 254)), (32, 704, (640,)), (1, 256, (128, 192, 254))):
            nbuf = int(os.environ.get("VCLA_BENCH_NBUF", 4 if B > 1 else 32))   # 1: the same K / V buffers every launch (<= 256 MB: Infinity-Cache resident)
            kv8 = int(os.environ.get("VCLA_BENCH_KV8", "0"))        # 1: e4m3 cache rows (VCLA_KV_FP8)
            mk = (lambda: (torch.randn(B, H, ctx_max, d, device=DEV)).to(torch.float8_e4m3fn).view(torch.uint8)) if kv8 else (lambda: rnd(B, H, ctx_max, d))
            kcs = [mk() for _ in range(nbuf)]
            vcs = [mk() for _ in range(nbuf)]
            qkv = rnd(B, 3 * H * d)
            frag = 1 if 2 <= B <= 64 else 0
            out = torch.zeros((H * d) // 32, (B + 15) // 16, 64, 8, dtype=torch.bfloat16, device=DEV) if frag else torch.empty(B, H * d, dtype=torch.bfloat16, device=DEV)
            L = _lib.load()
            for pos in poss:
                qparts = torch.randn(2, B, 3 * H * d, device=DEV)
                ssq = torch.rand(B, 16, device=DEV) * 100 + 1
                if int(os.environ.get("VCLA_BENCH_QP", "0")) and B * H >= 1024:
                    def run():
                        for kc, vc in zip(kcs, vcs):
                            _lib.check(L.vcla_attn_decode_fused_parts(qparts.data_ptr(), B * 3 * H * d, ssq.data_ptr(), None, 1e-6, kc.data_ptr(), vc.data_ptr(), cos.data_ptr(),
                                                                      sin.data_ptr(), out.data_ptr(), B, H, d, ctx_max, pos, None, None, 0, 1 / math.sqrt(d),
                                                                      _lib.dtype_code(torch.bfloat16) | (0x100 if kv8 else 0), frag, _lib.stream_ptr()))
                else:
                  def run():
                    for kc, vc in zip(kcs, vcs):
                        _lib.check(L.vcla_attn_decode_fused(qkv.data_ptr(), kc.data_ptr(), vc.data_ptr(), cos.data_ptr(), sin.data_ptr(), out.data_ptr(),
                                                            B, H, d, ctx_max, pos, None, None, 0, 1 / math.sqrt(d), _lib.dtype_code(torch.bfloat16) | (0x100 if kv8 else 0), frag,
                                                            _lib.stream_ptr()))
                t = timeit(run, reps=10) / nbuf
                by = B * H * pos * d * 2 * (1 if kv8 else 2)
                print(f"attndec B={B:3d} ctx={pos:4d}  {t*1e6:8.1f} us  {by/t/1e9:8.1f} GB/s of K/V rows  ({by/t/8e12*100:5.1f}% of HBM peak)")
            del kcs, vcs
    if "vitattn" in which:
        print("== MFMA flash attention, vision shapes; env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("VCLA_")))
        bench_attn("vit 224px (B=64) tile-by-tile", 64, 16, 257, 257, 64, False, 2)
        bench_attn("vit 224px (B=64) whole-seq", 64, 16, 257, 257, 64, False, 3)
        bench_attn("vit 224px (B=16) whole-seq", 16, 16, 257, 257, 64, False, 3)
        bench_attn("vit 224px (B=256) whole-seq", 256, 16, 257, 257, 64, False, 3)
        bench_attn("vit 336px (B=32)", 32, 16, 577, 577, 64, False, 2)
        bench_attn("resampler (B=64)", 64, 16, 64, 321, 64, False, 2)
        bench_attn("llama prefill (B=64,T=128)", 64, 32, 128, 128, 128, True, 2)
    if "attn577" in which:
        print("== ViT self-attention at 336 px (577 tokens, d = 64, fused-qkv strides), B = 32 x 16 heads: tile-by-tile kernel (fk 2) vs whole-sequence kernel (fk 3), hipGraph replay")
        import ctypes as C
        for B in (32, 64):
            H, T, D = 16, 577, 64
            qkv = rnd(B, T, 3 * H * D)
            out = torch.empty(B, T, H * D, dtype=torch.bfloat16, device=DEV)
            for fk in (2, 3):
                a = _lib.AttnArgs()
                base = qkv.data_ptr()
                a.q, a.k, a.v, a.o = base, base + H * D * 2, base + 2 * H * D * 2, out.data_ptr()
                a.q_bs = a.k_bs = a.v_bs = T * 3 * H * D
                a.q_hs = a.k_hs = a.v_hs = D
                a.q_rs = a.k_rs = a.v_rs = 3 * H * D
                a.o_bs, a.o_hs, a.o_rs = T * H * D, D, H * D
                a.B, a.H, a.Tq, a.Tk, a.D = B, H, T, T, D
                a.scale, a.causal, a.force_kernel = 1 / math.sqrt(D), 0, fk
                L = _lib.load()
                def run():
                    for _ in range(4):
                        _lib.check(L.vcla_attention(C.byref(a), _lib.dtype_code(torch.bfloat16), _lib.stream_ptr()))
                t = timeit_graph(run, reps=10) / 4
                fl = 4.0 * B * H * T * T * D
                print(f"attn577 B={B} fk={fk}  {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s ({fl/t/1e12/25:.1f}% of the bf16 MFMA peak)")
    if "attn" in which:
        print("== attention")
        for fk in (1, 2):
            bench_attn("vit (B=64)", 64, 16, 257, 257, 64, False, fk)
            bench_attn("resampler (B=64)", 64, 16, 64, 321, 64, False, fk)
            bench_attn("llama prefill (B=64,T=128)", 64, 32, 128, 128, 128, True, fk)
            bench_attn("llama prefill (B=1,T=128)", 1, 32, 128, 128, 128, True, fk)
            bench_attn("llama prefill (B=4,T=1024)", 4, 32, 1024, 1024, 128, True, fk)


if __name__ == "__main__":
    main()
